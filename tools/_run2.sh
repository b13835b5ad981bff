cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for lps in 16 8 1; do
  echo "=== LPS $lps"
  M3P2I_PANDA_LPS=$lps timeout 900 python -m pytest tests/test_hip_parity_panda.py tests/test_full_size_oracle_parity_gpu.py tests/test_planner_api_panda_gpu.py -x -q -k "panda or Panda" 2>&1 | tail -15
  for c in panda_pick panda; do
    echo -n "lps$lps $c: "
    M3P2I_PANDA_LPS=$lps timeout 300 python bench.py --config $c --no-cpu-baseline --no-extras --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), d['kernel_ms'])"
  done
done 2>&1 | tee gpurun_out/r05/lps_first.txt
