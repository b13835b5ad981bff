cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for lps in 16 8; do
  echo "=== LPS $lps"
  M3P2I_PANDA_LPS=$lps timeout 900 python -m pytest tests/test_hip_parity_panda.py tests/test_full_size_oracle_parity_gpu.py tests/test_planner_api_panda_gpu.py -q -k "panda or Panda" 2>&1 | tail -4
  for c in panda_pick panda; do
    echo -n "lps$lps $c: "
    M3P2I_PANDA_LPS=$lps timeout 300 python bench.py --config $c --no-cpu-baseline --no-extras --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), d['kernel_ms'])"
  done
done 2>&1 | tee gpurun_out/r05/lps_second.txt
for c in "panda 8" "panda_pick 16"; do set -- $c; M3P2I_HIP_LIB=$PWD/gpurun_variants/prof.so python tools/panda_wave_profile.py $1 --lps $2 2>&1 | tail -1 > /tmp/p.json; python -c "
import json; d=json.load(open('/tmp/p.json')); print(d['config'], d['samples_per_wave'], d['wave_clocks']); print(d['mean_over_samples'])"; done
