cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for so in gpurun_variants/*.so; do
  for lps in 16 1; do
    echo -n "$(basename $so) lps$lps panda_pick: "
    M3P2I_PANDA_LPS=$lps M3P2I_HIP_LIB=$PWD/$so timeout 300 python bench.py --config panda_pick --no-cpu-baseline --no-extras --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), d['kernel_ms']['rollout'])"
  done
done 2>&1 | tee gpurun_out/r05/ablate_lps.txt
