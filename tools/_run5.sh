cd $GRAFT_REPO_ROOT
for so in gpurun_variants/*.so; do
  for c in panda_pick panda; do
    echo -n "$(basename $so) $c: "
    M3P2I_HIP_LIB=$PWD/$so timeout 300 python bench.py --config $c --no-cpu-baseline --no-extras --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), d['kernel_ms']['rollout'])"
  done
done
