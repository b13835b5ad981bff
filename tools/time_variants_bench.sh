#!/bin/bash
# usage (GPU box): tools/time_variants_bench.sh [config ...] -- bench.py ms_per_step of every gpurun_variants/*.so
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
CFGS=${@:-push hybrid}
for so in $ROOT/gpurun_variants/*.so; do
  for c in $CFGS; do
    echo -n "$(basename $so) $c: "
    M3P2I_HIP_LIB=$so python $ROOT/bench.py --config $c --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), d['kernel_ms'])"
  done
done
