#!/bin/bash
# usage (GPU box): tools/time_variants_cfg.sh "v1 v2 base" "push hybrid northstar" [reps=2]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for rep in $(seq 1 ${3:-2}); do for c in $2; do for v in $1; do
  echo -n "$c $v "
  M3P2I_HIP_LIB=$ROOT/gpurun_variants/$v.so timeout 120 python $ROOT/bench.py --config $c --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],5), round(d['kernel_ms']['rollout'],5))"
done; done; done
