#!/bin/bash
# usage (GPU box): tools/time_variants_sweep.sh "K1,K2,..."  -- tools/k_sweep.py per gpurun_variants/*.so
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for so in $ROOT/gpurun_variants/*.so; do
  echo "== $(basename $so)"
  M3P2I_HIP_LIB=$so python $ROOT/tools/k_sweep.py "$1" 2>&1 | grep rollout_ms | cut -c1-200
done
