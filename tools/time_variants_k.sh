#!/bin/bash
# usage (GPU box): tools/time_variants_k.sh <kernel-substring> [K] [task] -- rocprof avg of one kernel per variant
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for so in $ROOT/gpurun_variants/*.so; do
  echo "== $(basename $so) $(M3P2I_HIP_LIB=$so $ROOT/tools/kstats.sh ${2:-2000} 0 ${3:-push} 30 | grep $1)"
done
