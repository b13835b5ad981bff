#!/bin/bash
# usage (GPU box): tools/time_variants.sh [K]  -- per-kernel times of every gpurun_variants/*.so
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
K=${1:-2000}
for so in $ROOT/gpurun_variants/*.so; do
  echo "== $(basename $so)"
  M3P2I_HIP_LIB=$so python $ROOT/tools/lanes_sweep.py $K 2>&1 | grep "'lanes': 0" | cut -c1-160
done
