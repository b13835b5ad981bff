#!/bin/bash
# Run ON THE GPU BOX (via gpurun).  rocprofv3 kernel-trace stats + two separate PMC passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots";
# never combined with --sys-trace etc.) of the default bench workload.
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 100 --warmup 10 --no-cpu-baseline $*"
export TRAFFIC_KEY=${TRAFFIC_KEY:-}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $ROOT/bench.py $ARGS > $OUT/bench_trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python $ROOT/bench.py $ARGS > $OUT/bench_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python $ROOT/bench.py $ARGS > $OUT/bench_pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o bench -- python $ROOT/bench.py $ARGS > $OUT/bench_pmc_sq.log 2>&1
find $OUT -name "*.csv" | head -20
python $ROOT/tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
