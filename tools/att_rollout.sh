#!/bin/bash
# Run ON THE GPU BOX (via gpurun): thread trace (ATT / SQTT) attempt on the lone rollout wave at C2, plus the list
# of counters this box offers.  Output under gpurun_out/att/.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/att
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/list_avail.txt 2>&1
grep -c "" $OUT/list_avail.txt
# ATT: needs the trace decoder library (librocprof-trace-decoder.so), a separate download that this image may lack
timeout 240 rocprofv3 --att --att-target-cu 0 --att-simd-select 0xF --att-shader-engine-mask 0xFFFFFFFF \
    --kernel-include-regex "k_rollout_point" --kernel-iteration-range "[5-5]" \
    -d $OUT/att -o att -- python $ROOT/tools/run_rollout.py 2000 0 push 10 > $OUT/att.log 2>&1
echo "att rc=$?" >> $OUT/att.log
tail -20 $OUT/att.log
find $OUT/att -type f | head -40
du -sh $OUT/att
