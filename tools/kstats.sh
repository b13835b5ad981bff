#!/bin/bash
# usage (GPU box): tools/kstats.sh <K> <lanes> <task> [n]  -> per-kernel avg/min/max ns
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
D=$ROOT/gpurun_out/kstats_$$
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o b -- python $ROOT/tools/run_rollout.py "$@" > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("$D/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "m3::" in r["Name"]: print("%-28s calls=%s avg=%.1fus min=%.1f max=%.1f" % (r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
rm -rf $D
