#!/bin/bash
# usage (on the GPU box): tools/pmc_rollout.sh <tag> <K> <lanes> [task]
TAG=$1; K=$2; LANES=$3; TASK=${4:-push}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/a -o p -- python $ROOT/tools/run_rollout.py $K $LANES $TASK > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS_F32 SQ_BUSY_CU_CYCLES SQ_WAVES --output-format csv -d $OUT/b -o p -- python $ROOT/tools/run_rollout.py $K $LANES $TASK > $OUT/b.log 2>&1
python - <<PY
import csv, glob, collections
for sub in ("a", "b"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "k_rollout" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("$TAG K=$K lanes=$LANES", {k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
