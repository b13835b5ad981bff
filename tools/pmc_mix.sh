#!/bin/bash
# Run ON THE GPU BOX.  Dynamic instruction mix of the rollout kernel from PMC counters (the thread-trace decoder
# library is not in this image: rocprofv3 --att stops with "rocprof-trace-decoder library path not found").
# usage: tools/pmc_mix.sh <tag> <K> <task>
TAG=$1; K=$2; TASK=${3:-push}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/mix_$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { rocprofv3 --pmc $2 --output-format csv -d $OUT/$1 -o p -- python $ROOT/tools/run_rollout.py $K 0 $TASK > $OUT/$1.log 2>&1; }
run a "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_BRANCH"
run b "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VSKIPPED SQ_INSTS_SENDMSG"
run c "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU"
run d "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_INSTS_FLAT"
python - <<PY
import csv, glob, collections, json
tot = {}
for sub in "abcd":
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not f: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "k_rollout" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[len(v) // 2:]          # converged plan: second half of the launches
        tot[k] = sum(v) / len(v)
w = tot.get("SQ_WAVES", 1.0)
per = {k: v / w for k, v in tot.items()}
per["waves"] = w
if "SQ_INSTS" in per:
    per["MISC_derived (s_nop, s_waitcnt, ...)"] = per["SQ_INSTS"] - sum(per.get(k, 0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM", "SQ_INSTS_BRANCH"))
if "SQ_INSTS_VALU" in per:
    per["VALU_other_derived (v_mov, v_cndmask, v_cmp, readlane, accvgpr, pk_*)"] = per["SQ_INSTS_VALU"] - sum(per.get(k, 0) for k in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT"))
print("$TAG K=$K $TASK per wave:", json.dumps({k: round(v, 1) for k, v in per.items()}, indent=1))
json.dump(per, open("$OUT/../mix_$TAG.json", "w"), indent=1)
PY
