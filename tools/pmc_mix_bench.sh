#!/bin/bash
# Run ON THE GPU BOX.  Dynamic instruction mix of the rollout kernel of a bench config (scene included) from PMC
# counters.   usage: tools/pmc_mix_bench.sh <config>
CFG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/mixb_$CFG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() { rocprofv3 --pmc $2 --kernel-include-regex "k_rollout" --output-format csv -d $OUT/$1 -o p -- python $ROOT/bench.py --config $CFG --no-extras --no-cpu-baseline --steps 30 --warmup 5 > $OUT/$1.log 2>&1; }
run a "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_BRANCH"
run b "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"
run c "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_WAIT_INST_ANY"
python - <<PY
import csv, glob, collections, json
tot = {}
names = set()
for sub in "abc":
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not f: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "k_rollout" in r["Kernel_Name"]:
            names.add(r["Kernel_Name"].split("(")[0])
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[-30:]           # the launches of the timed region of the LAST run_config (the headline config)
        tot[k] = sum(v) / len(v)
w = tot.get("SQ_WAVES", 1.0)
per = {k: v / w for k, v in tot.items()}
per["waves"] = w
per["kernels"] = sorted(names)
if "SQ_INSTS" in per:
    per["MISC_derived"] = per["SQ_INSTS"] - sum(per.get(k, 0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM", "SQ_INSTS_BRANCH"))
# the key bench.py's roofline_valu checks: the build these counters were taken from (m3_build_id: a hash of the kernel
# sources + compiler flags, the same for every rebuild of the same tree)
import sys
sys.path.insert(0, "$ROOT")
from m3p2i_aip_amd import _lib as L
per["build_id"] = L.load().m3_build_id().decode()
print("$CFG per wave:", json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in per.items()}, indent=1))
json.dump(per, open("$OUT/../mixb_$CFG.json", "w"), indent=1)
json.dump(per, open("$OUT/../mix_$CFG.json", "w"), indent=1)      # (the name bench.py looks for under profiles/r06/)
PY
