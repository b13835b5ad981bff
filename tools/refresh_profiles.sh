#!/bin/bash
# Run ON THE GPU BOX (via gpurun): regenerates everything under profiles/<round>/ into gpurun_out/.
# usage: tools/refresh_profiles.sh     (then, in the build container: python tools/collect_profiles.py profiles/r02)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out
mkdir -p $O
cd $ROOT
python bench.py > $O/bench_default.json 2> $O/bench_default.err            # the driver's line: headline + other_configs + closed loop + cpu baselines
for c in push hybrid panda northstar c5; do
  python bench.py --config $c --no-extras --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err
done
# profiling passes run the headline loop only (--no-extras): kernel-trace stats, then separate PMC passes
TRAFFIC_KEY=push:K2000:T30 tools/profile_gpu.sh push --config push --no-extras > $O/prof_push.log 2>&1
TRAFFIC_KEY=hybrid:K4000:T30 tools/profile_gpu.sh hybrid --config hybrid --no-extras > $O/prof_hybrid.log 2>&1
TRAFFIC_KEY=panda:K4000:T20 tools/profile_gpu.sh panda --config panda --no-extras > $O/prof_panda.log 2>&1
TRAFFIC_KEY=northstar:K10000:T30 tools/profile_gpu.sh northstar --config northstar --no-extras > $O/prof_northstar.log 2>&1
TRAFFIC_KEY=c5:K8000:T30 tools/profile_gpu.sh c5 --config c5 --no-extras > $O/prof_c5.log 2>&1
cd $ROOT
python tools/k_sweep.py > $O/k_sweep.log 2>&1
python tools/host_overhead.py > $O/host_overhead.txt 2>&1
python tools/closed_loop.py task=push "goal=[-1,-1]" mppi.num_samples=2000 mppi.horizon=30 --json $O/cl_push.json > $O/cl_push.log 2>&1
python tools/closed_loop.py task=push_pull multi_modal=True mppi.num_samples=4000 mppi.horizon=30 --json $O/cl_hybrid.json > $O/cl_hybrid.log 2>&1
python tools/closed_loop.py -cn config_panda mppi.num_samples=4000 mppi.horizon=20 --json $O/cl_panda.json > $O/cl_panda.log 2>&1
python tools/closed_loop_perf.py > $O/closed_loop_perf.log 2>&1
tools/pmc_rollout.sh final 2000 0 push > $O/pmc_final.txt 2>&1
tools/pmc_rollout.sh pandaf 4000 0 reach > $O/pmc_panda.txt 2>&1
# sharding: what the collectives cost on one rank, rank 0 of 8 emulated on this GPU (both protocols)
python tools/collective_overhead.py --config c5 --emulate-rank-of 8 --json $O/collective_overhead_c5.json > $O/collective_overhead_c5.log 2>&1
python tools/collective_overhead.py --config push --json $O/collective_overhead_push.json > $O/collective_overhead_push.log 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_emul -o emul -- python $ROOT/tools/collective_overhead.py --config c5 --steps 100 --emulate-rank-of 8 > $O/prof_emul.log 2>&1)
# evidence for the lane mapping
[ -x gpurun_variants/coop_rows ] && ./gpurun_variants/coop_rows 2000 > $O/coop_rows_K2000.json 2>&1
[ -f gpurun_variants/phases.so ] && M3P2I_HIP_LIB=$ROOT/gpurun_variants/phases.so python tools/phase_breakdown.py push northstar hybrid > $O/phase_breakdown.log 2>&1
[ -f gpurun_variants/count.so ] && M3P2I_HIP_LIB=$ROOT/gpurun_variants/count.so python tools/mask_count_closed_loop.py push pushcorner hybrid 400 > $O/mask_count_closed_loop.log 2>&1
echo done
