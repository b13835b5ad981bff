#!/bin/bash
# Run ON THE GPU BOX (via gpurun): regenerates everything under profiles/<round>/ into gpurun_out/.
# usage: tools/refresh_profiles.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out
cd $ROOT
for c in push hybrid panda northstar c5; do
  python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err
done
TRAFFIC_KEY=push:K2000:T30 tools/profile_gpu.sh push --config push > $O/prof_push.log 2>&1
TRAFFIC_KEY=hybrid:K4000:T30 tools/profile_gpu.sh hybrid --config hybrid > $O/prof_hybrid.log 2>&1
TRAFFIC_KEY=panda:K4000:T20 tools/profile_gpu.sh panda --config panda > $O/prof_panda.log 2>&1
TRAFFIC_KEY=northstar:K10000:T30 tools/profile_gpu.sh northstar --config northstar > $O/prof_northstar.log 2>&1
cd $ROOT
python tools/k_sweep.py > $O/k_sweep.log 2>&1
python tools/lanes_sweep.py 2000 > $O/lanes_sweep.log 2>&1
python tools/iters_sweep.py > $O/iters_sweep.txt 2>&1
python tools/host_overhead.py > $O/host_overhead.txt 2>&1
python tools/closed_loop.py task=push "goal=[-1,-1]" mppi.num_samples=2000 mppi.horizon=30 --json $O/cl_push.json > $O/cl_push.log 2>&1
python tools/closed_loop.py task=push_pull multi_modal=True mppi.num_samples=4000 mppi.horizon=30 --json $O/cl_hybrid.json > $O/cl_hybrid.log 2>&1
python tools/closed_loop.py -cn config_panda mppi.num_samples=4000 mppi.horizon=20 --json $O/cl_panda.json > $O/cl_panda.log 2>&1
tools/pmc_rollout.sh final 2000 0 push > $O/pmc_final.txt 2>&1
tools/pmc_rollout.sh pandaf 4000 0 reach > $O/pmc_panda.txt 2>&1
echo done
