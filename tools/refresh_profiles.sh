#!/bin/bash
# Run ON THE GPU BOX (via gpurun): regenerates everything under profiles/<round>/ into gpurun_out/.
# usage: tools/refresh_profiles.sh     (then, in the build container: python tools/collect_profiles.py profiles/r06)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out
mkdir -p $O
cd $ROOT
# dynamic instruction mix of the rollout kernels FIRST (PMC passes; the thread-trace decoder is not in this image:
# tools/att_rollout.sh): mix_<config>.json is what bench.py's roofline_valu reads, keyed by m3_build_id -- the bench lines
# below must find the mixes of THIS build under profiles/r06/ (copied there on the box; merged back through gpurun_out/)
for c in push hybrid panda panda_pick northstar c5 c5_unsharded worst_case; do
  tools/pmc_mix_bench.sh $c > $O/pmc_mix_bench_$c.log 2>&1
  cp $O/mix_$c.json $ROOT/profiles/r06/mix_$c.json
done
# the driver's own command, verbatim (BENCH_rNN.json: `python3 bench.py --gpus 1 --steps 20 --warmup 5`)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err            # 200 steps / 20 warm-ups: headline + other_configs + closed loop + cpu baselines
for c in push hybrid panda panda_pick northstar c5 c5_unsharded worst_case; do
  python bench.py --config $c --no-extras --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err
done
# profiling passes run the headline loop only (--no-extras): kernel-trace stats, then separate PMC passes
TRAFFIC_KEY=push:K2000:T30 tools/profile_gpu.sh push --config push --no-extras > $O/prof_push.log 2>&1
TRAFFIC_KEY=hybrid:K4000:T30 tools/profile_gpu.sh hybrid --config hybrid --no-extras > $O/prof_hybrid.log 2>&1
TRAFFIC_KEY=panda:K4000:T20 tools/profile_gpu.sh panda --config panda --no-extras > $O/prof_panda.log 2>&1
TRAFFIC_KEY=panda_pick:K4000:T20 tools/profile_gpu.sh panda_pick --config panda_pick --no-extras > $O/prof_panda_pick.log 2>&1
TRAFFIC_KEY=northstar:K10000:T30 tools/profile_gpu.sh northstar --config northstar --no-extras > $O/prof_northstar.log 2>&1
TRAFFIC_KEY=c5:K8000:T30 tools/profile_gpu.sh c5 --config c5 --no-extras > $O/prof_c5.log 2>&1
TRAFFIC_KEY=c5_unsharded:K64000:T30 tools/profile_gpu.sh c5_unsharded --config c5_unsharded --no-extras > $O/prof_c5_unsharded.log 2>&1
TRAFFIC_KEY=worst_case:K2000:T30 tools/profile_gpu.sh worst_case --config worst_case --no-extras > $O/prof_worst_case.log 2>&1
cd $ROOT
# ESSENTIAL=1: only what is keyed by m3_build_id (mixes, bench lines, kernel stats, PMC traffic) -- enough after a change that
# leaves the trajectories alone
if [ -n "$ESSENTIAL" ]; then echo done-essential; exit 0; fi
python tools/k_sweep.py > $O/k_sweep.log 2>&1
python tools/k_sweep.py 4000,16000,65536,262144 panda > $O/k_sweep_panda.log 2>&1
python tools/panda_lps_bench.py --json $O/panda_lps_bench.json > $O/panda_lps_bench.log 2>&1
python tools/host_overhead.py > $O/host_overhead.txt 2>&1
python tools/closed_loop.py task=push "goal=[-1,-1]" mppi.num_samples=2000 mppi.horizon=30 --json $O/cl_push.json > $O/cl_push.log 2>&1
python tools/closed_loop.py task=push_pull multi_modal=True mppi.num_samples=4000 mppi.horizon=30 --json $O/cl_hybrid.json > $O/cl_hybrid.log 2>&1
python tools/closed_loop.py -cn config_panda mppi.num_samples=4000 mppi.horizon=20 --json $O/cl_panda.json > $O/cl_panda.log 2>&1
python tools/closed_loop_perf.py > $O/closed_loop_perf.log 2>&1
tools/pmc_rollout.sh final 2000 0 push > $O/pmc_final.txt 2>&1
tools/pmc_rollout.sh pandaf 4000 0 reach > $O/pmc_panda.txt 2>&1
# dynamic instruction mix of the rollout kernels (the thread-trace decoder is not in this image: tools/att_rollout.sh)
tools/pmc_mix.sh push_K2000 2000 push > $O/pmc_mix_push.log 2>&1
python tools/codeobj_info.py --isa "k_rollout_point<false, 1>" --json $O/codeobj_info.json > $O/codeobj_info.txt 2>&1
# behaviour: N = 20 jittered episodes per scenario (tests/test_behaviour_band_gpu.py asserts on the first and the last)
python tools/band_stats.py --n 20 --json $O/behaviour_stats_baseline.json > $O/behaviour_stats_baseline.log 2>&1
python tools/band_stats.py --n 20 --size default --json $O/behaviour_stats_default_size.json > $O/behaviour_stats_default_size.log 2>&1
python tools/band_stats.py --n 20 --json $O/behaviour_stats_panda.json panda > $O/behaviour_stats_panda.log 2>&1
python tools/scramble_compare.py --json $O/halton_scramble.json > $O/halton_scramble.log 2>&1
# sharding: what the collectives cost on one rank, rank 0 of 8 emulated on this GPU (RCCL and the p2p exchange)
python tools/collective_overhead.py --config c5 --emulate-rank-of 8 --json $O/collective_overhead_c5.json > $O/collective_overhead_c5.log 2>&1
python tools/collective_overhead.py --config push --json $O/collective_overhead_push.json > $O/collective_overhead_push.log 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_emul -o emul -- python $ROOT/tools/collective_overhead.py --config c5 --steps 100 --emulate-rank-of 8 > $O/prof_emul.log 2>&1)
echo done
