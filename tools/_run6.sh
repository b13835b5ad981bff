cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python tools/fuzz_parity.py 7000 6000 21000 2>&1 | grep -v amdgpu.ids | tail -4 > $O/fuzz_parity_final_6000_point_21000_panda.txt
timeout 900 python tools/soak.py 6000 2>&1 | grep -v amdgpu.ids | tail -8 > $O/soak_6000.txt
timeout 1500 python tools/band_stats.py --n 60 --json $O/behaviour_stats_panda_n60.json panda > $O/behaviour_stats_panda_n60.log 2>&1
timeout 1500 python tools/band_stats.py --n 60 --json $O/behaviour_stats_baseline_n60.json > $O/behaviour_stats_baseline_n60.log 2>&1
timeout 1500 python tools/band_stats.py --n 60 --size default --json $O/behaviour_stats_default_size_n60.json > $O/behaviour_stats_default_size_n60.log 2>&1
tail -3 $O/fuzz_parity_final_6000_point_21000_panda.txt $O/soak_6000.txt; tail -6 $O/behaviour_stats_panda_n60.log; tail -6 $O/behaviour_stats_baseline_n60.log; tail -6 $O/behaviour_stats_default_size_n60.log
