"""Copies what tools/refresh_profiles.sh left in gpurun_out/ into profiles/<round>/ (tracked) and
merges the PMC traffic fragments into profiles/traffic.json (read by bench.py's roofline.traffic).
usage: python tools/collect_profiles.py [round_dir=profiles/r01]"""
import json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "profiles/r06")
os.makedirs(P, exist_ok=True)
NAMES = {"push": "push_K2000_T30", "hybrid": "hybrid_K4000_T30", "panda": "panda_K4000_T20", "panda_pick": "panda_pick_K4000_T20",
         "northstar": "northstar_K10000_T30", "c5": "c5shard_K8000_T30", "c5_unsharded": "c5_unsharded_K64000_T30",
         "worst_case": "worst_case_scene_K2000_T30"}


def cp(src, dst):
    s = os.path.join(G, src)
    if os.path.exists(s):
        shutil.copyfile(s, os.path.join(P, dst))
        print("copied", dst)
    else:
        print("MISSING", src)


for c, n in NAMES.items():
    s = os.path.join(G, f"bench_{c}.json")
    if os.path.exists(s):
        line = open(s).read().strip().splitlines()[-1]
        json.dump(json.loads(line), open(os.path.join(P, f"bench_{n}.json"), "w"), indent=1)
        print("copied", f"bench_{n}.json")
    cp(f"prof_{c}/trace/bench_kernel_stats.csv", f"bench_{n}_kernel_stats.csv")
    cp(f"prof_{c}/summary.txt", f"bench_{n}_summary.txt")
    cp(f"cl_{c}.json", f"closed_loop_{n}.json")
s = os.path.join(G, "bench_default.json")
if os.path.exists(s):
    json.dump(json.loads(open(s).read().strip().splitlines()[-1]), open(os.path.join(P, "bench_default_line.json"), "w"), indent=1)
    print("copied bench_default_line.json")
s = os.path.join(G, "bench_driver_cmd.json")
if os.path.exists(s):
    json.dump(json.loads(open(s).read().strip().splitlines()[-1]), open(os.path.join(P, "bench_driver_command_line.json"), "w"), indent=1)
    print("copied bench_driver_command_line.json")
for f in ("behaviour_stats_baseline.json", "behaviour_stats_default_size.json", "behaviour_stats_panda.json", "halton_scramble.json",
          "codeobj_info.txt", "mix_push_K2000.json", "mix_push.json", "mix_hybrid.json", "mix_panda_pick.json", "mix_panda.json",
          "mix_northstar.json", "mix_c5.json", "mix_c5_unsharded.json", "mix_worst_case.json"):
    cp(f, f)
for f in ("collective_overhead_c5.json", "collective_overhead_push.json", "closed_loop_perf.json"):   # (coop_rows / phase_breakdown /
    # mask_count are one-off experiments of rounds 2-3: their own tools write them, tools/refresh_profiles.sh does not)
    cp(f, f)
cp("prof_emul/emul_kernel_stats.csv", "rank0_of_8_emulation_kernel_stats.csv")
cp("host_overhead.txt", "host_overhead.txt")
cp("k_sweep.json", "k_sweep_push_T30.json")
cp("k_sweep_panda.json", "k_sweep_panda_T20.json")
cp("panda_lps_bench.json", "panda_lps_bench.json")
cp("pmc_final.txt", "pmc_rollout_push_K2000.txt")
cp("pmc_panda.txt", "pmc_rollout_panda_K4000.txt")
for f in ("panda_reach_mid_bench.json", "tamp_tick_profile.json", "behaviour_stats_baseline_n60.json", "behaviour_stats_default_size_n60.json",
          "behaviour_stats_panda_n60.json", "soak_6000.txt", "rcp_ab.txt", "prcp_ab.txt"):    # (written by their own tools when run)
    if os.path.exists(os.path.join(G, f)):
        cp(f, f)

tf = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(tf)) if os.path.exists(tf) else {}
for c in NAMES:
    f = os.path.join(G, f"prof_{c}", "traffic_fragment.json")
    if os.path.exists(f):
        tj.update(json.load(open(f)))
json.dump(tj, open(tf, "w"), indent=1)
print("traffic keys", [k for k in tj if not k.startswith("_")])
