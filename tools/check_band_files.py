"""The assertions of tests/test_behaviour_band_gpu.py evaluated on stored statistics (profiles/rNN/behaviour_stats_*.json, written by
tools/band_stats.py --n N --json ...): which of them the N = 20 / 60 runs meet.  CPU only.

    python tools/check_band_files.py profiles/r06/behaviour_stats_baseline.json [more files ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from cpu_fit_physx import BAND, LOGGED_SUCCESS  # noqa: E402

FASTER_THAN_LOGGED = ("corner2_push", "corner2_pull")       # (tests/test_behaviour_band_gpu.py: task time asserted on the slow side only)


def check(name, r):
    n, band, out = r["n"], BAND[name], []
    if r["successes"] < int(LOGGED_SUCCESS[name] * n):
        out.append("successes %d < %d" % (r["successes"], int(LOGGED_SUCCESS[name] * n)))
    for key in ("final_pos_error_m", "task_time_s"):
        ours, ref = r[key], band[key]
        if ours is None:
            continue
        z = (ours["mean"] - ref["mean"]) / ref["std"]
        if key == "final_pos_error_m" or name in FASTER_THAN_LOGGED:
            if z > 3.0:
                out.append("%s mean z = %.1f" % (key, z))
        elif abs(z) > 3.0:
            out.append("%s mean z = %.1f" % (key, z))
        if ours["std"] > 3.0 * ref["std"]:
            out.append("%s std %.3g > 3 x %.3g" % (key, ours["std"], ref["std"]))
    p = band["dyn_obs_collisions"]["mean"]
    bound = n * p + 3.0 * (n * p * (1.0 - p)) ** 0.5
    if r["dyn_obs_collided_episodes"] > bound:
        out.append("collided %d > %.2f" % (r["dyn_obs_collided_episodes"], bound))
    return out


for path in sys.argv[1:]:
    d = json.load(open(path))
    print(path)
    for name, r in d.items():
        if name not in BAND:
            continue
        v = check(name, r)
        t = r["task_time_s"]
        print("    %-24s n %2d  ok %2d  coll %2d  time %s   %s" % (name, r["n"], r["successes"], r["dyn_obs_collided_episodes"],
                                                                "%.2f+-%.2f" % (t["mean"], t["std"]) if t else "-", "; ".join(v) if v else "inside the band"))
