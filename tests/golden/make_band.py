"""Behavioural band of the reference's closed-loop runs -> tests/golden/behaviour_band.json.

Run in the build container only (reads the reference's recorded experiment logs):

    python tests/golden/make_band.py

The reference ships the logs of its own Isaac-Gym runs (src/m3p2i_aip/plot/point/*.npy, plot/panda/*.npy;
column legend plot/plot_point.py:26-34, plot/plot_panda.py:19-24).  PhysX itself is unpinnable (closed
binary), so these numbers are the only statement the reference makes about how its closed loop BEHAVES:
final block-to-goal error, task time, planner rate.  The fixture holds, per scenario, the statistics of
those columns; tests/test_behaviour_band_gpu.py runs the same scenarios on this build's integrator and
asserts success and a final error inside the logged band.  Data only -- nothing of the reference's code.
"""
import json
import os

import numpy as np

PLOT = "/root/reference/src/m3p2i_aip/plot"
HERE = os.path.dirname(os.path.abspath(__file__))


def stats(x):
    x = np.asarray(x, np.float64)
    return {"mean": float(x.mean()), "std": float(x.std()), "min": float(x.min()), "max": float(x.max()), "n": int(x.size)}


def point(name):
    d = np.load(os.path.join(PLOT, "point", name + ".npy"))
    err = np.linalg.norm(d[:, 5:7] - d[:, 12:14], axis=1)          # block xy vs block goal (plot_point.py:42-43)
    return {"goal": [float(np.round(d[0, 12], 3)), float(np.round(d[0, 13], 3))], "final_pos_error_m": stats(err),
            "task_time_s": stats(d[:, 18]), "command_hz": stats(d[:, 16]), "dyn_obs_collisions": stats(d[:, 17])}


def panda(name):
    d = np.load(os.path.join(PLOT, "panda", name + ".npy"))
    err = np.linalg.norm(d[:, 1:3] - d[:, 8:10], axis=1)            # cube xy vs goal xy (plot_panda.py:26-27)
    return {"final_xy_error_m": stats(err)}


band = {
    "_source": "statistics of the reference's recorded runs, src/m3p2i_aip/plot/{point,panda}/*.npy "
               "(hardware, K and T of those runs are not recorded; repo defaults K=200, T=15)",
    "point": {n: point(n) for n in ("case2_halton_push_coll", "case2_halton_pull_coll", "corner1_push", "corner1_pull",
                                    "corner1_hybrid", "corner2_push", "corner2_pull", "corner2_hybrid")},
    "panda": {n: panda(n) for n in ("normal_pick", "reactive_pick")},
    "success_threshold_m": {"point": 0.1, "panda_place_xy": 0.04},   # task_planner.py:17,104
}
path = os.path.join(HERE, "behaviour_band.json")
json.dump(band, open(path, "w"), indent=1)
print("wrote", path)
for k, v in band["point"].items():
    print(k, v["goal"], "err %.3f+-%.3f [%.3f, %.3f]" % tuple(v["final_pos_error_m"][q] for q in ("mean", "std", "min", "max")),
          "time %.1f+-%.1f" % (v["task_time_s"]["mean"], v["task_time_s"]["std"]), "hz %.1f" % v["command_hz"]["mean"])
for k, v in band["panda"].items():
    print(k, v["final_xy_error_m"])
