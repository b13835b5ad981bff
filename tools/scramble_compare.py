"""Plain vs generalized (Faure-permuted) Halton knots (MPPIConfig.halton_scramble), VERDICT r2 item 6:
  (1) sample-set quality: max / mean |pairwise correlation| of the uniform knots in dims 30-45 of the panda_env's
      45-dimensional set, and the same for the low dims, at the BASELINE sizes;
  (2) closed loop: the panda_env reactive pick (45-dim knots) and the point_env push with either set, at the
      BASELINE size and at a reduced sample count (where the quality of the set matters more): success tick and
      final error.
    python tools/scramble_compare.py [--json out.json]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def quality():
    from m3p2i_aip_amd import sampling as S
    out = {}
    for K in (500, 4000):
        for sc in S.SCRAMBLES:
            u = S.halton_uniform(K, 45, sc).double().numpy()
            row = {}
            for name, sl in (("dims_1_15", slice(0, 15)), ("dims_16_30", slice(15, 30)), ("dims_30_45", slice(29, 45))):
                c = np.corrcoef(u[:, sl].T)
                np.fill_diagonal(c, 0.0)
                row[name] = {"max_abs_corr": float(np.abs(c).max()), "mean_abs_corr": float(np.abs(c).mean())}
            out[f"K{K}_{sc}"] = row
    return out


def closed_loops():
    import closed_loop
    out = {}
    for tag, cn, base, ticks in (("panda_K4000_T20", "config_panda", ["mppi.num_samples=4000", "mppi.horizon=20"], 600),
                                 ("panda_K500_T20", "config_panda", ["mppi.num_samples=500", "mppi.horizon=20"], 600),
                                 ("panda_K200_T12", "config_panda", ["mppi.num_samples=200", "mppi.horizon=12"], 800),
                                 ("push_K2000_T30", "config_point", ["task=push", "goal=[-1,-1]", "mppi.num_samples=2000", "mppi.horizon=30"], 800),
                                 ("push_K200_T15", "config_point", ["task=push", "goal=[-1,-1]", "mppi.num_samples=200", "mppi.horizon=15"], 800)):
        for sc in ("none", "faure"):
            r = closed_loop.run(cn, base + [f"mppi.halton_scramble={sc}"], ticks=ticks)
            keep = {k: r.get(k) for k in ("success", "ticks", "sim_time_s", "timeline", "final_pos_error", "cube_to_goal_xy",
                                          "cube_height_above_goal", "command_ms_p50")}
            out[f"{tag}_{sc}"] = keep
            print(tag, sc, keep, flush=True)
    return out


if __name__ == "__main__":
    res = {"sample_set": quality()}
    for k, v in res["sample_set"].items():
        print(k, v)
    res["closed_loop"] = closed_loops()
    if "--json" in sys.argv:
        p = sys.argv[sys.argv.index("--json") + 1]
        os.makedirs(os.path.dirname(p) or ".", exist_ok=True)
        json.dump(res, open(p, "w"), indent=1)
