"""GPU: the Panda bench scenes (reach in the initial scene / with the cubes settled, pick from the product's own closed loop) per
kernel form (lanes per sample 1 / 8 / 16; world spec v3): command() and rollout kernel times.
    python tools/panda_lps_bench.py [--json out.json]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from types import SimpleNamespace  # noqa: E402


def main():
    out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    device = "cuda:0"
    args = SimpleNamespace(shard_mix=None, transport="rccl")
    pick = bench.make_pick_scene(bench.panda_pick_scene(device))
    rows = []
    for name, key, scene in (("panda", "reach, initial scene", None), ("panda", "reach, cubes settled", bench.settled_panda_scene),
                             ("panda_pick", "pick", pick)):
        for lps in (1, 8, 16):
            def sc(pl, sim, obj, cfg, scene=scene, lps=lps):
                if scene is not None:
                    scene(pl, sim, obj, cfg)
                pl._engine.set_panda_lanes_per_sample(lps)
            r = bench.run_config(name, args, 1, 0, device, None, 100, 10, scene=sc, latency=False)
            rows.append(dict(scene=key, lanes_per_sample=lps, ms_per_command=r["ms_per_step"], rollout_ms=r["rollout_ms"], update_ms=r["update_ms"]))
            print(rows[-1], flush=True)
            r["pl"]._engine.close()
    if out:
        json.dump(dict(K=4000, T=20, rows=rows), open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
