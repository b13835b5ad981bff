"""GPU: the Panda REACH command in the scenes an episode actually passes through -- the product's closed loop (tools/closed_loop.py)
stopped 20 / 40 / 60 ticks into its reach phase (the gripper on its way down to cubeA; the pick starts at tick ~66) -- per kernel
form (lanes per sample 1 / 8 / 16).  bench.py's reach rows are the episode's FIRST command and the cubes-settled scene with the arm
still at its initial pose; this is what the ticks in between cost.
    python tools/panda_reach_mid_bench.py [--json out.json]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
import closed_loop  # noqa: E402
from types import SimpleNamespace  # noqa: E402


def reach_scene(cap):
    def scene(pl, sim, obj, cfg):
        dev = sim._dof_state.device
        sim._dof_state[:] = torch.tensor(cap["dof_state"], device=dev)
        sim._root_state[:] = torch.tensor(cap["root_state"], device=dev)
        sim.set_dof_state_tensor(sim._dof_state)
        sim.set_actor_root_state_tensor(sim._root_state)
    return scene


def main():
    out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    device = "cuda:0"
    args = SimpleNamespace(shard_mix=None, transport="rccl")
    rows = []
    scenes = [("reach, initial scene", lambda *a: None), ("reach, cubes settled", bench.settled_panda_scene)]
    for tick in (10, 20, 30, 40, 50, 60):
        res = closed_loop.run("config_panda", ["mppi.num_samples=4000", "mppi.horizon=20", f"mppi.device={device}"],
                              ticks=400, until_task="reach", extra_ticks=tick)
        cap = res["captured"]
        assert cap["task"] == "reach", cap["task"]
        scenes.append((f"reach, tick {tick} of the episode", reach_scene(cap)))
    for name, base in scenes:
        for lps in (1, 8, 16, 0):     # 0: the automatic choice (by the share of near (sample, substep) pairs the kernel reports)
            def sc(pl, sim, obj, cfg, lps=lps, base=base):
                base(pl, sim, obj, cfg)
                pl._engine.set_panda_lanes_per_sample(lps)
            r = bench.run_config("panda", args, 1, 0, device, None, 100, 10, scene=sc, latency=False)
            e = r["pl"]._engine
            rows.append(dict(scene=name, lanes_per_sample=lps, ms_per_command=r["ms_per_step"], rollout_ms=r["rollout_ms"],
                             update_ms=r["update_ms"], near_share_permille=e.panda_near_share(), form_used=e.panda_lanes_per_sample_used()))
            print(rows[-1], flush=True)
            r["pl"]._engine.close()
    if out:
        json.dump(dict(K=4000, T=20, rows=rows), open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
