"""Why do the first commands of a process take longer than later ones?  Per-command rollout durations (the library's
HIP events) of the C2 workload for three planners built one after the other in ONE process: if the plan's warm start
(state of the kernel) were the cause, each fresh planner would repeat the curve; if it is the device's clock / power
state after idling, only the first does.

    python tools/warm_state.py [--n 120] [--out gpurun_out/warm_state.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def curve(n, device, idle_s=0.0):
    env, task, goal, mm, K, T = bench.CONFIGS["push"]
    pl, sim, obj, cfg = bench.build_tamp(env, task, goal, mm, K, 0, 1, T, device)
    state = sim._dof_state[0]
    pl.command(state)          # (the init-time sampler runs inside the first command)
    torch.cuda.synchronize()
    if idle_s:
        time.sleep(idle_s)
    eng = pl._engine
    eng.enable_timing(True)
    out, wall = [], []
    for _ in range(n):
        t0 = time.perf_counter()
        pl.command(state)
        t = eng.timing()
        wall.append((time.perf_counter() - t0) * 1e3)
        out.append(t.rollout_ms)
    eng.enable_timing(False)
    # the same commands back to back without reading events (the bench's timed region)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        pl.command(state)
    torch.cuda.synchronize()
    b2b = (time.perf_counter() - t0) / 20 * 1e3
    return dict(rollout_ms=[round(float(x), 4) for x in out], back_to_back_ms_per_command=b2b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=120)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "warm_state.json"))
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    res = {}
    for name, idle in (("first_planner_of_the_process", 0.0), ("second_planner", 0.0), ("third_planner_after_2s_idle", 2.0)):
        r = curve(a.n, dev, idle)
        x = np.asarray(r["rollout_ms"])
        r["mean_first_20"] = float(x[:20].mean())
        r["mean_20_45"] = float(x[20:45].mean())
        r["mean_last_20"] = float(x[-20:].mean())
        res[name] = r
        print(name, "first20 %.4f  20..45 %.4f  last20 %.4f  b2b %.4f" % (r["mean_first_20"], r["mean_20_45"], r["mean_last_20"],
                                                                       r["back_to_back_ms_per_command"]), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
