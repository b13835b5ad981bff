"""What the per-command collectives cost on the host + stream of ONE rank (1-GPU box, world_size = 1 RCCL):
the transport-independent part of the sharded command() -- torch.distributed call overhead, RCCL kernel
launch, stream dependencies -- measured inside the real command loop on the real library-owned buffers.
(The xGMI hop itself needs a multi-GPU node; it adds wire latency to the numbers below, not host time.)

    python tools/collective_overhead.py [--config c5|push|hybrid] [--steps 300] [--json out.json]

Per config three loops over the same planner (K per GPU as in bench.py):
  fused        unsharded command(): rollout + one-launch update (what N = 1 runs)
  split        the sharded phase sequence with the exchanges stubbed out (what sharding costs in launches)
  collectives  the same with world_size-1 RCCL collectives on the buffers (all_gather / all_reduce)
and HIP-event time of each collective on the stream.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def loop(pl, state, steps):
    for _ in range(20):
        pl.command(state)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pl.command(state)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c5")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    from m3p2i_aip_amd.distributed import attach_collectives
    env, task, goal, mm, K, T = bench.CONFIGS[a.config]
    out = {"config": a.config, "K": K, "T": T, "multi_modal": mm, "steps": a.steps, "backend": "nccl (RCCL), world_size 1"}
    pl, sim, obj, cfg = bench.build_tamp(env, task, goal, mm, K, 0, 1, T, "cuda:0")
    state = sim._dof_state[0]
    out["fused_ms"] = loop(pl, state, a.steps)
    pl.collective = lambda p, phase: None
    out["split_no_collectives_ms"] = loop(pl, state, a.steps)
    attach_collectives(pl)
    out["split_with_collectives_ms"] = loop(pl, state, a.steps)
    pl.collective_times = []
    for _ in range(100):
        pl.command(state)
    torch.cuda.synchronize()
    per = {}
    for ph, e0, e1 in pl.collective_times:
        per.setdefault(ph, []).append(e0.elapsed_time(e1))
    out["collective_stream_ms"] = {ph: {"mean": float(np.mean(v)), "p50": float(np.percentile(v, 50)),
                                        "p99": float(np.percentile(v, 99))} for ph, v in per.items()}
    out["collectives_per_command"] = len(pl.collective_times) / 100
    pl.collective_times = None
    out["overhead_per_command_ms"] = out["split_with_collectives_ms"] - out["split_no_collectives_ms"]
    # host time of the calls alone (no stream work in between)
    e = pl._engine
    from m3p2i_aip_amd import _lib as L
    bufs = {"gather": (e.buffer(L.BUF_TRAJ_COST_ALL), e.buffer(L.BUF_TRAJ_COST)), "reduce": (e.buffer(L.BUF_REDUCE),)}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        dist.all_gather_into_tensor(*bufs["gather"])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(200):
        dist.all_reduce(bufs["reduce"][0])
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    out["host_call_us"] = {"all_gather_into_tensor": (t1 - t0) / 200 * 1e6, "all_reduce": (t3 - t2) / 200 * 1e6}
    print(json.dumps(out))
    if a.json:
        os.makedirs(os.path.dirname(a.json) or ".", exist_ok=True)
        json.dump(out, open(a.json, "w"), indent=1)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
