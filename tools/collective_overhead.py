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

  --emulate-rank-of N   additionally: rank 0's share of an N-rank run of the config (K_global = N x K),
                        on this one GPU -- its real kernel sequence (rollout of K samples, records, the update
                        on all N x K costs) with the collectives as world_size-1 RCCL calls into rank 0's slot;
                        the other ranks' records / costs are filled once with shifted copies.  All three protocols:
                        one collective (shard_mix 2, the default, and 1, the bit-identical variant) and all-gather
                        + all-reduce.  What is missing vs a real
                        node: the xGMI wire time of the collective, nothing else.
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


def emulate_rank(name, N, steps, K_override=None):
    """rank 0 of N on one GPU (see module docstring)."""
    from m3p2i_aip_amd import _lib as L
    env, task, goal, mm, K, T = bench.CONFIGS[name]
    K = K_override or K
    res = {}
    only = os.environ.get("M3P2I_EMUL_PROTOCOLS")      # (profiling runs: one protocol at a time)
    for label, mix in (("one_collective", None), ("one_collective_p2p", None), ("one_collective_exact", 1), ("gather_reduce", False),
                       ("two_small_exchanges", 3)):
        if only and label not in only.split(","):
            continue
        p2p = label.endswith("_p2p")
        from m3p2i_aip_amd import isaacgym_wrapper as wrapper
        from m3p2i_aip_amd.cost_functions import Objective
        from m3p2i_aip_amd.planner import M3P2I, MPPIConfig
        from types import SimpleNamespace
        m = MPPIConfig(num_samples=K * N, horizon=T, nx=4, device="cuda:0", lambda_=0.5, u_min=[-3.0, -3.0],
                       u_max=[3.0, 3.0], noise_sigma=[[3.0, 0.0], [0.0, 3.0]], u_per_command=T, sample_null_action=True,
                       filter_u=True, fused=True, rank=0, world_size=N, shard_mix=mix)
        cfg = SimpleNamespace(env_type=env, multi_modal=mm, suction_active=True, kp_suction=400, pre_height_diff=0.0,
                              task=task, goal=list(goal), cube_on_shelf=False, mppi=m, isaacgym=wrapper.IsaacGymConfig(dt=0.05))
        sim = wrapper.IsaacGymWrapper(cfg.isaacgym, env, num_envs=64, device="cuda:0")
        obj = Objective(cfg)
        obj.update_objective(task, list(goal))
        pl = M3P2I(cfg).attach(sim, obj)
        e = pl._engine

        def exchange(p, phase, e=e, K=K):
            if phase == "records":
                dist.all_gather_into_tensor(e.buffer(L.BUF_RECORDS_ALL)[0], e.buffer(L.BUF_RECORD))
            elif phase == "records_b":
                dist.all_gather_into_tensor(e.buffer(L.BUF_RECORDS_B_ALL)[0], e.buffer(L.BUF_RECORD_B))
            elif phase == "gather":
                dist.all_gather_into_tensor(e.buffer(L.BUF_TRAJ_COST_ALL)[:K], e.buffer(L.BUF_TRAJ_COST))
            else:
                dist.all_reduce(e.buffer(L.BUF_REDUCE))
        pl.collective = exchange
        state = sim._dof_state[0]
        pl.command(state)
        torch.cuda.synchronize()
        peers = []
        if p2p:
            # the library's device-side exchange (csrc/p2p.hip) instead of RCCL: rank 0 puts its record into the
            # blocks of seven peer handles and acquires their flags; the peers (bare handles of ranks 1..N-1 on a side
            # stream, records filled once below) only put, every command -- what is missing vs a node is the xGMI hop
            from m3p2i_aip_amd.engine import HipEngine
            import copy
            side = torch.cuda.Stream()
            for r in range(1, N):
                c = copy.copy(e.cfg)
                c.k_offset = r * K
                q = HipEngine(c)
                q.use_torch_stream(side)
                peers.append(q)
            ranks = [e] + peers
            for q in ranks:
                q.p2p_connect_local(ranks)

            def exchange_p2p(p, phase, e=e, peers=peers):
                # the peers' puts for exchange n are enqueued BEFORE rank 0's rollout of that command (hook below): on this
                # one GPU every stream ends up in the same hardware queue, and seven 4.6 us put kernels between rank 0's
                # exchange and its next kernel were 32 of the 33 us first measured here -- on a node they run on the
                # other GPUs.  Rank 0's stream then carries exactly its own share: its put into the eight blocks and a
                # wait that finds the flags raised.  (ms_per_command of this row includes the peers' ~32 us.)
                assert phase == "records"
                e.p2p_exchange()
            real_rollout = e.rollout

            def rollout_with_peer_puts(peers=peers):
                for q in peers:
                    q.p2p_put()
                real_rollout()
            e.rollout = rollout_with_peer_puts
            primed = []
            switch_to_p2p = exchange_p2p      # (installed below, once the peers' records are in place)
        # the other ranks' contributions: shifted copies of rank 0's (filled once; only slot 0 is live)
        if pl.shard_mix:
            R = e.buffer(L.BUF_RECORDS_ALL)
            if p2p:   # (the same shifted copies, staged here and then installed as the peers' own records)
                R[0].copy_(e.buffer(L.BUF_RECORD))
            for r in range(1, N):
                R[r].copy_(R[0])
                R[r, :K] += 0.37 * r
                R[r, K:K + 20] += 0.37 * r
                R[r, K + 20:K + 40].view(torch.int32).add_(r * K)
                om = K + 40 + 40 * T                      # the shard's minima (shard_mix = 2 records): shifted alike
                R[r, om:om + 3] += 0.37 * r
                if r >= N // 2:                           # a rank of the second half holds mode-2 samples only
                    tab = R[r, om + 4:om + 4 + 96 * 3].view(96, 3)
                    tab[:, 2] = tab[:, 1]
                    tab[:, 1] = 0.0
                    R[r, om + 2] = R[r, om + 1]
                    R[r, om + 1] = float("inf")
            if mix == 3:   # second records of the other ranks: copies of rank 0's with their own (never winning) best keys
                RB = e.buffer(L.BUF_RECORDS_B_ALL)
                for r in range(1, N):
                    RB[r].copy_(RB[0])
                    RB[r, 0:6:2] += 0.5          # -w of the best samples: worse than rank 0's
                    RB[r, 6:] = 0.0              # no weight on the other ranks' samples (the plan follows rank 0's)
            for r, q in enumerate(peers, start=1):
                q.buffer(L.BUF_RECORD).copy_(R[r])
            torch.cuda.synchronize()
            if p2p:
                pl.collective = switch_to_p2p
        else:
            J = e.buffer(L.BUF_TRAJ_COST_ALL)
            for r in range(1, N):
                J[r * K:(r + 1) * K] = J[:K] + 0.37 * r
        ms = loop(pl, state, steps)
        e.enable_timing(True)
        pl.collective_times = []
        tr, tu, tf = [], [], []
        for _ in range(100):
            pl.command(state)
            t = e.timing()
            tr.append(t.rollout_ms); tu.append(t.update_ms); tf.append(t.finalize_ms)
        e.enable_timing(False)
        torch.cuda.synchronize()
        per = {}
        for ph, e0, e1 in pl.collective_times:
            per.setdefault(ph, []).append(e0.elapsed_time(e1))
        pl.collective_times = None
        info = e.info()
        res[label] = {"ms_per_command": ms, "rollout_ms": float(np.mean(tr)),
                      "between_rollout_and_collective_ms": float(np.mean(tu)),
                      "collective_to_end_ms": float(np.mean(tf)),
                      "collective_stream_ms": {ph: float(np.mean(v)) for ph, v in per.items()},
                      "collectives_per_command": len(per), "eta": [info.eta, info.eta_1, info.eta_2],
                      "iters": [info.iters, info.iters_1, info.iters_2]}
        if p2p:
            missing, kind = e.p2p_status()
            res[label]["p2p"] = {"missing_rank": missing, "memory_kind": {1: "uncached", 2: "fine-grained", 3: "device"}[kind]}
            for q in peers:
                q.close()
        e.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--emulate-rank-of", type=int, default=0)
    ap.add_argument("--config", default="c5")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--json", default=None)
    ap.add_argument("--samples-per-gpu", type=int, default=None, help="K per rank of the emulation (default: the config's)")
    ap.add_argument("--only-emulation", action="store_true")
    a = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    from m3p2i_aip_amd.distributed import attach_collectives
    env, task, goal, mm, K, T = bench.CONFIGS[a.config]
    out = {"config": a.config, "K": K, "T": T, "multi_modal": mm, "steps": a.steps, "backend": "nccl (RCCL), world_size 1"}
    if a.only_emulation:
        out["K"] = a.samples_per_gpu or K
        out[f"rank0_of_{a.emulate_rank_of}"] = emulate_rank(a.config, a.emulate_rank_of, a.steps, a.samples_per_gpu)
        print(json.dumps(out))
        if a.json:
            os.makedirs(os.path.dirname(a.json) or ".", exist_ok=True)
            json.dump(out, open(a.json, "w"), indent=1)
        dist.destroy_process_group()
        return
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
    if a.emulate_rank_of > 1:
        out[f"rank0_of_{a.emulate_rank_of}"] = emulate_rank(a.config, a.emulate_rank_of, a.steps, a.samples_per_gpu)
    print(json.dumps(out))
    if a.json:
        os.makedirs(os.path.dirname(a.json) or ".", exist_ok=True)
        json.dump(out, open(a.json, "w"), indent=1)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
