#!/usr/bin/env python
"""bench.py -- the MPPI/M3P2I command() hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config push|hybrid|northstar|panda]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full MPPI iteration (M3P2I.command(): rollout of every sample over the
horizon through the contact dynamics + per-step task cost, softmin weights, mean update,
top-k, filter) on synthetic input: the reference's initial scene.

Workload at N=1 (BASELINE.json configs[1], the config the metric is quoted on):
task=push goal=[-1,-1], K=2000 samples, T=30 horizon, single-mode, halton-spline noise.
For N>1 every rank keeps the per-GPU sample count (weak scaling, K_global = K*N); the ranks
exchange the K_global trajectory costs (all-gather) and one packed buffer of weighted sums
(all-reduce) per step over RCCL.

Prints ONE JSON line (rank 0).  `value` = K_global*T*steps / wall time (state-steps/s) with
inputs resident in HBM.  `roofline` is for the dominant kernel (the fused rollout kernel):
algorithmic bytes per launch (36 B per state-step for the point env, 92 B for the panda env,
DESIGN.md section 6) / its average duration measured with HIP events on the launch stream.
`cpu_baseline` times the CPU oracle (a C port of the same algorithm, oracle/) on this box's
host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

CONFIGS = {
    # name: (env, task, goal, multi_modal, K per GPU, T)
    "push": ("point_env", "push", (-1.0, -1.0), False, 2000, 30),          # BASELINE configs[1]
    "hybrid": ("point_env", "push_pull", (-3.75, -3.75), True, 4000, 30),  # BASELINE configs[2]
    "panda": ("panda_env", "reach", (0.0,) * 7, False, 4000, 20),          # BASELINE configs[3]
    "northstar": ("point_env", "push", (-1.0, -1.0), False, 10000, 30),    # north_star target point
    "c5": ("point_env", "push_pull", (-3.75, -3.75), True, 8000, 30),      # BASELINE configs[4] = 8 x this (--gpus 8)
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# rollout kernel, per state-step: delta read (4*nu) + state 16 + action 4*nu + cost 4 written
BYTES_PER_STATE_STEP_ROLLOUT = {"point_env": 36, "panda_env": 92}
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")  # PMC FETCH/WRITE_SIZE per launch


def build_tamp(env, task, goal, multi_modal, K_global, rank, world, T, device):
    from m3p2i_aip_amd import isaacgym_wrapper as wrapper
    from m3p2i_aip_amd.cost_functions import Objective
    from m3p2i_aip_amd.planner import M3P2I, MPPIConfig
    if env == "point_env":
        m = MPPIConfig(num_samples=K_global, horizon=T, nx=4, device=device, lambda_=0.5,
                       u_min=[-3.0, -3.0], u_max=[3.0, 3.0], noise_sigma=[[3.0, 0.0], [0.0, 3.0]],
                       u_per_command=T, sample_null_action=True, filter_u=True, fused=True, rank=rank,
                       world_size=world)
        dt = 0.05
    else:
        sig = [[0.0] * 9 for _ in range(9)]
        for i in range(7):
            sig[i][i] = 10.0
        sig[7][7] = sig[8][8] = 0.8
        m = MPPIConfig(num_samples=K_global, horizon=T, nx=18, device=device, lambda_=0.05,
                       u_min=[-2.0] * 7 + [-1.5] * 2, u_max=[2.0] * 7 + [1.5] * 2, noise_sigma=sig,
                       u_per_command=T, sample_null_action=True, filter_u=True, fused=True, rank=rank,
                       world_size=world)
        dt = 0.01
    cfg = SimpleNamespace(env_type=env, multi_modal=multi_modal, suction_active=True, kp_suction=400,
                          pre_height_diff=0.05 if env == "panda_env" else 0.0, task=task, goal=list(goal),
                          cube_on_shelf=False, mppi=m, isaacgym=wrapper.IsaacGymConfig(dt=dt))
    # the wrapper only supplies the world state (env 0) to the fused planner: 64 envs suffice
    sim = wrapper.IsaacGymWrapper(cfg.isaacgym, cfg.env_type, num_envs=64, device=device)
    obj = Objective(cfg)
    obj.update_objective(task, list(goal))
    pl = M3P2I(cfg).attach(sim, obj)
    pl.update_gripper_command(task)
    return pl, sim, obj


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(env, task, goal, multi_modal, K, T, delta):
    """Oracle (kind='port') on the host cores: bounded sample of the same workload."""
    import oracle as O
    O.load()
    if env == "point_env":
        cfg = O.make_cfg(K, T, 2, task=task, goal=goal, multi_modal=multi_modal)
        w0 = O.init_world(1)[0]
        make = lambda: O.OraclePointPlanner(cfg, delta)
    else:
        import oracle.panda as P
        cfg = P.make_cfg(K, T, multi_modal=multi_modal, task=task, goal=goal)
        w0 = P.init_world(1)[0]
        make = lambda: P.OraclePandaPlanner(cfg, delta)
    out = {}
    ncpu = usable_cores()
    for label, threads, budget in (("all", ncpu, 8.0), ("one", 1, 8.0)):
        O.load().m3o_set_threads(threads)
        pl = make()
        pl.command(w0)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget and n < 2000:
            pl.command(w0)
            n += 1
        dt = time.perf_counter() - t0
        out[label] = dict(threads=threads, calls=n, ms=dt / n * 1e3, value=K * T * n / dt)
    best = max(out.values(), key=lambda r: r["value"])
    return {"value": best["value"], "unit": "state-steps/s", "cores": best["threads"],
            "kind": "port", "ms_per_command": best["ms"],
            "single_thread_value": out["one"]["value"], "host_cores_usable": ncpu,
            "host_cores_total": os.cpu_count(),
            "sample": f"{best['calls']} command() calls of the same K={K},T={T} {task} workload "
                      f"(oracle/: C port of planner + dynamics spec, OpenMP over samples)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="push", choices=list(CONFIGS))
    ap.add_argument("--samples-per-gpu", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with python -m torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # M3_BENCH_SHARE_GPU=1 (tests only, tests/test_bench_two_ranks_gpu.py): all ranks on cuda:0 with
    # gloo collectives, so that the N > 1 code path of this file can be exercised on a 1-GPU box
    # (RCCL refuses two ranks on one device).  The numbers of such a run mean nothing.
    share = os.environ.get("M3_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(device))

    env, task, goal, multi_modal, K_local, T = CONFIGS[args.config]
    nu = 2 if env == "point_env" else 9
    if args.samples_per_gpu:
        K_local = args.samples_per_gpu
    K_global = K_local * world

    pl, sim, obj = build_tamp(env, task, goal, multi_modal, K_global, rank, world, T, device)
    # synthetic noise: the reference's Halton-spline sampler for this rank's rows of the global
    # sample set, generated by the planner on its first command() (device sampler; init only, not
    # the hot path).  NOT tiled -- duplicated samples would make the reference's beta search
    # non-terminating: eta >= number of copies of the best sample.
    if world > 1:
        from m3p2i_aip_amd.distributed import attach_collectives
        attach_collectives(pl)
    eng = pl._engine
    state = sim._dof_state[0]

    def sync():
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        pl.command(state)
    # dominant kernel: average launch duration from HIP events recorded by the library on the
    # stream it launches on, over min(steps, 50) commands of the same workload run between the
    # warm-up and the timed region (reading the events back synchronises, so they are not read
    # inside it)
    eng.enable_timing(True)
    tr, tu, tf = [], [], []
    for _ in range(min(args.steps, 50)):
        pl.command(state)
        t = eng.timing()
        tr.append(t.rollout_ms)
        tu.append(t.update_ms)
        tf.append(t.finalize_ms)
    eng.enable_timing(False)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pl.command(state)
    sync()
    wall = time.perf_counter() - t0
    if dist is not None:
        tw = torch.tensor([wall], device=device, dtype=torch.float64)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())

    # command() latency: host clock from the call to the [T, nu] plan's first action on the host
    # (SURVEY.md section 8(d)(ii)); outside the timed region, rank-local
    lat = []
    for _ in range(min(args.steps, 200)):
        t1 = time.perf_counter()
        pl.command(state)[0].cpu()
        lat.append(time.perf_counter() - t1)
    lat_ms = np.asarray(lat) * 1e3

    rollout_ms = float(np.mean(tr))
    alg_bytes = BYTES_PER_STATE_STEP_ROLLOUT[env] * K_local * T
    achieved = alg_bytes / (rollout_ms * 1e-3) / 1e9

    if rank == 0:
        traffic = None
        try:  # HBM bytes per launch from the PMC passes of tools/profile_gpu.sh, same workload only
            tj = json.load(open(TRAFFIC_FILE))
            key = f"{args.config}:K{K_local}:T{T}"
            if key in tj:
                traffic = tj[key]["hbm_bytes_per_launch"]
        except Exception:
            pass
        value = K_global * T * args.steps / wall
        kern = "k_rollout_point" if env == "point_env" else "k_rollout_panda"
        line = {
            "metric": "mppi_state_steps_per_sec (K x T per command())",
            "value": value, "unit": "state-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{env} task={task} goal={list(goal)[:3]} K={K_global} ({K_local}/GPU) T={T} "
                                   f"{'multi-modal' if multi_modal else 'single-mode'} halton-spline, "
                                   "initial scene, open loop (fixed world, warm-started plan)",
                       "name": args.config, "command_hz": args.steps / wall,
                       "command_latency_ms": {"p50": float(np.percentile(lat_ms, 50)),
                                              "p99": float(np.percentile(lat_ms, 99)),
                                              "what": "host clock, command() + action on host, synchronous"},
                       "parallelism": (f"samples sharded x{world}, " + ("one all-gather of per-rank softmin records"
                                                                             if pl.shard_mix else
                                                                             "all-gather J + all-reduce packed sums")
                                       + " (RCCL)") if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kern, "kernel_ms": rollout_ms, "bytes_per_launch": alg_bytes,
                         "kernel_ms_from": "HIP events recorded by the library right before / after the "
                                           "launch on its stream (mean over the commands between warm-up and "
                                           "the timed region); the interval includes the ~5-9 us dispatch "
                                           "latency that rocprofv3's kernel duration (profiles/) excludes",
                         "note": "latency-bound at this K (sequential T x substeps x solver passes chain); "
                                 "DESIGN.md section 6"},
            "kernel_ms": {"rollout": rollout_ms, "update": float(np.mean(tu)), "finalize": float(np.mean(tf))},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(env, task, goal, multi_modal, K_local, T,
                                                  pl.delta.contiguous().cpu().numpy())
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
