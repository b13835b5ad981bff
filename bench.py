#!/usr/bin/env python
"""bench.py -- the MPPI/M3P2I command() hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config push|hybrid|northstar|panda|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full MPPI iteration (M3P2I.command(): rollout of every sample over the
horizon through the contact dynamics + per-step task cost, softmin weights / multi-modal beta
search, mean update, top-k, filter) on synthetic input: the reference's initial scene.

Workload
  N = 1   BASELINE.json configs[1], the config the metric is quoted on: task=push goal=[-1,-1],
          K=2000 samples, T=30 horizon, single-mode, halton-spline noise.  After the headline loop
          the same process also times the other BASELINE configs that fit one GPU and the north-star
          operating point (`other_configs`: hybrid C3, panda C4, northstar K=10000, c5shard = one
          rank's share of C5 run unsharded) and the headline config in CLOSED loop (`closed_loop`:
          a 1-env world stepped between commands, reference flow of scripts/sim.py).
  N > 1   BASELINE.json configs[4] (C5): task=push_pull multi_modal, 8000 samples per GPU (K = 8000 N:
          64000 at N = 8), T=30, samples sharded over the ranks, collectives over RCCL.  Weak scaling:
          the per-GPU sample count is fixed as N grows.  `scaling_reference` is the SAME per-GPU
          workload on one GPU (unsharded handle on rank 0, same process), so that the line carries
          its own 1-GPU point; `collective_ms` is the time per command spent in the collectives.
          (--config overrides the workload for any N.)

Prints ONE JSON line (rank 0).  `value` = K_global*T*steps / wall time (state-steps/s) with
inputs resident in HBM.  `roofline` is for the dominant kernel (the fused rollout kernel):
algorithmic bytes per launch (36 B per state-step for the point env, 92 B for the panda env,
DESIGN.md section 6) / its average duration measured with HIP events on the launch stream.
`cpu_baseline` times the CPU oracle (kind "port": a C/OpenMP port of the same algorithm, oracle/)
on this box's host cores on a bounded sample of the same workload, and the same workload in the
reference's loop shape (`reference_shaped`: per-t Python loop of per-op torch-CPU tensors around a
batched simulator step, oracle/refshaped.py -- SURVEY.md section 8(d), BASELINE.md section 3 "B2").
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
    # BASELINE configs[0], the reference's own CPU-runnable case: navigation, K = 100, T = 10 -- too short for the
    # Halton spline (T >= 12), so mppi_mode 'simple' with in-kernel random noise
    "c1": ("point_env", "navigation", (-3.0, 3.0), False, 100, 10),
}
SIMPLE_MODE = {"c1"}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# rollout kernel, per state-step: delta read (4*nu) + state 16 + action 4*nu + cost 4 written
BYTES_PER_STATE_STEP_ROLLOUT = {"point_env": 36, "panda_env": 92}
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")  # PMC FETCH/WRITE_SIZE per launch


def build_tamp(env, task, goal, multi_modal, K_global, rank, world, T, device, simple=False):
    from m3p2i_aip_amd import isaacgym_wrapper as wrapper
    from m3p2i_aip_amd.cost_functions import Objective
    from m3p2i_aip_amd.planner import M3P2I, MPPIConfig
    if env == "point_env":
        m = MPPIConfig(num_samples=K_global, horizon=T, nx=4, device=device, lambda_=0.5,
                       u_min=[-3.0, -3.0], u_max=[3.0, 3.0], noise_sigma=[[3.0, 0.0], [0.0, 3.0]],
                       u_per_command=T, sample_null_action=True, filter_u=True, fused=True, rank=rank,
                       world_size=world, **(dict(mppi_mode="simple", sampling_method="random") if simple else {}))
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
    return pl, sim, obj, cfg


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
    """Oracle (kind='port') on the host cores + the reference-shaped torch-CPU loop: bounded samples
    of the same workload.  The ONLY place of this file that touches oracle/."""
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
    for label, threads, budget in (("all", ncpu, 7.0), ("one", 1, 7.0)):
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
    res = {"value": best["value"], "unit": "state-steps/s", "cores": best["threads"],
           "kind": "port", "ms_per_command": best["ms"],
           "single_thread_value": out["one"]["value"], "host_cores_usable": ncpu,
           "host_cores_total": os.cpu_count(),
           "sample": f"{best['calls']} command() calls of the same K={K},T={T} {task} workload "
                     f"(oracle/: C port of planner + dynamics spec, OpenMP over samples)"}
    if env == "point_env":
        try:
            from oracle import refshaped
            O.load().m3o_set_threads(ncpu)
            torch.set_num_threads(ncpu)
            res["reference_shaped"] = refshaped.time_commands(task, goal, multi_modal, K, T, delta, budget_s=7.0,
                                                              threads=ncpu)
        except Exception as e:  # the baseline is a report, never a reason to lose the bench line
            res["reference_shaped"] = {"error": repr(e)}
    return res


def run_config(name, args, world, rank, device, dist, steps, warmup, K_local=None, latency=True, time_collectives=False):
    """Builds the planner for CONFIGS[name] and measures it: `warmup` untimed commands, an event loop
    for the kernel durations, then EXACTLY `steps` commands between barrier + synchronize pairs."""
    env, task, goal, multi_modal, K_cfg, T = CONFIGS[name]
    K_local = K_local or K_cfg
    K_global = K_local * world
    pl, sim, obj, cfg = build_tamp(env, task, goal, multi_modal, K_global, rank, world, T, device,
                                   simple=name in SIMPLE_MODE)
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

    for _ in range(warmup):
        pl.command(state)
    # dominant kernel: average launch duration from HIP events recorded by the library on the
    # stream it launches on, over min(steps, 50) commands of the same workload run between the
    # warm-up and the timed region (reading the events back synchronises, so they are not read
    # inside it)
    eng.enable_timing(True)
    pl.collective_times = [] if (time_collectives and world > 1) else None
    tr, tu, tf = [], [], []
    for _ in range(min(steps, 50)):
        pl.command(state)
        t = eng.timing()
        tr.append(t.rollout_ms)
        tu.append(t.update_ms)
        tf.append(t.finalize_ms)
    eng.enable_timing(False)
    coll_ms = None
    if pl.collective_times is not None:
        torch.cuda.synchronize()
        per_phase = {}
        for phase, e0, e1 in pl.collective_times:
            per_phase.setdefault(phase, []).append(e0.elapsed_time(e1))
        n_cmd = min(steps, 50)
        coll_ms = {ph: float(np.sum(v)) / n_cmd for ph, v in per_phase.items()}
        coll_ms["total"] = float(sum(coll_ms.values()))
        coll_ms["per_command"] = len(pl.collective_times) / n_cmd
    pl.collective_times = None
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        pl.command(state)
    sync()
    wall = time.perf_counter() - t0
    if dist is not None:
        tw = torch.tensor([wall], device=device, dtype=torch.float64)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    lat_ms = None
    if latency:
        # command() latency: host clock from the call to the [T, nu] plan's first action on the host
        # (SURVEY.md section 8(d)(ii)); outside the timed region, rank-local
        lat = []
        for _ in range(min(steps, 200)):
            t1 = time.perf_counter()
            pl.command(state)[0].cpu()
            lat.append(time.perf_counter() - t1)
        lat_ms = np.asarray(lat) * 1e3
    rollout_ms = float(np.mean(tr))
    # (in-kernel noise: no delta read, 8 B fewer per state-step)
    alg_bytes = (BYTES_PER_STATE_STEP_ROLLOUT[env] - (8 if name in SIMPLE_MODE else 0)) * K_local * T
    achieved = alg_bytes / (rollout_ms * 1e-3) / 1e9
    return dict(pl=pl, sim=sim, cfg=cfg, env=env, task=task, goal=goal, multi_modal=multi_modal, K_local=K_local,
                K_global=K_global, T=T, wall=wall, steps=steps, value=K_global * T * steps / wall,
                ms_per_step=wall / steps * 1e3, rollout_ms=rollout_ms, update_ms=float(np.mean(tu)),
                finalize_ms=float(np.mean(tf)), alg_bytes=alg_bytes, achieved=achieved, lat_ms=lat_ms,
                collective_ms=coll_ms, simple=name in SIMPLE_MODE)


def brief(r):
    """Entry of `other_configs`."""
    out = {"workload": f"{r['env']} task={r['task']} K={r['K_global']} T={r['T']} "
                       f"{'multi-modal' if r['multi_modal'] else 'single-mode'}" + (" simple mode, in-kernel noise" if r.get("simple") else ""),
           "steps": r["steps"], "ms_per_step": r["ms_per_step"], "command_hz": 1e3 / r["ms_per_step"],
           "value": r["value"], "unit": "state-steps/s",
           "kernel_ms": {"rollout": r["rollout_ms"], "update": r["update_ms"], "finalize": r["finalize_ms"]},
           "roofline": {"bound": "hbm", "achieved": r["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": r["achieved"] / HBM_PEAK_GBS, "bytes_per_launch": r["alg_bytes"]}}
    if r["lat_ms"] is not None:
        out["command_latency_ms"] = {"p50": float(np.percentile(r["lat_ms"], 50)), "p99": float(np.percentile(r["lat_ms"], 99))}
    return out


def closed_loop(r, ticks, device):
    """The headline planner in CLOSED loop: a 1-env "real world" (the same integrator at K = 1) is stepped
    with the first action of every plan and its state is what the next command() starts from -- the flow
    of scripts/sim.py:36-52 + reactive_tamp.py:43-60 in one process.  Nothing is read back on the host
    inside the loop (the action stays on the device), so the commands pipeline as in the open-loop run.
    In one process the planner reads the world's state tensors in place (planner.attach(sim=real): the
    fused rollout takes env 0 of the bound tensors), so the state hand-over of reactive_tamp.py:45-48 costs
    nothing; over RPC it is two blobs per tick (tools/closed_loop.py --connect)."""
    from m3p2i_aip_amd import isaacgym_wrapper as wrapper
    from m3p2i_aip_amd.compat import check_and_apply_suction
    pl, sim, cfg = r["pl"], r["sim"], r["cfg"]
    real = wrapper.IsaacGymWrapper(cfg.isaacgym, cfg.env_type, num_envs=1, device=device)
    nu = real.dofs_per_robot
    point = cfg.env_type == "point_env"
    pull = cfg.task in ("pull", "push_pull")

    pl.attach(sim=real)

    def tick(i):
        if point:
            real.update_dyn_obs(i)
        a = pl.command(real._dof_state[0])[0]
        real.set_dof_velocity_target_tensor(a.view(1, nu))
        if pull:
            cfg.suction_active = pl.pull_preference_tensor()   # stays on the device (the reference reads it back: .item())
            check_and_apply_suction(cfg, real, a.view(1, nu))
        real.step()

    for i in range(10):
        tick(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10, 10 + ticks):
        tick(i)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    out = {"ticks": ticks, "ms_per_step": wall / ticks * 1e3, "command_hz": ticks / wall,
           "value": r["K_global"] * r["T"] * ticks / wall, "unit": "state-steps/s",
           "what": "command() + 1-env world step per tick, world state fed back (dyn-obs moving), action kept on "
                   "the device; host clock over the whole loop"}
    if point:
        goal = torch.tensor(list(r["goal"])[:2], device=device)
        who = real.robot_pos[0] if cfg.task == "navigation" else real.get_actor_position_by_name("box")[0, :2]
        out["final_pos_error_m"] = float(torch.norm(who - goal))
        out["sim_time_s"] = (10 + ticks) * cfg.isaacgym.dt
    pl.attach(sim=sim)
    real.stop_sim()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default=None, choices=list(CONFIGS),
                    help="default: push (BASELINE configs[1]) on 1 GPU, c5 (configs[4]) on N > 1")
    ap.add_argument("--samples-per-gpu", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline line only (profiling runs)")
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

    name = args.config or ("c5" if world > 1 else "push")
    r = run_config(name, args, world, rank, device, dist, args.steps, args.warmup, K_local=args.samples_per_gpu,
                   time_collectives=True)
    env, task, goal, multi_modal, K_local, K_global, T = (r[k] for k in ("env", "task", "goal", "multi_modal", "K_local",
                                                                          "K_global", "T"))
    pl = r["pl"]
    extras = not args.no_extras

    if rank == 0:
        traffic = None
        try:  # HBM bytes per launch from the PMC passes of tools/profile_gpu.sh, same workload only
            tj = json.load(open(TRAFFIC_FILE))
            key = f"{name}:K{K_local}:T{T}"
            if key in tj:
                traffic = tj[key]["hbm_bytes_per_launch"]
        except Exception:
            pass
        kern = "k_rollout_point" if env == "point_env" else "k_rollout_panda"
        if world == 1:
            par = "single GPU"
        else:
            par = (f"samples sharded x{world}, " + ("ONE collective per command: all-gather of per-rank records"
                                                    if pl.shard_mix else "all-gather J + all-reduce packed sums")
                   + (" (gloo, shared GPU: test mode)" if share else " (RCCL)"))
        line = {
            "metric": "mppi_state_steps_per_sec (K x T per command())",
            "value": r["value"], "unit": "state-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
            "higher_is_better": True,
            # the per-GPU sample count is fixed as N grows (K_global = K_local * N)
            "scaling": "weak", "vs_baseline": None,   # BASELINE.md holds no published number for this metric
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{env} task={task} goal={list(goal)[:3]} K={K_global} ({K_local}/GPU) T={T} "
                                   f"{'multi-modal' if multi_modal else 'single-mode'} "
                                   f"{'mppi_mode simple, in-kernel noise' if name in SIMPLE_MODE else 'halton-spline'}, "
                                   "initial scene, open loop (fixed world, warm-started plan)",
                       "name": name, "command_hz": args.steps / r["wall"],
                       "command_latency_ms": {"p50": float(np.percentile(r["lat_ms"], 50)),
                                              "p99": float(np.percentile(r["lat_ms"], 99)),
                                              "what": "host clock, command() + action on host, synchronous"},
                       "parallelism": par},
            "roofline": {"bound": "hbm", "achieved": r["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": r["achieved"] / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kern, "kernel_ms": r["rollout_ms"], "bytes_per_launch": r["alg_bytes"],
                         "kernel_ms_from": "HIP events recorded by the library right before / after the "
                                           "launch on its stream (mean over the commands between warm-up and "
                                           "the timed region); the interval includes the ~5-9 us dispatch "
                                           "latency that rocprofv3's kernel duration (profiles/) excludes",
                         "note": "latency-bound at this K (sequential T x substeps x solver passes chain); "
                                 "DESIGN.md section 6"},
            "kernel_ms": {"rollout": r["rollout_ms"], "update": r["update_ms"], "finalize": r["finalize_ms"]},
        }
        if r["collective_ms"] is not None:
            line["collective_ms"] = r["collective_ms"]
    delta_np = pl.delta.contiguous().cpu().numpy() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None

    if world > 1 and extras and name != "push":
        # the N = 1 headline's workload (C2: push, 2000 samples per GPU, single-mode) weak-scaled over the same
        # ranks (one collective per command): value(N) / (N x the N = 1 line's value) is a like-for-like efficiency
        rw = run_config("push", args, world, rank, device, dist, args.steps, args.warmup, latency=False,
                        time_collectives=True)
        if rank == 0:
            b = brief(rw)
            b["n_gpus"], b["collective_ms"] = world, rw["collective_ms"]
            b["workload"] += f" ({rw['K_local']}/GPU, sharded x{world}: BASELINE configs[1] weak-scaled)"
            line["other_configs"] = {"push_weak": b}
        rw["pl"]._engine.close()
    if world > 1 and extras and not share:
        # the same per-GPU workload on ONE GPU, unsharded (rank 0, after the distributed run; the other
        # ranks wait at the barrier below)
        if rank == 0:
            r1 = run_config(name, args, 1, 0, device, None, min(args.steps, 200), args.warmup, K_local=K_local, latency=False)
            line["scaling_reference"] = {"n_gpus": 1, "value": r1["value"], "ms_per_step": r1["ms_per_step"],
                                         "kernel_ms": {"rollout": r1["rollout_ms"], "update": r1["update_ms"],
                                                       "finalize": r1["finalize_ms"]},
                                         "what": f"same per-GPU workload ({K_local} samples, unsharded handle) on rank 0's GPU"}
        torch.cuda.synchronize()
        dist.barrier()
    if world == 1 and rank == 0 and extras:
        line["closed_loop"] = closed_loop(r, min(args.steps, 200), device)
        others = {}
        for oname, key in (("northstar", "northstar"), ("hybrid", "hybrid"), ("panda", "panda"), ("c5", "c5shard"), ("c1", "c1")):
            if oname == name:
                continue
            try:
                ro = run_config(oname, args, 1, 0, device, None, max(200, min(args.steps, 400)), args.warmup)
                others[key] = brief(ro)
                if oname == "hybrid":
                    others[key]["closed_loop"] = closed_loop(ro, 200, device)
                ro["pl"]._engine.close()
            except Exception as e:
                others[key] = {"error": repr(e)}
        line["other_configs"] = others
    if rank == 0:
        if delta_np is not None:
            line["cpu_baseline"] = cpu_baseline(env, task, goal, multi_modal, K_local, T, delta_np)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
