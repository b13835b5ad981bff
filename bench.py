#!/usr/bin/env python
"""bench.py -- the MPPI/M3P2I command() hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config push|hybrid|northstar|panda|c5|...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full MPPI iteration (M3P2I.command(): rollout of every sample over the
horizon through the contact dynamics + per-step task cost, softmin weights / multi-modal beta
search, mean update, top-k, filter) on synthetic input: the reference's initial scene.

Workload
  N = 1   BASELINE.json configs[1], the config the metric is quoted on: task=push goal=[-1,-1],
          K=2000 samples, T=30 horizon, single-mode, halton-spline noise.  After the headline loop
          the same process also times the other BASELINE configs that fit one GPU and the north-star
          operating point (`other_configs`: hybrid C3, panda C4 in its reach phase AND in its pick phase
          (`panda_pick`: the scene the product's own closed-loop reach ends in, cube held, FORCES kernel
          instance), northstar K=10000, c5shard = one rank's share of C5 run unsharded, c5_unsharded = ALL
          of C5 (K=64000 multi-modal) on this ONE GPU -- the honest denominator of any sharding claim --,
          worst_case_scene = the headline workload with robot, box and dyn-obs packed into the wall corner
          (every substep in the all-contact-slots instance), c1) and the headline config in CLOSED loop
          (`closed_loop`: a 1-env world stepped between commands, reference flow of scripts/sim.py).
  N > 1   the SAME per-GPU workload as N = 1 -- push, 2000 samples per GPU (K = 2000 N), single-mode --
          sharded over the ranks with ONE collective per command, so that value(N) / (N value(1)) is a
          like-for-like weak-scaling efficiency (`scaling`: "weak").  `other_configs` carries, each row with
          its own `scaling` and 1-GPU references measured in the same run on rank 0:
            c5               BASELINE configs[4]: push_pull multi-modal, 8000 samples per GPU (64000 at N = 8),
                             weak; `scaling_reference` = the same 8000 samples unsharded on one GPU,
                             `strong_scaling_reference` = ALL K_global samples unsharded on one GPU (at every
                             BASELINE size one MI355X is latency-bound: it runs the whole of C5 in ~0.2 ms, so
                             sharding C5 cannot be faster than not sharding it -- DESIGN.md section 7)
            push_saturating  single-mode push with 131072 samples per GPU (K = 2^17 N): the smallest per-GPU
                             size at which the rollout kernel is throughput-bound (2 waves per SIMD), i.e.
                             where adding GPUs can pay; weak, with its own 1-GPU reference
          `collective_ms` is the time per command spent in the collectives.
          (--config overrides the headline workload for any N.)

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
    # C4 as BASELINE names it, reactive PICK: same planner, task=pick (gripper override close, pick + motion cost on
    # the penalty forces: the <FORCES=true> kernel instance); scene + goal come from panda_pick_scene()
    "panda_pick": ("panda_env", "pick", (0.0,) * 7, False, 4000, 20),
    "c5_unsharded": ("point_env", "push_pull", (-3.75, -3.75), True, 64000, 30),   # ALL of C5 on one GPU
    "worst_case": ("point_env", "push", (-1.0, -1.0), False, 2000, 30),            # C2 in the corner scene below
    "push_sat": ("point_env", "push", (-1.0, -1.0), False, 131072, 30),            # saturating per-GPU size
    # the reference's SHIPPED planner size (config/mppi/point.yaml: 200 samples, horizon 15) on the scenario of its logged
    # runs (push to (-3, 3)): the only size its published planner rates (21.2 Hz) can have been taken at
    "refsize": ("point_env", "push", (-3.0, 3.0), False, 200, 15),
}
# per-GPU sample count fixed as N grows for every workload of this file (K_global = K_local * N)
SCALING = "weak"
SIMPLE_MODE = {"c1"}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# rollout kernel, per state-step: delta read (4*nu) + state 16 + action 4*nu + cost 4 written
BYTES_PER_STATE_STEP_ROLLOUT = {"point_env": 36, "panda_env": 92}
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")  # PMC FETCH/WRITE_SIZE per launch


def build_tamp(env, task, goal, multi_modal, K_global, rank, world, T, device, simple=False, shard_mix=None):
    from m3p2i_aip_amd import isaacgym_wrapper as wrapper
    from m3p2i_aip_amd.cost_functions import Objective
    from m3p2i_aip_amd.planner import M3P2I, MPPIConfig
    if env == "point_env":
        m = MPPIConfig(num_samples=K_global, horizon=T, nx=4, device=device, lambda_=0.5,
                       u_min=[-3.0, -3.0], u_max=[3.0, 3.0], noise_sigma=[[3.0, 0.0], [0.0, 3.0]],
                       u_per_command=T, sample_null_action=True, filter_u=True, fused=True, rank=rank,
                       world_size=world, shard_mix=shard_mix if multi_modal else None,
                       **(dict(mppi_mode="simple", sampling_method="random") if simple else {}))
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


def corner_scene(pl, sim, obj, cfg):
    """Worst case of the point_env rollout: box in the wall corner (touching both walls), dyn-obs beside it against
    the wall and the box, the robot on top of the box against the other wall -- robot-box, robot-dyn-obs, robot-walls, box-walls, dyn-obs-walls
    and box-dyn-obs pairs are all inside their broad-phase ranges, so every substep of every wavefront runs the
    instance with all 19 contact slots."""
    from m3p2i_aip_amd import scenes
    ib, idn = scenes.actor_index("point_env", "box"), scenes.actor_index("point_env", "dyn-obs")
    # walls' inner faces at +-3.95, boxes 0.4 x 0.4, robot radius 0.2; 5 mm gaps (contact_offset is 10 mm)
    sim._root_state[:, ib, 0] = -3.745
    sim._root_state[:, ib, 1] = -3.745
    sim._root_state[:, idn, 0] = -3.34
    sim._root_state[:, idn, 1] = -3.745
    sim._dof_state[:] = torch.tensor([-3.745, 0.0, -3.34, 0.0], device=sim._dof_state.device)
    sim.set_dof_state_tensor(sim._dof_state)
    sim.set_actor_root_state_tensor(sim._root_state)


def settled_panda_scene(pl, sim, obj, cfg):
    """The configured panda_env scene after the cubes have landed: the yaml files start cubeA and cubeB 1 cm above the
    table (5_cubeA.yaml, 6_cubeB.yaml), so in the INITIAL scene every rollout first simulates two cubes falling and
    settling (awake free bodies: corner contacts in every pass) -- true of the first commands of an episode only.  Thirty
    ticks of the zero action in the 1-env sense (here: every env of the wrapper) put them to sleep on the table."""
    z = torch.zeros(sim.num_envs, 9, device=sim._dof_state.device)
    for _ in range(30):
        sim.set_dof_velocity_target_tensor(z)
        sim.step()
    sim.set_dof_state_tensor(sim._dof_state)
    sim.set_actor_root_state_tensor(sim._root_state)


def panda_pick_scene(device):
    """The scene C4's pick phase starts in, produced by the PRODUCT: the closed loop of tools/closed_loop.py
    (1-env world + planner K=4000, T=20 + the active-inference task planner) is run through its reach phase
    and 12 ticks into `pick` -- gripper closed on the cube, cube held and on its way to the goal."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import closed_loop
    res = closed_loop.run("config_panda", ["mppi.num_samples=4000", "mppi.horizon=20", f"mppi.device={device}"],
                          ticks=400, until_task="pick", extra_ticks=12)
    cap = res.get("captured")
    if cap is None:
        raise RuntimeError(f"the closed loop never reached the pick phase: {res.get('timeline')}")
    return cap


def panda_reach_mid_scene(device, tick=40):
    """The world `tick` ticks into the reach phase of the product's own closed loop (the pick starts at tick ~66): the gripper on
    its way down to cubeA, rollouts next to the cube -- what most reach commands of an episode look like, unlike the first one."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import closed_loop
    res = closed_loop.run("config_panda", ["mppi.num_samples=4000", "mppi.horizon=20", f"mppi.device={device}"],
                          ticks=400, until_task="reach", extra_ticks=tick)
    cap = res.get("captured")
    if cap is None or cap.get("task") != "reach":
        raise RuntimeError(f"no reach scene at tick {tick}: {res.get('timeline')}")

    def scene(pl, sim, obj, cfg):
        dev = sim._dof_state.device
        sim._dof_state[:] = torch.tensor(cap["dof_state"], device=dev)
        sim._root_state[:] = torch.tensor(cap["root_state"], device=dev)
        sim.set_dof_state_tensor(sim._dof_state)
        sim.set_actor_root_state_tensor(sim._root_state)
    return scene


def make_pick_scene(cap):
    def scene(pl, sim, obj, cfg):
        dev = sim._dof_state.device
        sim._dof_state[:] = torch.tensor(cap["dof_state"], device=dev)
        sim._root_state[:] = torch.tensor(cap["root_state"], device=dev)
        sim.set_dof_state_tensor(sim._dof_state)
        sim.set_actor_root_state_tensor(sim._root_state)
        obj.update_objective("pick", [float(x) for x in cap["goal"]])
        pl.update_gripper_command("pick")
    return scene


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


def cpu_baseline(env, task, goal, multi_modal, K, T, delta, seconds=21.0):
    """Oracle (kind='port') on the host cores + the reference-shaped torch-CPU loop: bounded samples
    of the same workload (`seconds` of CPU work in three equal legs: --cpu-baseline-seconds, default 21).  The ONLY
    place of this file that touches oracle/."""
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
    leg = max(0.2, seconds / 3.0)
    for label, threads, budget in (("all", ncpu, leg), ("one", 1, leg)):
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
            res["reference_shaped"] = refshaped.time_commands(task, goal, multi_modal, K, T, delta, budget_s=leg,
                                                              threads=ncpu)
        except Exception as e:  # the baseline is a report, never a reason to lose the bench line
            res["reference_shaped"] = {"error": repr(e)}
    return res


def run_config(name, args, world, rank, device, dist, steps, warmup, K_local=None, latency=True, time_collectives=False,
               scene=None, repeats=1):
    """Builds the planner for CONFIGS[name] and measures it: `warmup` untimed commands, an event loop
    for the kernel durations, then EXACTLY `steps` commands between barrier + synchronize pairs."""
    env, task, goal, multi_modal, K_cfg, T = CONFIGS[name]
    K_local = K_local or K_cfg
    K_global = K_local * world
    pl, sim, obj, cfg = build_tamp(env, task, goal, multi_modal, K_global, rank, world, T, device,
                                   simple=name in SIMPLE_MODE, shard_mix=getattr(args, "shard_mix", None))
    if scene is not None:
        scene(pl, sim, obj, cfg)     # a world other than the reference's initial scene (and its objective)
    # synthetic noise: the reference's Halton-spline sampler for this rank's rows of the global
    # sample set, generated by the planner on its first command() (device sampler; init only, not
    # the hot path).  NOT tiled -- duplicated samples would make the reference's beta search
    # non-terminating: eta >= number of copies of the best sample.
    if world > 1:
        from m3p2i_aip_amd.distributed import attach_collectives, attach_p2p
        # --transport p2p: the records exchange through peer-mapped device memory (csrc/p2p.hip) instead of RCCL
        (attach_p2p if getattr(args, "transport", "rccl") == "p2p" else attach_collectives)(pl)
    eng = pl._engine
    state = sim._dof_state[0]

    def sync():
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(min(warmup, 5)):
        pl.command(state)        # (first calls: sampler + sort of the wave order; the W warm-up commands proper come below)
    # dominant kernel: average launch duration from HIP events recorded by the library on the
    # stream it launches on, over min(steps, 50) commands of the same workload run BEFORE the
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
    import gc
    walls = []
    for _ in range(repeats):     # (the headline: exactly once; `other_configs` rows: the better of two timed regions,
        gc.collect()             #  both reported -- a one-off host hiccup of ~65 ms has been seen inside a 40 ms region;
        gc.disable()             #  the collector is kept out of the region)
        # the W untimed warm-up commands, IMMEDIATELY before the timed ones: a full collection (~30 ms over torch's objects)
        # leaves the interpreter's working set cold, which a 20-command region that started right after it paid for with
        # +6 % (tools: 0.1284 -> 0.1365 ms per command; a 50 ms sleep instead costs 1 %) -- that is what a warm-up is for
        for _ in range(warmup):
            pl.command(state)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            pl.command(state)
        sync()
        walls.append(time.perf_counter() - t0)
        gc.enable()
    wall = min(walls)
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
                collective_ms=coll_ms, simple=name in SIMPLE_MODE, walls_ms_per_step=[w / steps * 1e3 for w in walls])


MIX_DIR = os.path.join(ROOT, "profiles", "r06")


def lib_build_id():
    """m3_build_id() of the HIP library this process runs: a hash of the kernel sources + compiler flags that is the same for
    every rebuild of the same tree (the .so's own sha256 is not) -- the key of the committed PMC instruction-mix profiles"""
    from m3p2i_aip_amd import _lib as L
    return L.load().m3_build_id().decode()


def roofline_valu(name, r, n_waves):
    """The roofline that binds this path -- VALU issue, not HBM: the rollout kernel's dynamic
    instruction counts per wavefront from the committed PMC profile of THIS build (profiles/r06/mix_<config>.json,
    written by tools/pmc_mix_bench.sh; keyed by m3_build_id(): a profile of other sources / flags is refused), against
      lone_wave_issue_frac  = VALU instructions / wave cycles -- how close ONE wavefront alone on its SIMD comes to
                              issuing a VALU instruction every 4 clocks (the bound of every BASELINE size: at most one
                              wavefront per SIMD, so the dependent-instruction chain of a wave is the command's time)
      chip_issue_frac       = (VALU instructions of all waves x 4 clocks) / (kernel duration x 1024 SIMDs): how much of
                              the chip's VALU issue capacity the launch uses (the bound of the saturated regime).
    Both SQ counters are in units of 4 clocks (MI355X_MICROARCH.md)."""
    path = os.path.join(MIX_DIR, f"mix_{name}.json")
    if not os.path.exists(path):
        return None
    mix = json.load(open(path))
    sha = lib_build_id()
    if mix.get("build_id") != sha:
        return {"stale": f"{os.path.relpath(path, ROOT)} profiles build {mix.get('build_id')}, this run uses {sha}: "
                         "re-run tools/pmc_mix_bench.sh"}
    valu, cyc = mix["SQ_INSTS_VALU"], mix["SQ_WAVE_CYCLES"]
    n_waves = int(mix.get("waves") or n_waves)      # (what the launch really had: the Panda kernel shares a sample among 8 / 16 lanes)
    clock_ghz = mix.get("clock_ghz", 2.4)
    kernel_clocks = r["rollout_ms"] * 1e-3 * clock_ghz * 1e9
    return {"bound": "valu_issue", "kernel": mix.get("kernels"), "valu_per_wave": valu, "wave_cycles_x4": cyc,
            "lone_wave_issue_frac": valu / cyc, "waves": n_waves, "simds": 1024,
            "chip_issue_frac": n_waves * valu * 4.0 / (kernel_clocks * 1024),
            "arithmetic_frac_of_valu": (mix.get("SQ_INSTS_VALU_ADD_F32", 0) + mix.get("SQ_INSTS_VALU_MUL_F32", 0) +
                                        mix.get("SQ_INSTS_VALU_FMA_F32", 0) + mix.get("SQ_INSTS_VALU_TRANS_F32", 0)) / valu,
            "clock_ghz_assumed": clock_ghz, "profile": os.path.relpath(path, ROOT), "build_id": sha}


def brief(r, mix_name=None):
    """Entry of `other_configs`."""
    out = {"workload": f"{r['env']} task={r['task']} K={r['K_global']} T={r['T']} "
                       f"{'multi-modal' if r['multi_modal'] else 'single-mode'}" + (" simple mode, in-kernel noise" if r.get("simple") else ""),
           "steps": r["steps"], "ms_per_step": r["ms_per_step"], "command_hz": 1e3 / r["ms_per_step"],
           "value": r["value"], "unit": "state-steps/s", "scaling": SCALING,
           "kernel_ms": {"rollout": r["rollout_ms"], "update": r["update_ms"], "finalize": r["finalize_ms"]},
           "roofline": {"bound": "hbm", "achieved": r["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": r["achieved"] / HBM_PEAK_GBS, "bytes_per_launch": r["alg_bytes"]}}
    pl = r.get("pl")
    if pl is not None and getattr(pl, "world_size", 1) > 1:
        # what the communicator really was (RCCL has only ever been exercised at world_size 1 by this build's own runs:
        # the first multi-GPU run is the driver's) and which transport / protocol carried the exchange
        out["ranks_seen"], out["transport"], out["protocol"] = pl.ranks_seen, pl.transport, pl.protocol
    if mix_name is not None:
        # (samples per wavefront: the Panda kernel's forms share a sample among 8 / 16 lanes; SQ_WAVES of the profile says how many)
        rv = roofline_valu(mix_name, r, (r["K_local"] + 63) // 64)
        if rv is not None:
            out["roofline_valu"] = rv
    if r["lat_ms"] is not None:
        out["command_latency_ms"] = {"p50": float(np.percentile(r["lat_ms"], 50)), "p99": float(np.percentile(r["lat_ms"], 99))}
    if len(r["walls_ms_per_step"]) > 1:
        out["timed_regions_ms_per_step"] = r["walls_ms_per_step"]     # ms_per_step is the smaller
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
    real.zero_copy_targets = True     # the action slot handed over below is not touched again before step()
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


def self_launch(n, argv):
    """`python3 bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks (one process per GPU) under
    torch.distributed.run on a free loopback port -- the command line the driver documents --, pass rank 0's ONE JSON
    line through, and return non-zero if any rank failed, if there is not exactly one line, or if the communicator the
    ranks really formed (`ranks_seen`) is not N."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL, hipIpc)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)     # (stderr passes straight through)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    rest = [ln for ln in p.stdout.splitlines() if not ln.startswith("{")]
    if rest:
        print("\n".join(rest), file=sys.stderr)
    if p.returncode != 0:
        print(f"bench.py: the {n}-rank launch failed with exit code {p.returncode}", file=sys.stderr)
        return p.returncode or 1
    if len(lines) != 1:
        print(f"bench.py: expected ONE JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        return 1
    try:
        d = json.loads(lines[0])
    except ValueError as e:
        print(f"bench.py: rank 0's line is not JSON: {e}", file=sys.stderr)
        return 1
    print(lines[0], flush=True)
    if d.get("n_gpus") != n or d.get("ranks_seen") != n:
        print(f"bench.py: asked for {n} ranks, the line reports n_gpus={d.get('n_gpus')} ranks_seen={d.get('ranks_seen')}",
              file=sys.stderr)
        return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default=None, choices=list(CONFIGS),
                    help="default: push (BASELINE configs[1]), 2000 samples per GPU, for every N")
    ap.add_argument("--samples-per-gpu", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=21.0,
                    help="CPU work of the cpu_baseline object (three equal legs: oracle on all usable cores, on one, the "
                         "reference-shaped torch loop); the contract tests pass a small value")
    ap.add_argument("--shard-mix", type=int, default=None, choices=[1, 2, 3],
                    help="multi-modal sharding protocol (N > 1): 2 = one collective (default), 1 = its bit-identical variant, "
                         "3 = two small exchanges with O(K_local) work per rank")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "p2p"],
                    help="N > 1: the records exchange as an RCCL all-gather (default) or through peer-mapped device memory")
    ap.add_argument("--no-extras", action="store_true", help="headline line only (profiling runs)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started the way the driver starts the N = 1 run (`python3 bench.py --gpus N ...`): launch the ranks ourselves
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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

    name = args.config or "push"
    # (--config panda_pick / worst_case as the headline of a profiling run: their scenes)
    head_scene = make_pick_scene(panda_pick_scene(device)) if name == "panda_pick" else (corner_scene if name == "worst_case" else None)
    r = run_config(name, args, world, rank, device, dist, args.steps, args.warmup, K_local=args.samples_per_gpu,
                   time_collectives=True, scene=head_scene)
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
            par = (f"samples sharded x{world}, " + (("TWO small exchanges per command (shard_mix 3)" if pl._shard_mix_level == 3 else
                                                     "ONE collective per command: all-gather of per-rank records")
                                                    if pl.shard_mix else "all-gather J + all-reduce packed sums")
                   + " (" + ("p2p device-side exchange" if (args.transport == "p2p" and pl.shard_mix) else ("gloo" if share else "RCCL"))
                   + (", shared GPU: test mode)" if share else ")"))
        line = {
            "metric": "mppi_state_steps_per_sec (K x T per command())",
            "value": r["value"], "unit": "state-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
            "higher_is_better": True,
            # the per-GPU sample count is fixed as N grows (K_global = K_local * N); the default headline of an
            # N > 1 run is the N = 1 headline's own per-GPU workload, so the driver's value(N) / (N value(1)) compares like with like
            "scaling": SCALING, "vs_baseline": None,   # BASELINE.md holds no published number for this metric
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
        line["roofline_valu"] = roofline_valu(name, r, (K_local + 63) // 64)
        if world > 1:
            line["ranks_seen"], line["transport"], line["protocol"] = pl.ranks_seen, pl.transport, pl.protocol
        if r["collective_ms"] is not None:
            line["collective_ms"] = r["collective_ms"]
    delta_np = pl.delta.contiguous().cpu().numpy() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None

    def one_gpu_reference(cname, K, what):
        """The named workload with K samples on ONE GPU (unsharded handle, rank 0's GPU, the other ranks wait at
        the caller's barrier): the denominator of a scaling claim, measured in the same run."""
        r1 = run_config(cname, args, 1, 0, device, None, min(args.steps, 200), args.warmup, K_local=K, latency=False)
        out = {"n_gpus": 1, "K": K, "value": r1["value"], "ms_per_step": r1["ms_per_step"],
               "kernel_ms": {"rollout": r1["rollout_ms"], "update": r1["update_ms"], "finalize": r1["finalize_ms"]},
               "what": what}
        r1["pl"]._engine.close()
        return out

    if world > 1 and extras:
        small = args.samples_per_gpu if share else None      # (test mode: every row at the test's size)
        others = {}
        rows = [("c5", "c5", "BASELINE configs[4]: 8000 samples per GPU, push_pull multi-modal"),
                ("push_sat", "push_saturating", "131072 samples per GPU: the rollout kernel's throughput-bound regime")]
        if name != "push":
            rows.insert(0, ("push", "push_weak", "BASELINE configs[1] weak-scaled"))
        def sharded_row(cname, key, what):
            rw = run_config(cname, args, world, rank, device, dist, args.steps, args.warmup, K_local=small,
                            latency=False, time_collectives=True)
            if rank == 0:
                bq = brief(rw)
                bq["n_gpus"], bq["collective_ms"] = world, rw["collective_ms"]
                bq["workload"] += f" ({rw['K_local']}/GPU, sharded x{world}: {what})"
                others[key] = bq
            rw["pl"]._engine.close()
            if not share:
                # 1-GPU references of this row, measured now on rank 0
                if rank == 0:
                    others[key]["scaling_reference"] = one_gpu_reference(
                        cname, rw["K_local"], f"weak-scaling denominator: the same per-GPU workload ({rw['K_local']} samples) unsharded on one GPU")
                    if cname == "c5":
                        others[key]["strong_scaling_reference"] = one_gpu_reference(
                            "c5", rw["K_global"], f"strong-scaling denominator: ALL {rw['K_global']} samples unsharded on ONE GPU "
                            "(below ~65536 samples per GPU the rollout is latency-bound: one MI355X runs this in about the "
                            "time a rank needs for its 1/N share, DESIGN.md section 7)")
                        others[key]["speedup_vs_one_gpu_same_K"] = others[key]["strong_scaling_reference"]["ms_per_step"] / bq["ms_per_step"]

        for cname, key, what in rows:
            if cname == name:
                continue
            # (a row that fails -- on every rank alike: a create-time refusal, memory -- must not take the headline
            # line with it; a failure on one rank only still ends the job through the launcher)
            try:
                sharded_row(cname, key, what)
            except Exception as e:
                if rank == 0:
                    others.setdefault(key, {})["error"] = repr(e)
            if not share:
                torch.cuda.synchronize()
                dist.barrier()
        if rank == 0:
            line["other_configs"] = others
    if world > 1 and extras and not share:
        if rank == 0:
            line["scaling_reference"] = one_gpu_reference(
                name, K_local, f"same per-GPU workload ({K_local} samples, unsharded handle) on rank 0's GPU")
        torch.cuda.synchronize()
        dist.barrier()
    if world == 1 and rank == 0 and extras:
        line["closed_loop"] = closed_loop(r, 200, device)      # (200 ticks whatever --steps says: 20 ticks say nothing)
        others = {}
        pick_scene = None
        try:
            pick_scene = make_pick_scene(panda_pick_scene(device))
        except Exception as e:
            others["panda_pick"] = {"error": repr(e)}
        mid_scene = None
        try:
            mid_scene = panda_reach_mid_scene(device)
        except Exception as e:
            others["panda_reach_mid"] = {"error": repr(e)}
        for oname, key, scene in (("northstar", "northstar", None), ("hybrid", "hybrid", None), ("panda", "panda", None),
                                  ("panda", "panda_settled", settled_panda_scene), ("panda", "panda_reach_mid", mid_scene),
                                  ("panda_pick", "panda_pick", pick_scene), ("c5", "c5shard", None),
                                  ("c5_unsharded", "c5_unsharded", None), ("worst_case", "worst_case_scene", corner_scene),
                                  ("c1", "c1", None), ("refsize", "reference_default_size", None)):
            if oname == name or (oname == "panda_pick" and pick_scene is None) or (key == "panda_reach_mid" and mid_scene is None):
                continue
            try:
                ro = run_config(oname, args, 1, 0, device, None, max(200, min(args.steps, 400)), args.warmup, scene=scene,
                                repeats=2)
                others[key] = brief(ro, mix_name={"northstar": "northstar", "hybrid": "hybrid", "panda": "panda", "panda_settled": None, "panda_reach_mid": None,
                                                  "panda_pick": "panda_pick", "c5shard": "c5", "c5_unsharded": "c5_unsharded",
                                                  "worst_case_scene": "worst_case"}.get(key))
                if oname == "hybrid":
                    others[key]["closed_loop"] = closed_loop(ro, 200, device)
                if key == "panda":
                    others[key]["workload"] += (" -- C4's reach phase in the INITIAL scene: the cubes start 1 cm above the table and "
                                                "every rollout simulates them landing (world spec v2: free bodies)")
                if key == "panda_settled":
                    others[key]["workload"] += (" -- C4's reach phase once the cubes have landed and sleep (every command of an "
                                                "episode but the first few)")
                if key == "panda_reach_mid":
                    e = ro["pl"]._engine
                    others[key]["workload"] += (" -- C4's reach phase 40 ticks into the product's own closed loop (gripper on its way "
                                                "down to the cube): the kernel form is chosen from the share of (sample, substep) pairs "
                                                "the last rollout reported near a box")
                    others[key]["lanes_per_sample_used"] = e.panda_lanes_per_sample_used()
                    others[key]["near_share_permille"] = e.panda_near_share()
                if oname == "panda_pick":
                    others[key]["workload"] += (" -- C4's pick phase: scene = 12 ticks into `pick` of the product's own closed "
                                                "loop (cube held), gripper override close, k_rollout_panda<FORCES=true>")
                if oname == "worst_case":
                    others[key]["workload"] += (" -- robot, box and dyn-obs packed into the wall corner (open loop from that "
                                                "scene): the all-contact-slots substep instance")
                if oname == "refsize":
                    others[key]["workload"] += (" -- the reference's shipped planner size; its own logged planner rate for this "
                                                "scenario is 21.2 Hz (BASELINE.md: Isaac Gym, hardware / K / T of the runs not recorded)")
                if oname == "c5_unsharded":
                    others[key]["workload"] += " -- ALL of BASELINE configs[4] on ONE GPU (the denominator of any sharding claim)"
                ro["pl"]._engine.close()
            except Exception as e:
                others[key] = {"error": repr(e)}
        line["other_configs"] = others
    if rank == 0:
        if delta_np is not None:
            line["cpu_baseline"] = cpu_baseline(env, task, goal, multi_modal, K_local, T, delta_np, seconds=args.cpu_baseline_seconds)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
