"""Generate tests/golden/ref_golden_full.npz: command() traces of the REFERENCE's own planner at BASELINE.json's full
sizes (missing #3: until round 5 the reference-generated traces stopped at K = 256 and the BASELINE sizes
were pinned through the oracle only).

    python tests/golden/make_golden_full.py          (build container only: imports /root/reference)

Three configurations, three closed-loop calls each, the reference's `M3P2I` + `Objective` driven through their plugin
API exactly as `make_golden.py`'s G9 does (reactive_tamp.py:22-73 wiring; the rollouts behind `dynamics()` are the
oracle's worlds -- PhysX is a closed binary --, so what these arrays pin is planner + sampler + cost + update at full K):

    c2  push            K = 2000  T = 30  single mode       BASELINE configs[1]
    c3  push_pull       K = 4000  T = 30  multi_modal       BASELINE configs[2]
    c4  panda reach     K = 4000  T = 20  single mode       BASELINE configs[3] (reach: the task whose cost carries
                                                            quirk Q8 -- every rollout measured against env 0's cube)

Per call: the world handed to command(), the returned action, weights (+ weights_1 / weights_2), mean action(s), the
top-20 indices and trajectories, the pull preference, the adapted beta (single mode) and the pass counts of the three
on-the-fly beta searches (m3p2i.py:24-44: counted by wrapping `update_infinite_beta`).  The Halton-spline noise
`delta` is stored once per configuration (it is an INPUT of the HIP side: `m3_set_noise`).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import make_golden as MG  # noqa: E402  (imports the reference; generates nothing at import)

ref, refshim, O = MG.ref, MG.refshim, MG.O
out = {}


def count_search_passes(pl):
    """Wrap the planner's update_infinite_beta so that every call appends its number of passes (= evaluations of
    torch.exp over the costs) to the returned list."""
    passes = []
    inner = pl.update_infinite_beta

    def counted(costs, beta, ub, lb):
        n = [0]
        real_exp = torch.exp

        def counting_exp(x):
            n[0] += 1
            return real_exp(x)

        ref.m3p2i.torch.exp = counting_exp
        try:
            r = inner(costs, beta, ub, lb)
        finally:
            ref.m3p2i.torch.exp = real_exp
        passes.append(n[0])
        return r

    pl.update_infinite_beta = counted
    return passes


def record(tag, pl, rec, multi_modal):
    for k, v in rec.items():
        out[f"full_{tag}_{k}"] = np.stack(v) if isinstance(v[0], np.ndarray) else np.array(v)
    out[f"full_{tag}_delta"] = pl.delta.numpy().copy()


def point_trace(tag, cfg, ncalls, world0):
    K, T = cfg.mppi.num_samples, cfg.mppi.horizon
    sim = refshim.OracleSim(K, world0)
    real = O.init_world(1)
    real[0] = np.array(world0, np.float32)
    obj = ref.cost_functions.Objective(cfg)
    obj.update_objective(cfg.task, cfg.goal)

    def dynamics(_, u, t=None):
        sim.set_dof_velocity_target_tensor(u)
        sim.step()
        return torch.stack([sim.robot_pos[:, 0], sim.robot_vel[:, 0], sim.robot_pos[:, 1], sim.robot_vel[:, 1]], dim=1), u

    pl = MG.make_planner(cfg, dynamics=dynamics, running_cost=lambda _: obj.compute_cost(sim))
    pl.delta = MG.halton_delta(K, T, 2)
    passes = count_search_passes(pl)
    sc = O.default_scene()
    keys = ["world", "action", "weights", "mean", "top_idx", "top_trajs", "pref"] + \
           (["weights_1", "weights_2", "mean_1", "mean_2", "best_idx", "iters"] if cfg.multi_modal else ["J", "beta"])
    rec = {k: [] for k in keys}
    for call in range(ncalls):
        rec["world"].append(real[0].copy())
        sim.reset(real[0])
        del passes[:]
        a = pl.command(sim._dof_state[0])
        rec["action"].append(a.numpy().copy())
        rec["weights"].append(pl.weights.numpy().copy())
        rec["mean"].append(pl.mean_action.numpy().copy())
        rec["top_idx"].append(pl.top_idx.numpy().astype(np.int32))
        rec["top_trajs"].append(pl.top_trajs.numpy().copy())
        rec["pref"].append(np.int32(pl.get_pull_preference()))
        if cfg.multi_modal:
            rec["weights_1"].append(pl.weights_1.numpy().copy())
            rec["weights_2"].append(pl.weights_2.numpy().copy())
            rec["mean_1"].append(pl.mean_action_1.numpy().copy())
            rec["mean_2"].append(pl.mean_action_2.numpy().copy())
            rec["best_idx"].append(np.array([int(pl.best_idx_1), int(pl.best_idx_2)], np.int32))
            assert len(passes) == 3
            rec["iters"].append(np.array(passes, np.int32))       # order: first half, second half, all (m3p2i.py:58-60)
        else:
            rec["J"].append(pl.total_costs.numpy().copy())
            rec["beta"].append(np.float32(pl.beta))
        O.step_batch(sc, real, a[0:1].numpy())
    record(tag, pl, rec, cfg.multi_modal)


def panda_trace(tag, K, T, task, world0, goal7, ncalls):
    import oracle.panda as P
    cfg = MG.panda_cfg(K, T, multi_modal=False)
    sim = refshim.OraclePandaSim(K, world0)
    obj = ref.cost_functions.Objective(cfg)
    obj.update_objective(task, torch.from_numpy(np.array(goal7, np.float32)))

    def dynamics(_, u, t=None):
        sim.set_dof_velocity_target_tensor(u)
        sim.step()
        return torch.stack([sim.robot_pos[:, 0], sim.robot_vel[:, 0], sim.robot_pos[:, 1], sim.robot_vel[:, 1]], dim=1), u

    pl = MG.make_planner(cfg, dynamics=dynamics, running_cost=lambda _: obj.compute_cost(sim))
    pl.update_gripper_command(task)
    pl.delta = MG.halton_delta(K, T, 9)
    sc = P.default_scene()
    real = np.array(world0, np.float32).reshape(1, -1).copy()
    rec = {k: [] for k in ("world", "action", "weights", "mean", "top_idx", "top_trajs", "pref", "J", "beta")}
    for call in range(ncalls):
        rec["world"].append(real[0].copy())
        sim.reset(real[0])
        a = pl.command(sim._dof_state[0])
        rec["action"].append(a.numpy().copy())
        rec["weights"].append(pl.weights.numpy().copy())
        rec["mean"].append(pl.mean_action.numpy().copy())
        rec["top_idx"].append(pl.top_idx.numpy().astype(np.int32))
        rec["top_trajs"].append(pl.top_trajs.numpy().copy())
        rec["pref"].append(np.int32(pl.get_pull_preference()))
        rec["J"].append(pl.total_costs.numpy().copy())
        rec["beta"].append(np.float32(pl.beta))
        u1 = np.zeros((1, 9), np.float32)
        u1[0] = a[0].numpy()
        P.step_batch(sc, real, u1)
    record(tag, pl, rec, False)


if __name__ == "__main__":
    import time
    import oracle.panda as P
    t0 = time.time()
    w0 = O.init_world(1)[0]
    point_trace("c2", MG.point_cfg(2000, 30, task="push", goal=(-1.0, -1.0)), 3, w0)
    print("c2 ok %.1f s" % (time.time() - t0))
    # C3: the robot within suction range of the box, as G9's `hybrid` -- both modes have something to do from call 0
    w2 = w0.copy()
    w2[0], w2[1] = 0.0, 1.5
    point_trace("c3", MG.point_cfg(4000, 30, multi_modal=True, task="push_pull", goal=(-3.75, -3.75)), 3, w2)
    print("c3 ok %.1f s" % (time.time() - t0))
    goal7 = [0.2, 0.2, 1.115, 0.0, 0.0, 0.0, 1.0]
    panda_trace("c4", 4000, 20, "reach", P.init_world(1)[0], goal7, 3)
    print("c4 ok %.1f s" % (time.time() - t0))
    path = os.path.join(HERE, "ref_golden_full.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")
