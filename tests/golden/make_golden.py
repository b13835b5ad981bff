"""Generate tests/golden/*.npz by importing the REFERENCE's own Python.

Run in the build container only (reference mounted read-only at /root/reference):

    python tests/golden/make_golden.py

The fixtures are data: seeded inputs + the outputs the reference's functions returned.
SURVEY.md section 8(c) lists the groups G1..G10; group names below follow it.  The
reference cannot travel to the GPU box, these files can.  The dynamics behind G9 are the
oracle's (the reference's dynamics are Isaac Gym, a closed binary): G9 pins the reference's
planner + cost code driven through its own plugin API, not PhysX.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import refshim  # noqa: E402
import oracle as O  # noqa: E402

ref = refshim.import_reference()
MPPIConfig = ref.mppi.MPPIConfig
torch.set_num_threads(1)


def point_cfg(K, T, multi_modal=False, task="push", goal=(-1.0, -1.0), mode="halton-spline",
              sampling="halton", filter_u=True, null=True, u_per_command=None):
    # values of config/mppi/point.yaml + config/config_point.yaml, K/T overridden
    m = MPPIConfig(num_samples=K, horizon=T, nx=4, mppi_mode=mode, sampling_method=sampling,
                   device="cpu", lambda_=0.5, u_min=[-3.0, -3.0], u_max=[3.0, 3.0],
                   noise_sigma=[[3.0, 0.0], [0.0, 3.0]],
                   u_per_command=T if u_per_command is None else u_per_command,
                   sample_null_action=null, filter_u=filter_u, use_priors=False)
    return SimpleNamespace(env_type="point_env", multi_modal=multi_modal, suction_active=True,
                           kp_suction=400, pre_height_diff=0.0, task=task, goal=list(goal),
                           mppi=m)


def panda_cfg(K, T, multi_modal=False):
    sig = [[0.0] * 9 for _ in range(9)]
    for i in range(7):
        sig[i][i] = 10.0
    sig[7][7] = sig[8][8] = 0.8
    m = MPPIConfig(num_samples=K, horizon=T, nx=18, mppi_mode="halton-spline",
                   sampling_method="halton", device="cpu", lambda_=0.05,
                   u_min=[-2.0] * 7 + [-1.5] * 2, u_max=[2.0] * 7 + [1.5] * 2, noise_sigma=sig,
                   u_per_command=T, sample_null_action=True, filter_u=True, use_priors=False)
    return SimpleNamespace(env_type="panda_env", multi_modal=multi_modal, suction_active=False,
                           kp_suction=0, pre_height_diff=0.05, task="reach", goal=[0.0] * 7,
                           mppi=m)


def make_planner(cfg, dynamics=None, running_cost=None):
    return ref.m3p2i.M3P2I(cfg, dynamics=dynamics, running_cost=running_cost)


out = {}


# ---------------------------------------------------------------- G1 cost_to_go
def g1():
    rng = np.random.default_rng(101)
    c = rng.uniform(0, 50, (64, 30)).astype(np.float32)
    gamma_seq = torch.cumprod(torch.tensor([1.0] + [0.95] * 29), dim=0).reshape(1, 30)
    ctg = ref.mppi_utils.cost_to_go(torch.from_numpy(c.copy()), gamma_seq)
    out["g1_cost"] = c
    out["g1_ctg"] = ctg.numpy()


# ---------------------------------------------------------------- G2 _exp_util traces
def g2():
    for name, cfg in (("point", point_cfg(128, 30)), ("panda", panda_cfg(128, 20))):
        pl = make_planner(cfg)
        rng = np.random.default_rng(202)
        costs, ws, betas, etas = [], [], [], []
        for call in range(6):
            scale = [40, 5, 0.5, 100, 1.0, 0.05][call]
            c = (rng.uniform(0, scale, (128, cfg.mppi.horizon)) + 10).astype(np.float32)
            pl._exp_util(torch.from_numpy(c.copy()))
            costs.append(c)
            ws.append(pl.weights.numpy().copy())
            betas.append(float(pl.beta))
        out[f"g2_{name}_costs"] = np.stack(costs)
        out[f"g2_{name}_weights"] = np.stack(ws)
        out[f"g2_{name}_beta_after"] = np.array(betas, np.float32)


# ---------------------------------------------------------------- G3 beta search
def g3():
    cfg = point_cfg(256, 30, multi_modal=True, task="push_pull")
    pl = make_planner(cfg)
    rng = np.random.default_rng(303)
    sets = []
    # (a) rollout-like costs with large spread, (b) nearly flat costs (eta >> 10 -> beta
    # shrinks), (c) a few strong outliers (eta < 3 -> beta grows), (d) mixed halves
    a = (rng.uniform(50, 90, (256, 30))).astype(np.float32)
    b = (60 + rng.uniform(0, 0.05, (256, 30))).astype(np.float32)
    c = (rng.uniform(80, 90, (256, 30))).astype(np.float32)
    c[3] -= 60
    c[200] -= 55
    d = np.concatenate([a[:128], b[128:]]).astype(np.float32)
    for i, cs in enumerate((a, b, c, d)):
        pl._multi_modal_exp_util(torch.from_numpy(cs.copy()))
        sets.append(cs)
        out[f"g3_w1_{i}"] = pl.weights_1.numpy().copy()
        out[f"g3_w2_{i}"] = pl.weights_2.numpy().copy()
        out[f"g3_w_{i}"] = pl.weights.numpy().copy()
        # iteration counts, by re-running the search with a counting exp
        J = ref.mppi_utils.cost_to_go(torch.from_numpy(cs.copy()), pl.gamma_seq)[:, 0]
        its = []
        for sub in (J[:128] - J[:128].min(), J[128:] - J[128:].min(), J - J.min()):
            n = [0]
            real_exp = torch.exp

            def counting_exp(x, _n=n, _e=real_exp):
                _n[0] += 1
                return _e(x)

            ref.m3p2i.torch.exp = counting_exp
            eta, _ = pl.update_infinite_beta(sub, 1, 10, 3)
            ref.m3p2i.torch.exp = real_exp
            its.append(n[0])
        out[f"g3_iters_{i}"] = np.array(its, np.int32)
    out["g3_costs"] = np.stack(sets)


# ---------------------------------------------------------------- G4 distribution updates
def g4():
    rng = np.random.default_rng(404)
    K, T, nu = 128, 30, 2
    costs = rng.uniform(20, 60, (K, T)).astype(np.float32)
    actions = rng.uniform(-3, 3, (K, T, nu)).astype(np.float32)
    mean0 = rng.uniform(-1, 1, (T, nu)).astype(np.float32)
    out["g4_costs"], out["g4_actions"], out["g4_mean0"] = costs, actions, mean0
    pl = make_planner(point_cfg(K, T))
    pl.mean_action = torch.from_numpy(mean0.copy())
    delta = pl._update_distribution(torch.from_numpy(costs.copy()),
                                    torch.from_numpy(actions.copy()))
    out["g4_s_mean"] = pl.mean_action.numpy().copy()
    out["g4_s_best"] = pl.best_traj.numpy().copy()
    out["g4_s_best_idx"] = np.array(int(pl.best_idx), np.int32)
    out["g4_s_weights"] = pl.weights.numpy().copy()
    out["g4_s_delta"] = delta.numpy().copy()
    pl = make_planner(point_cfg(K, T, multi_modal=True, task="push_pull"))
    pl.mean_action = torch.from_numpy(mean0.copy())
    delta = pl._update_multi_modal_distribution(torch.from_numpy(costs.copy()),
                                                torch.from_numpy(actions.copy()))
    out["g4_m_mean"] = pl.mean_action.numpy().copy()
    out["g4_m_mean1"] = pl.mean_action_1.numpy().copy()
    out["g4_m_mean2"] = pl.mean_action_2.numpy().copy()
    out["g4_m_best1"] = pl.best_traj_1.numpy().copy()
    out["g4_m_best2"] = pl.best_traj_2.numpy().copy()
    out["g4_m_idx"] = np.array([int(pl.best_idx_1), int(pl.best_idx_2)], np.int32)
    out["g4_m_weights"] = pl.weights.numpy().copy()
    pl.weights = pl.weights  # get_pull_preference reads self.weights
    out["g4_m_pref"] = np.array(pl.get_pull_preference(), np.int32)


# ---------------------------------------------------------------- G5 action assembly
def g5():
    rng = np.random.default_rng(505)
    for tag, cfg, nu in (("s", point_cfg(64, 30), 2),
                         ("m", point_cfg(64, 30, multi_modal=True, task="push_pull"), 2),
                         ("p", panda_cfg(64, 20), 9),
                         ("pm", panda_cfg(64, 20, multi_modal=True), 9)):
        K, T = cfg.mppi.num_samples, cfg.mppi.horizon
        rec = []

        def dyn(state, u, t=None, _rec=rec):
            _rec.append(u.clone())
            return torch.zeros(K, 4), u

        pl = make_planner(cfg, dynamics=dyn, running_cost=lambda s: torch.zeros(K))
        delta = (rng.standard_normal((K, T, nu)) * 1.2).astype(np.float32)
        pl.delta = torch.from_numpy(delta.copy())
        means = [rng.uniform(-2.5, 2.5, (T, nu)).astype(np.float32) for _ in range(5)]
        pl.mean_action = torch.from_numpy(means[0].copy())
        pl.mean_action_1 = torch.from_numpy(means[1].copy())
        pl.mean_action_2 = torch.from_numpy(means[2].copy())
        pl.best_traj_1 = torch.from_numpy(means[3].copy())
        pl.best_traj_2 = torch.from_numpy(means[4].copy())
        pl.state = torch.zeros(4)
        grip = 0
        if cfg.env_type == "panda_env":
            pl.update_gripper_command("pick" if tag == "p" else "reach")
            grip = 2 if tag == "p" else 1
        pl._compute_total_cost_batch_halton()
        out[f"g5_{tag}_delta"] = delta
        out[f"g5_{tag}_means"] = np.stack(means)
        out[f"g5_{tag}_grip"] = np.array(grip, np.int32)
        out[f"g5_{tag}_u"] = torch.stack(rec, dim=1).numpy().copy()  # [K,T,nu] as fed to dynamics
        out[f"g5_{tag}_actions"] = pl.actions.numpy().copy()


# ---------------------------------------------------------------- G6 point-env costs, G7 suction
def g6_g7():
    rng = np.random.default_rng(606)
    K = 64
    box = np.tile(np.array([[0.0, 2.0]], np.float32), (K, 1))
    box[32:] += rng.uniform(-1, 1, (32, 2)).astype(np.float32)
    robot = (box + rng.uniform(-1.2, 1.2, (K, 2))).astype(np.float32)
    # edge cases: close to the 0.5 and 1/1.8 thresholds, straight behind / in front
    robot[0] = box[0] + np.array([0.5, 0.0], np.float32)
    robot[1] = box[1] + np.array([0.0, -0.5], np.float32)
    robot[2] = box[2] + np.array([1 / 1.8 - 1e-3, 0.0], np.float32)
    robot[3] = box[3] + np.array([1 / 1.8 + 1e-3, 0.0], np.float32)
    robot[4] = box[4] + np.array([0.3, 0.0], np.float32)
    robot[5] = box[5] + np.array([-0.3, 0.1], np.float32)
    robot[6] = box[6] + np.array([0.62, 0.0], np.float32)   # between 1/1.8 and 1/1.5
    vel = rng.uniform(-2, 2, (K, 2)).astype(np.float32)
    vel[4] = [-1.0, 0.0]
    vel[5] = [-1.0, 0.0]
    vel[7] = [0.0, 0.0]
    dynf = np.zeros((K, 3), np.float32)
    dynf[::3, 0] = rng.uniform(-5, 5, len(dynf[::3]))
    dynf[1::4, 1] = rng.uniform(-0.1, 0.1, len(dynf[1::4]))
    dynf[10] = [0.06, 0.05, 3.0]   # |fx|+|fy| = 0.11 > 0.1
    dynf[11] = [0.05, -0.04, 9.0]  # 0.09 <= 0.1
    out["g6_robot"], out["g6_vel"], out["g6_box"], out["g6_dynf"] = robot, vel, box, dynf
    for task, goal, mm in (("push", (-1.0, -1.0), False), ("pull", (0.0, 0.0), False),
                           ("push_pull", (-3.75, -3.75), True), ("navigation", (-3.0, 3.0), False),
                           ("pull", (1.0, 3.0), True)):
        cfg = point_cfg(K, 30, multi_modal=mm, task=task, goal=goal)
        obj = ref.cost_functions.Objective(cfg)
        obj.update_objective(task, list(goal))
        sim = refshim.SynthSim(robot, vel, box, dynf)
        c = obj.compute_cost(sim)
        key = f"{task}_{int(mm)}"
        out[f"g6_cost_{key}"] = c.numpy().copy()
        out[f"g6_goal_{key}"] = np.array(goal, np.float32)
        if sim.applied is not None:
            f = sim.applied.view(K, 13, 3).numpy()
            out[f"g7_fbox_{key}"] = f[:, refshim.POINT_ACTORS.index("box"), :2].copy()
            out[f"g7_frobot_{key}"] = f[:, -1, :2].copy()
            assert np.all(np.delete(f, [6, 12], axis=1) == 0)
    # K == 1 threshold 1.5 (skill_utils.py:75-78)
    for i, off in enumerate((0.62, 0.70)):
        cfg = point_cfg(1, 30, task="pull", goal=(0.0, 0.0))
        sim = refshim.SynthSim(np.array([[off, 2.0]], np.float32), np.zeros((1, 2), np.float32),
                               np.array([[0.0, 2.0]], np.float32))
        f = ref.skill_utils.calculate_suction(cfg, sim).numpy()
        out[f"g7_k1_fbox_{i}"] = f[0, 6, :2].copy()
        out[f"g7_k1_off_{i}"] = np.array(off, np.float32)


# ---------------------------------------------------------------- G7b quaternion costs
def g7_quat():
    rng = np.random.default_rng(707)
    n = 64
    q = rng.standard_normal((3, n, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    # some axis-aligned / flipped cubes
    s = np.float32(np.sqrt(0.5))
    q[1, 0] = [0, 0, 0, 1]
    q[1, 1] = [1, 0, 0, 0]
    q[1, 2] = [0, s, 0, s]
    q[1, 3] = [s, 0, 0, s]
    qe, qc, qg = (torch.from_numpy(q[i].copy()) for i in range(3))
    out["g7q_qe"], out["g7q_qc"], out["g7q_qg"] = q[0], q[1], q[2]
    out["g7q_cube2goal"] = ref.skill_utils.get_general_ori_cube2goal(qc, qg).numpy().copy()
    out["g7q_ee2cube_0"] = ref.skill_utils.get_general_ori_ee2cube(qe, qc, tilt_value=0).numpy().copy()
    out["g7q_ee2cube_t"] = ref.skill_utils.get_general_ori_ee2cube(qe, qc, tilt_value=0.5).numpy().copy()
    out["g7q_rot"] = ref.skill_utils.quaternion_rotation_matrix(qc).numpy().copy()


# ---------------------------------------------------------------- G6b panda costs
class SynthPanda:
    def __init__(self, left, right, cubeA, cubeB, forces):
        self.links = {("panda", "panda_leftfinger"): left, ("panda", "panda_rightfinger"): right,
                      ("cubeA", "box"): cubeA, ("cubeB", "box"): cubeB}
        self.forces = forces
        self.num_envs = left.shape[0]

    def get_actor_link_by_name(self, a, l):
        return torch.from_numpy(self.links[(a, l)].copy())

    def get_actor_orientation_by_name(self, a):
        return torch.from_numpy(self.links[(a, "box")][:, 3:7].copy())

    def get_actor_contact_forces_by_name(self, a, l):
        return torch.from_numpy(self.forces[a].copy())


def g6_panda():
    rng = np.random.default_rng(808)
    K = 64

    def links(center, spread):
        x = np.zeros((K, 13), np.float32)
        x[:, :3] = center + rng.uniform(-spread, spread, (K, 3))
        q = rng.standard_normal((K, 4)).astype(np.float32)
        x[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
        x[:, 7:] = rng.uniform(-1, 1, (K, 6))
        return x.astype(np.float32)

    left = links(np.array([0.1, -0.02, 1.4]), 0.3)
    right = left.copy()
    right[:, :3] += rng.uniform(-0.05, 0.05, (K, 3)).astype(np.float32)
    cubeA = links(np.array([0.2, -0.2, 1.06]), 0.02)
    cubeA[:8, 3:7] = [0, 0, 0, 1]
    cubeB = links(np.array([0.2, 0.2, 1.06]), 0.0)
    forces = {n: np.zeros((K, 3), np.float32) for n in ("table", "shelf_stand", "cubeB")}
    forces["table"][::5, 0] = 0.3
    forces["shelf_stand"][1::7, 1] = 0.03   # x4 = 0.12 > 0.1
    forces["cubeB"][2::9, 0] = 0.09
    for k, v in (("left", left), ("right", right), ("cubeA", cubeA), ("cubeB", cubeB)):
        out[f"g6p_{k}"] = v
    for n, f in forces.items():
        out[f"g6p_f_{n}"] = f
    goal7 = np.array([0.2, 0.2, 1.115, 0, 0, 0, 1], np.float32)
    out["g6p_goal7"] = goal7
    for mm in (False, True):
        cfg = panda_cfg(K, 20, multi_modal=mm)
        obj = ref.cost_functions.Objective(cfg)
        for task in ("reach", "pick", "place"):
            obj.update_objective(task, torch.from_numpy(goal7.copy()))
            sim = SynthPanda(left, right, cubeA, cubeB, {k: v.copy() for k, v in forces.items()})
            c = obj.compute_cost(sim)
            out[f"g6p_cost_{task}_{int(mm)}"] = c.numpy().copy()


# ---------------------------------------------------------------- G8 halton + bspline
def g8():
    for (K, T, nu) in ((64, 12, 2), (64, 30, 2), (64, 20, 9)):
        n_knots = T // 4
        knots = ref.mppi_utils.generate_gaussian_halton_samples(
            K, n_knots * nu, use_ghalton=False, seed_val=0, device="cpu",
            float_dtype=torch.float32)
        ks = knots.view(K, nu, n_knots)
        delta = torch.zeros(K, T, nu)
        for i in range(K):
            for j in range(nu):
                delta[i, :, j] = ref.skill_utils.bspline(ks[i, j, :], n=T, degree=2)
        out[f"g8_knots_{K}_{T}_{nu}"] = knots.numpy().copy()
        out[f"g8_delta_{K}_{T}_{nu}"] = delta.numpy().copy()
    out["g8_primes"] = np.array(ref.mppi_utils.generate_prime_numbers(20), np.int32)


# ---------------------------------------------------------------- G10 savgol
def g10():
    from scipy import signal
    rng = np.random.default_rng(1010)
    for (T, nu) in ((30, 2), (12, 9), (9, 2), (20, 9)):
        x = rng.uniform(-3, 3, (T, nu)).astype(np.float32)
        y = signal.savgol_filter(x, 9, 2, deriv=0, delta=1.0, axis=0, mode="interp", cval=0.0)
        out[f"g10_in_{T}_{nu}"] = x
        out[f"g10_out_{T}_{nu}"] = y.astype(np.float32)


# ---------------------------------------------------------------- G9 full command() traces
def halton_delta(K, T, nu):
    """The build's own sampler (in-tree van-der-Corput branch + FITPACK spline), computed
    here by the reference's functions so the fixture is self-contained."""
    n_knots = T // 4
    knots = ref.mppi_utils.generate_gaussian_halton_samples(
        K, n_knots * nu, use_ghalton=False, seed_val=0, device="cpu", float_dtype=torch.float32)
    ks = knots.view(K, nu, n_knots)
    delta = torch.zeros(K, T, nu)
    for i in range(K):
        for j in range(nu):
            delta[i, :, j] = ref.skill_utils.bspline(ks[i, j, :], n=T, degree=2)
    return delta


class StreamDist:
    """Stands in for the planner's MultivariateNormal (mppi.py:129-131): N(noise_mu, noise_sigma) draws of the
    build's counter-based stream -- noise_mu + L z with L = chol(noise_sigma), accumulated in the order
    DESIGN.md section 5 fixes (for the diagonal default: z * sqrt(sigma) bit for bit)."""

    def __init__(self, cfg, nu, seed, calls):
        m = cfg.mppi
        self.K, self.seed, self.calls = m.num_samples, seed, calls
        self.ocfg = O.make_cfg(m.num_samples, m.horizon, nu, noise_mu=m.noise_mu or None, noise_sigma=m.noise_sigma)

    def sample(self, shape):
        return torch.from_numpy(O.noise_fill(self.ocfg, self.seed, self.calls[0], self.K))


def g9_trace(tag, cfg, ncalls, world0, closed_loop=True, noise_stream=False, seed=7, extra=None):
    """reactive_tamp.py wiring (:22-41, :43-73) around the reference planner."""
    K, T = cfg.mppi.num_samples, cfg.mppi.horizon
    sim = refshim.OracleSim(K, world0)
    real = O.init_world(1)
    real[0] = np.array(world0, np.float32)
    obj = ref.cost_functions.Objective(cfg)
    obj.update_objective(cfg.task, cfg.goal)

    def dynamics(_, u, t=None):
        sim.set_dof_velocity_target_tensor(u)
        sim.step()
        states = torch.stack([sim.robot_pos[:, 0], sim.robot_vel[:, 0], sim.robot_pos[:, 1],
                              sim.robot_vel[:, 1]], dim=1)
        return states, u

    pl = make_planner(cfg, dynamics=dynamics, running_cost=lambda _: obj.compute_cost(sim))
    if cfg.mppi.sampling_method == "halton" and cfg.mppi.mppi_mode == "halton-spline":
        pl.delta = halton_delta(K, T, 2)
        out[f"g9_{tag}_delta"] = pl.delta.numpy().copy()
    calls = [0]
    if noise_stream:
        # replace torch's global RNG draw (mppi.py:340 / :481) by the build's counter-based
        # stream so both sides see identical noise: N(0, Sigma) = z * sqrt(diag Sigma)
        pl.noise_dist = StreamDist(cfg, 2, seed, calls)
        if cfg.mppi.mppi_mode == "simple":
            pl.U = torch.zeros(T, 2)  # reference draws U from the global RNG (mppi.py:134)
    sc = O.default_scene()
    acts, ws, tops, prefs, worlds, means, Js, extras = [], [], [], [], [], [], [], []
    for call in range(ncalls):
        worlds.append(real[0].copy())
        sim.reset(real[0])
        a = pl.command(sim._dof_state[0])
        calls[0] += 1
        acts.append(a.numpy().copy())
        ws.append(pl.weights.numpy().copy())
        tops.append(pl.top_trajs.numpy().copy())
        prefs.append(int(pl.get_pull_preference()))
        if cfg.mppi.mppi_mode == "simple":
            means.append(pl.U.numpy().copy())
            Js.append(pl.cost_total.numpy().copy())
        else:
            means.append(pl.mean_action.numpy().copy())
            Js.append(pl.total_costs.numpy().copy() if hasattr(pl, "total_costs") and
                      not cfg.multi_modal else np.zeros(K, np.float32))
        if extra is not None:
            extras.append(extra(pl))
        if closed_loop:
            O.step_batch(sc, real, a[0:1].numpy())
    if extras:
        out[f"g9_{tag}_extra"] = np.stack(extras)
    out[f"g9_{tag}_world"] = np.stack(worlds)
    out[f"g9_{tag}_action"] = np.stack(acts)
    out[f"g9_{tag}_weights"] = np.stack(ws)
    out[f"g9_{tag}_top_trajs"] = np.stack(tops)
    out[f"g9_{tag}_pref"] = np.array(prefs, np.int32)
    out[f"g9_{tag}_mean"] = np.stack(means)
    out[f"g9_{tag}_J"] = np.stack(Js)
    out[f"g9_{tag}_states_last"] = pl.states.numpy().copy()
    out[f"g9_{tag}_actions_last"] = pl.actions.numpy().copy()


def g9():
    w0 = O.init_world(1)[0]
    # C2-shaped: push, goal [-1,-1], halton-spline, reduced K
    g9_trace("push", point_cfg(256, 30, task="push", goal=(-1.0, -1.0)), 6, w0)
    # robot already next to the box so contact happens inside the horizon from call 0
    w1 = w0.copy()
    w1[0], w1[1] = 0.1, 1.45
    g9_trace("pushc", point_cfg(256, 30, task="push", goal=(-1.0, 3.0)), 6, w1)
    # pull single-mode, robot within suction range
    w2 = w0.copy()
    w2[0], w2[1] = 0.0, 1.5
    g9_trace("pull", point_cfg(256, 30, task="pull", goal=(0.0, 0.0)), 6, w2)
    # C3-shaped: push_pull multi-modal
    g9_trace("hybrid", point_cfg(256, 30, multi_modal=True, task="push_pull",
                                 goal=(-3.75, -3.75)), 6, w2)
    # C1: navigation K=100,T=10 simple mode (T=10 cannot use the spline)
    cfg = point_cfg(100, 10, task="navigation", goal=(-3.0, 3.0), mode="simple",
                    sampling="random", filter_u=True, u_per_command=10)
    g9_trace("nav", cfg, 6, w0, noise_stream=True)
    # halton-spline + random sampling (quirk Q4: noise scaled twice)
    cfg = point_cfg(128, 12, task="navigation", goal=(-3.0, 3.0), sampling="random")
    g9_trace("navr", cfg, 4, w0, noise_stream=True)


# ---------------------------------------------------------------- G11 the optional branches of command()
def g11():
    """Reference traces of the MPPIConfig switches no shipped config turns on (mppi.py:39-54): u_scale != 1
    (:297 -- the distribution update consumes the SCALED stack, :313,:331), U_init / u_init (dead: :122-123,
    :132-133), noise_mu and a non-diagonal noise_sigma (MultivariateNormal, :129-131), noise_abs_cost
    (:366-367), update_cov (:508-516)."""
    w0 = O.init_world(1)[0]
    w1 = w0.copy()
    w1[0], w1[1] = 0.1, 1.45
    cfg = point_cfg(256, 30, task="push", goal=(-1.0, 3.0))
    cfg.mppi.u_scale = 0.5
    g9_trace("opt_uscale", cfg, 4, w1)
    cfg = point_cfg(256, 30, task="push", goal=(-1.0, 3.0))
    cfg.mppi.U_init = [[1.0, -1.0]] * 30
    cfg.mppi.u_init = 0.7
    g9_trace("opt_dead", cfg, 3, w1)
    cfg = point_cfg(256, 30, task="push", goal=(-1.0, 3.0))
    cfg.mppi.update_cov = True
    g9_trace("opt_cov", cfg, 5, w1, extra=lambda pl: pl.scale_tril.numpy().copy())
    # simple mode: |noise| action cost, scaled controls, biased full-covariance noise
    cfg = point_cfg(100, 10, task="navigation", goal=(-3.0, 3.0), mode="simple", sampling="random",
                    filter_u=True, u_per_command=10)
    cfg.mppi.noise_abs_cost = True
    cfg.mppi.u_scale = 0.8
    cfg.mppi.noise_mu = [0.3, -0.2]
    cfg.mppi.noise_sigma = [[3.0, 1.0], [1.0, 2.0]]
    g9_trace("opt_abs", cfg, 4, w0, noise_stream=True)
    # halton-spline + random sampling with the same noise distribution (the sample is scaled once more by
    # sqrt(diag Sigma), quirk Q4)
    cfg = point_cfg(128, 12, task="navigation", goal=(-3.0, 3.0), sampling="random")
    cfg.mppi.noise_mu = [0.3, -0.2]
    cfg.mppi.noise_sigma = [[3.0, 1.0], [1.0, 2.0]]
    g9_trace("opt_navr", cfg, 4, w0, noise_stream=True)
    # the same switches on the panda_env (C4-shaped, reduced K)
    import oracle.panda as P
    goal7 = [0.2, 0.2, 1.115, 0.0, 0.0, 0.0, 1.0]
    wp = P.init_world(1)[0]
    g9_panda_trace("panda_opt_cov", 256, 20, "reach", False, wp, goal7, ncalls=4, mppi_kw=dict(update_cov=True),
                   extra=lambda pl: pl.scale_tril.numpy().copy())
    sig = [[0.0] * 9 for _ in range(9)]
    for i in range(7):
        sig[i][i] = 10.0
    sig[7][7] = sig[8][8] = 0.8
    sig[0][1] = sig[1][0] = 4.0
    sig[2][5] = sig[5][2] = -3.0
    mu = [0.2, -0.1, 0.0, 0.1, 0.0, 0.0, -0.2, 0.0, 0.0]
    g9_panda_trace("panda_opt_rand", 128, 12, "reach", False, wp, goal7, ncalls=4, noise_stream=True,
                   mppi_kw=dict(sampling_method="random", noise_mu=mu, noise_sigma=sig))
    g9_panda_trace("panda_opt_simple", 128, 12, "reach", False, wp, goal7, ncalls=4, noise_stream=True,
                   mppi_kw=dict(mppi_mode="simple", sampling_method="random", noise_mu=mu, noise_sigma=sig,
                                noise_abs_cost=True, u_scale=0.9, u_per_command=12))


# ---------------------------------------------------------------- G9 (panda): full command() traces
def g9_panda_trace(tag, K, T, task, multi_modal, world0, goal7, ncalls=5, mppi_kw=None, noise_stream=False, seed=7,
                   extra=None):
    """The reference's M3P2I + Objective driven through their plugin API on the panda_env (C4-shaped, reduced
    K), the oracle's chain dynamics behind the wrapper API: pins, in ONE reference trace, what G2 / G5 / G6b
    pin piecewise -- the gripper override (mppi.py:412-416), the persistent adapted beta (mppi.py:446-454),
    the reach / pick dispatch (cost_functions.py:19-36, 91-136) and update_gripper_command (m3p2i.py:10-14).
    Closed loop: the first action of every plan steps the 1-env world (scripts/sim.py:41-52)."""
    import oracle.panda as P
    cfg = panda_cfg(K, T, multi_modal=multi_modal)
    for k, v in (mppi_kw or {}).items():
        setattr(cfg.mppi, k, v)
    simple = cfg.mppi.mppi_mode == "simple"
    sim = refshim.OraclePandaSim(K, world0)
    obj = ref.cost_functions.Objective(cfg)
    obj.update_objective(task, torch.from_numpy(np.array(goal7, np.float32)))

    def dynamics(_, u, t=None):          # reactive_tamp.py:63-70
        sim.set_dof_velocity_target_tensor(u)
        sim.step()
        return torch.stack([sim.robot_pos[:, 0], sim.robot_vel[:, 0], sim.robot_pos[:, 1], sim.robot_vel[:, 1]], dim=1), u

    pl = make_planner(cfg, dynamics=dynamics, running_cost=lambda _: obj.compute_cost(sim))
    pl.update_gripper_command(task)      # reactive_tamp.py:78
    calls = [0]
    if noise_stream:     # the build's counter-based stream in place of torch's global generator (as g9_trace)
        pl.noise_dist = StreamDist(cfg, 9, seed, calls)
        if simple:
            pl.U = torch.zeros(T, 9)
    else:
        pl.delta = halton_delta(K, T, 9)
        out[f"g9_{tag}_delta"] = pl.delta.numpy().copy()
    sc = P.default_scene()
    real = np.array(world0, np.float32).reshape(1, -1).copy()
    rec = {k: [] for k in ("world", "action", "weights", "top_trajs", "mean", "beta", "pref", "J")}
    extras = []
    for call in range(ncalls):
        rec["world"].append(real[0].copy())
        sim.reset(real[0])
        a = pl.command(sim._dof_state[0])
        calls[0] += 1
        rec["action"].append(a.numpy().copy())
        rec["weights"].append(pl.weights.numpy().copy())
        rec["top_trajs"].append(pl.top_trajs.numpy().copy())
        rec["mean"].append((pl.U if simple else pl.mean_action).numpy().copy())
        rec["beta"].append(float(pl.beta))
        rec["pref"].append(int(pl.get_pull_preference()))
        rec["J"].append(pl.cost_total.numpy().copy() if simple else np.zeros(K, np.float32))
        if extra is not None:
            extras.append(extra(pl))
        u1 = np.zeros((1, 9), np.float32)
        u1[0] = a[0].numpy()
        P.step_batch(sc, real, u1)
    if not simple:
        rec.pop("J")
    if extras:
        out[f"g9_{tag}_extra"] = np.stack(extras)
    for k, v in rec.items():
        out[f"g9_{tag}_{k}"] = np.array(v, np.int32 if k == "pref" else np.float32) if k in ("beta", "pref") else np.stack(v)
    out[f"g9_{tag}_states_last"] = pl.states.numpy().copy()
    out[f"g9_{tag}_actions_last"] = pl.actions.numpy().copy()


def g9_panda():
    import oracle.panda as P
    from tests.panda_worlds import grasp_world
    sc = P.default_scene()
    goal7 = [0.2, 0.2, 1.115, 0.0, 0.0, 0.0, 1.0]   # pre-place pose above cubeB (task_planner.py:96-97)
    w_init = P.init_world(1)[0]
    g9_panda_trace("panda_reach", 256, 20, "reach", False, w_init, goal7)
    g9_panda_trace("panda_reachmm", 256, 20, "reach", True, w_init, goal7)
    g9_panda_trace("panda_pick", 256, 20, "pick", False, grasp_world(P, sc), goal7)
    # quirk Q8 (cost_functions.py:97 `cube_state[0, :3]`, skill_utils.py:274): the open gripper stands around cubeA, 3 mm
    # from one finger, so the rollouts' arm motions push the cube -- environment 0's too -- and the reach cost of EVERY
    # rollout is measured against the cube of environment 0 (tilted mode: + the orientation of the first environment of
    # the second half)
    w_touch = grasp_world(P, sc, close_gripper=False, offset=(0.0, 0.012))
    g9_panda_trace("panda_reach_touch", 256, 20, "reach", False, w_touch, goal7, ncalls=4)
    g9_panda_trace("panda_reachmm_touch", 256, 20, "reach", True, w_touch, goal7, ncalls=4)


# ---------------------------------------------------------------- G7 (skill side): the real world's suction
def g7_skill():
    """utils/skill_utils.py:36-94 as scripts/sim.py:41-49 calls them on its 1-env world: calculate_suction on
    K = 64 environments (threshold 1.8) and on single environments (threshold 1.5), and
    check_suction_condition for single environments with given actions."""
    rng = np.random.default_rng(707)
    K = 64
    box = np.tile(np.array([[0.0, 2.0]], np.float32), (K, 1)) + rng.uniform(-0.2, 0.2, (K, 2)).astype(np.float32)
    ang = rng.uniform(0, 2 * np.pi, K)
    rad = rng.uniform(0.25, 0.9, K)          # both sides of 1/1.8 = 0.556 and of 1/1.5 = 0.667
    robot = (box + np.stack([rad * np.cos(ang), rad * np.sin(ang)], 1)).astype(np.float32)
    cfg = point_cfg(K, 30, task="pull", goal=(0.0, 0.0))
    sim = refshim.SynthSim(robot, np.zeros((K, 2), np.float32), box)
    out["g7s_robot"], out["g7s_box"] = robot, box
    out["g7s_forces_K64"] = ref.skill_utils.calculate_suction(cfg, sim).numpy().copy()
    f1, cond, acts = [], [], []
    for i in range(K):
        s1 = refshim.SynthSim(robot[i:i + 1], np.zeros((1, 2), np.float32), box[i:i + 1])
        f1.append(ref.skill_utils.calculate_suction(cfg, s1).numpy()[0].copy())
        a = rng.uniform(-1, 1, 2).astype(np.float32)
        acts.append(a)
        cfg.task, cfg.suction_active = "pull", True
        cond.append(bool(ref.skill_utils.check_suction_condition(cfg, s1, torch.from_numpy(a))))
    out["g7s_forces_K1"] = np.stack(f1)
    out["g7s_action"] = np.stack(acts)
    out["g7s_condition"] = np.array(cond, np.int32)


if __name__ == "__main__":
    for fn in (g1, g2, g3, g4, g5, g6_g7, g7_quat, g6_panda, g8, g10, g9, g9_panda, g7_skill, g11):
        fn()
        print(fn.__name__, "ok")
    path = os.path.join(HERE, "ref_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")
