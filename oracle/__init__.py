"""CPU ORACLE -- test infrastructure, NOT product code.

ctypes bindings for ``oracle/_build/libm3oracle.so`` (plain-C restatement of the
reference's MPPI/M3P2I arithmetic + this repository's own dynamics spec, see
``oracle/m3_oracle.h``).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package.  Nothing under
``m3p2i_aip_amd/`` imports it.

Build: ``make -C oracle`` (gcc, OpenMP).  ``load()`` builds on demand when gcc is present.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libm3oracle.so")
_lib = None

MAX_NU = 9
WORLD_FLOATS = 31  # m3o_point_world = 3 bodies x 7 + fext(4) + fc(6)
# field offsets inside one world row
W_R, W_B, W_D = 0, 7, 14
W_FEXT_R, W_FEXT_B = 21, 23
W_FC_R, W_FC_B, W_FC_D = 25, 27, 29

TASKS = {"navigation": 0, "push": 1, "pull": 2, "push_pull": 3, "reach": 4, "pick": 5,
         "place": 6, "idle": 7}


class PointScene(C.Structure):
    _fields_ = [("dt", C.c_float), ("substeps", C.c_int), ("iters", C.c_int), ("g", C.c_float),
                ("robot_r", C.c_float), ("robot_m", C.c_float), ("drive_damping", C.c_float),
                ("drive_fmax", C.c_float),
                ("box_hx", C.c_float), ("box_hy", C.c_float), ("box_m", C.c_float),
                ("box_I", C.c_float), ("box_mu_g", C.c_float), ("box_req", C.c_float),
                ("dyn_hx", C.c_float), ("dyn_hy", C.c_float), ("dyn_m", C.c_float),
                ("dyn_I", C.c_float), ("dyn_mu_g", C.c_float), ("dyn_req", C.c_float),
                ("obs_x", C.c_float), ("obs_y", C.c_float), ("obs_hx", C.c_float),
                ("obs_hy", C.c_float), ("wall", C.c_float),
                ("mu_rb", C.c_float), ("mu_rd", C.c_float), ("mu_ro", C.c_float),
                ("mu_rw", C.c_float), ("mu_bw", C.c_float), ("mu_dw", C.c_float),
                ("mu_bd", C.c_float), ("mu_bo", C.c_float), ("mu_do", C.c_float),
                ("contact_offset", C.c_float), ("baumgarte", C.c_float), ("slop", C.c_float),
                ("max_bias", C.c_float), ("face_tol", C.c_float), ("friction_coupling", C.c_int),
                ("fext_substeps", C.c_int)]


class Cfg(C.Structure):
    _fields_ = [("K", C.c_int), ("T", C.c_int), ("nu", C.c_int), ("multi_modal", C.c_int),
                ("env_type", C.c_int), ("sample_null_action", C.c_int),
                ("mode_simple", C.c_int), ("filter_u", C.c_int), ("u_per_command", C.c_int),
                ("u_min", C.c_float * MAX_NU), ("u_max", C.c_float * MAX_NU),
                ("scale_tril", C.c_float * MAX_NU), ("sigma_inv", C.c_float * MAX_NU),
                ("u_scale", C.c_float), ("gamma", C.c_float), ("lambda_", C.c_float),
                ("step_size_mean", C.c_float), ("task", C.c_int), ("goal", C.c_float * 7),
                ("kp_suction", C.c_float), ("suction_thresh", C.c_float),
                ("gripper_cmd", C.c_int), ("pre_height_diff", C.c_float),
                ("tilt_cos_theta", C.c_float), ("noise_abs_cost", C.c_int), ("full_sigma", C.c_int),
                ("noise_mu", C.c_float * MAX_NU), ("chol", C.c_float * (MAX_NU * MAX_NU)),
                ("sigma_inv_full", C.c_float * (MAX_NU * MAX_NU)), ("avoid_dyn_obs", C.c_int)]


class UpdateInfo(C.Structure):
    _fields_ = [("beta", C.c_float), ("best_idx", C.c_int), ("best_idx_1", C.c_int),
                ("best_idx_2", C.c_int), ("eta", C.c_float), ("eta_1", C.c_float),
                ("eta_2", C.c_float), ("iters", C.c_int), ("iters_1", C.c_int),
                ("iters_2", C.c_int), ("wsum_push", C.c_float), ("wsum_pull", C.c_float)]


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in os.listdir(_HERE) if f.endswith((".c", ".h"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def load():
    global _lib
    if _lib is not None:
        return _lib
    try:
        build()
    except Exception:  # no gcc on the box: use the prebuilt library that travelled with us
        if not os.path.exists(_LIB_PATH):
            raise
    lib = C.CDLL(_LIB_PATH)
    FP, IP = C.POINTER(C.c_float), C.POINTER(C.c_int)
    lib.m3o_point_scene_default.argtypes = [C.POINTER(PointScene)]
    lib.m3o_point_step_batch.argtypes = [C.POINTER(PointScene), FP, C.c_int, FP]
    lib.m3o_point_cost_batch.argtypes = [C.POINTER(Cfg), FP, C.c_int, C.c_int, FP]
    lib.m3o_assemble_actions.argtypes = [C.POINTER(Cfg), FP, FP, FP, FP, FP, FP, C.c_int,
                                         C.c_int, FP]
    lib.m3o_point_rollout.argtypes = [C.POINTER(Cfg), C.POINTER(PointScene), FP, FP, FP,
                                      C.c_int, C.c_int, FP, FP, FP, FP, FP]
    lib.m3o_cost_to_go0.argtypes = [FP, C.c_int, C.c_int, C.c_float, FP]
    lib.m3o_softmin.argtypes = [FP, C.c_int, C.c_float, FP]
    lib.m3o_softmin.restype = C.c_float
    lib.m3o_beta_search.argtypes = [FP, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, FP,
                                    IP, FP]
    lib.m3o_beta_search.restype = C.c_float
    lib.m3o_update_weights.argtypes = [C.POINTER(Cfg), FP, FP, FP, FP, C.POINTER(UpdateInfo)]
    lib.m3o_partial_sums.argtypes = [C.POINTER(Cfg), FP, FP, FP, FP, C.c_int, C.c_int, FP, FP,
                                     FP]
    lib.m3o_mean_update.argtypes = [C.POINTER(Cfg), FP, FP]
    lib.m3o_topk.argtypes = [FP, C.c_int, C.c_int, IP, FP]
    lib.m3o_savgol9.argtypes = [FP, C.c_int, C.c_int, FP]
    lib.m3o_shift.argtypes = [FP, C.c_int, C.c_int]
    lib.m3o_simple_update.argtypes = [C.POINTER(Cfg), FP, FP, FP, FP, FP]
    lib.m3o_gauss.argtypes = [C.c_ulonglong, C.c_uint, C.c_uint, C.c_uint, C.c_uint]
    lib.m3o_gauss.restype = C.c_float
    lib.m3o_noise_fill.argtypes = [C.POINTER(Cfg), C.c_ulonglong, C.c_uint, C.c_int, C.c_int, C.POINTER(C.c_float)]
    lib.m3o_gauss_fill.argtypes = [C.c_ulonglong, C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int,
                                   FP]
    lib.m3o_ori_cube2goal.argtypes = [FP, FP]
    lib.m3o_ori_cube2goal.restype = C.c_float
    lib.m3o_ori_ee2cube.argtypes = [FP, FP, C.c_float, FP]
    lib.m3o_ori_ee2cube.restype = C.c_float
    lib.m3o_set_threads.argtypes = [C.c_int]
    lib.m3o_max_threads.restype = C.c_int
    _lib = lib
    return lib


# ----------------------------------------------------------------------------------------
# numpy-level helpers
# ----------------------------------------------------------------------------------------
def default_scene() -> PointScene:
    sc = PointScene()
    load().m3o_point_scene_default(C.byref(sc))
    return sc


def make_cfg(K, T, nu=2, multi_modal=False, env_type="point_env", task="push", goal=(0, 0),
             u_min=None, u_max=None, noise_sigma_diag=None, u_scale=1.0, gamma=0.95,
             lambda_=1.0, sample_null_action=True, mode_simple=False, filter_u=True,
             u_per_command=1, kp_suction=400.0, gripper_cmd=0, pre_height_diff=0.05,
             suction_thresh=None, noise_mu=None, noise_sigma=None, noise_abs_cost=False) -> Cfg:
    """noise_sigma: the full [nu][nu] matrix (overrides noise_sigma_diag); its Cholesky factor and inverse are
    formed in binary64 and rounded to f32 -- for a diagonal matrix that is sqrtf / 1.0f/x bit for bit."""
    c = Cfg()
    c.K, c.T, c.nu = int(K), int(T), int(nu)
    c.multi_modal = int(bool(multi_modal))
    c.env_type = 0 if env_type == "point_env" else 1
    c.sample_null_action = int(bool(sample_null_action))
    c.mode_simple = int(bool(mode_simple))
    c.filter_u = int(bool(filter_u))
    c.u_per_command = int(u_per_command)
    if u_min is None:
        u_min = [-3.0] * nu
    if u_max is None:
        u_max = [3.0] * nu
    if noise_sigma is not None:
        sig = np.array(noise_sigma, np.float32).astype(np.float64)
        noise_sigma_diag = [float(sig[j, j]) for j in range(nu)]
        if np.any(sig != np.diag(np.diag(sig))):
            c.full_sigma = 1
            L, inv = np.linalg.cholesky(sig).astype(np.float32), np.linalg.inv(sig).astype(np.float32)
            for i in range(nu):
                for j in range(nu):
                    c.chol[i * nu + j], c.sigma_inv_full[i * nu + j] = float(L[i, j]), float(inv[i, j])
    if noise_sigma_diag is None:
        noise_sigma_diag = [3.0] * nu
    for j in range(nu):      # diagonal of the Cholesky factor: the sampling scale of a diagonal noise_sigma
        if not c.full_sigma:
            c.chol[j * nu + j] = float(np.sqrt(np.float32(noise_sigma_diag[j])))
    c.noise_abs_cost = int(bool(noise_abs_cost))
    for j in range(nu):
        c.noise_mu[j] = float(noise_mu[j]) if noise_mu is not None else 0.0
    for j in range(nu):
        c.u_min[j], c.u_max[j] = float(u_min[j]), float(u_max[j])
        c.scale_tril[j] = float(np.sqrt(np.float32(noise_sigma_diag[j])))
        c.sigma_inv[j] = float(np.float32(1.0) / np.float32(noise_sigma_diag[j]))
    c.u_scale, c.gamma, c.lambda_ = float(u_scale), float(gamma), float(lambda_)
    c.step_size_mean = 0.98
    c.task = TASKS[task] if isinstance(task, str) else int(task)
    g = list(goal) + [0.0] * (7 - len(goal))
    for i in range(7):
        c.goal[i] = float(g[i])
    c.kp_suction = float(kp_suction)
    c.suction_thresh = float(suction_thresh if suction_thresh is not None
                             else (1.5 if K == 1 else 1.8))
    c.gripper_cmd = int(gripper_cmd)
    c.pre_height_diff = float(pre_height_diff)
    c.tilt_cos_theta = 0.5
    return c


def init_world(n=1) -> np.ndarray:
    """n copies of the reference's initial point_env scene (7_box.yaml, 6_dyn_obs.yaml)."""
    w = np.zeros((n, WORLD_FLOATS), np.float32)
    w[:, W_R + 2] = 1.0
    w[:, W_B + 0], w[:, W_B + 1], w[:, W_B + 2] = 0.0, 2.0, 1.0
    w[:, W_D + 0], w[:, W_D + 1], w[:, W_D + 2] = -2.0, 2.0, 1.0
    return w


def step_batch(sc: PointScene, worlds: np.ndarray, u: np.ndarray) -> None:
    assert worlds.dtype == np.float32 and worlds.flags.c_contiguous
    u = f32(u)
    load().m3o_point_step_batch(C.byref(sc), _fp(worlds), worlds.shape[0], _fp(u))


def cost_batch(cfg: Cfg, worlds: np.ndarray, k0: int = 0) -> np.ndarray:
    c = np.zeros(worlds.shape[0], np.float32)
    load().m3o_point_cost_batch(C.byref(cfg), _fp(worlds), worlds.shape[0], k0, _fp(c))
    return c


def assemble_actions(cfg, delta, mean, mean1, mean2, best1, best2, k0=0, k1=None):
    k1 = cfg.K if k1 is None else k1
    act = np.zeros((k1 - k0, cfg.T, cfg.nu), np.float32)
    delta, mean, mean1, mean2, best1, best2 = map(f32, (delta, mean, mean1, mean2, best1, best2))
    load().m3o_assemble_actions(C.byref(cfg), _fp(delta), _fp(mean), _fp(mean1), _fp(mean2),
                                _fp(best1), _fp(best2), k0, k1, _fp(act))
    return act


def point_rollout(cfg, sc, world0, act, pend=None, k0=0, k1=None):
    """Returns dict(states[n,T,4], actions[n,T,nu], cost_h[n,T], J[n], S[n]); pend updated."""
    k1 = cfg.K if k1 is None else k1
    n = k1 - k0
    act = f32(act)
    world0 = f32(world0).reshape(-1)[:WORLD_FLOATS].copy()
    states = np.zeros((n, cfg.T, 4), np.float32)
    actions = np.zeros((n, cfg.T, cfg.nu), np.float32)
    cost_h = np.zeros((n, cfg.T), np.float32)
    J = np.zeros(n, np.float32)
    S = np.zeros(n, np.float32)
    pp = _fp(pend) if pend is not None else None
    load().m3o_point_rollout(C.byref(cfg), C.byref(sc), _fp(world0), pp, _fp(act), k0, k1,
                             _fp(states), _fp(actions), _fp(cost_h), _fp(J), _fp(S))
    return dict(states=states, actions=actions, cost_h=cost_h, J=J, S=S)


def cost_to_go0(cost_h, gamma=0.95):
    cost_h = f32(cost_h)
    J = np.zeros(cost_h.shape[0], np.float32)
    load().m3o_cost_to_go0(_fp(cost_h), cost_h.shape[0], cost_h.shape[1], gamma, _fp(J))
    return J


def softmin(J, beta=1.0):
    J = f32(J)
    w = np.zeros_like(J)
    eta = load().m3o_softmin(_fp(J), J.shape[0], beta, _fp(w))
    return w, eta


def beta_search(J, beta0=1.0, eta_u=10.0, eta_l=3.0, max_iters=1000):
    J = f32(J)
    w = np.zeros_like(J)
    it = C.c_int(0)
    b = C.c_float(0)
    eta = load().m3o_beta_search(_fp(J), J.shape[0], beta0, eta_u, eta_l, max_iters, _fp(w),
                                 C.byref(it), C.byref(b))
    return w, eta, it.value, b.value


def update_weights(cfg, J, beta=1.0):
    J = f32(J)
    K = cfg.K
    w = np.zeros(K, np.float32)
    w1 = np.zeros(K // 2, np.float32)
    w2 = np.zeros(K - K // 2, np.float32)
    info = UpdateInfo()
    info.beta = beta
    load().m3o_update_weights(C.byref(cfg), _fp(J), _fp(w), _fp(w1), _fp(w2), C.byref(info))
    return w, w1, w2, info


def partial_sums(cfg, w, w1, w2, actions, k0=0, k1=None):
    k1 = cfg.K if k1 is None else k1
    n = cfg.T * cfg.nu
    ps = np.zeros((3, cfg.T, cfg.nu), np.float32)
    actions = f32(actions)
    load().m3o_partial_sums(C.byref(cfg), _fp(f32(w)), _fp(f32(w1)), _fp(f32(w2)), _fp(actions),
                            k0, k1, _fp(ps[0]), _fp(ps[1]), _fp(ps[2]))
    return ps


def mean_update(cfg, mean, s):
    mean = f32(mean).copy()
    load().m3o_mean_update(C.byref(cfg), _fp(mean), _fp(f32(s)))
    return mean


def topk(w, n=20):
    w = f32(w)
    idx = np.zeros(n, np.int32)
    val = np.zeros(n, np.float32)
    load().m3o_topk(_fp(w), w.shape[0], n, _ip(idx), _fp(val))
    return idx, val


def savgol9(x):
    x = f32(x)
    out = np.zeros_like(x)
    load().m3o_savgol9(_fp(x), x.shape[0], x.shape[1], _fp(out))
    return out


def shift(seq):
    seq = f32(seq).copy()
    load().m3o_shift(_fp(seq), seq.shape[0], seq.shape[1])
    return seq


def simple_update(cfg, S, perturbed, U):
    U = f32(U).copy()
    ct = np.zeros(cfg.K, np.float32)
    w = np.zeros(cfg.K, np.float32)
    load().m3o_simple_update(C.byref(cfg), _fp(f32(S)), _fp(f32(perturbed)), _fp(U), _fp(ct),
                             _fp(w))
    return U, ct, w


def gauss_fill(seed, call, K, T, nu, k0=0):
    out = np.zeros((K, T, nu), np.float32)
    load().m3o_gauss_fill(seed, call, k0, K, T, nu, _fp(out))
    return out


def noise_fill(cfg, seed, call, K, k0=0):
    """[K, T, nu] draws of N(noise_mu, noise_sigma) from the build's stream (mppi.py:129-131, :340, :481)."""
    out = np.zeros((K, cfg.T, cfg.nu), np.float32)
    load().m3o_noise_fill(C.byref(cfg), seed, call, k0, K, _fp(out))
    return out


def ori_cube2goal(qc, qg):
    return load().m3o_ori_cube2goal(_fp(f32(qc)), _fp(f32(qg)))


def ori_ee2cube(qe, qc, tilt, qc0):
    return load().m3o_ori_ee2cube(_fp(f32(qe)), _fp(f32(qc)), tilt, _fp(f32(qc0)))


class OraclePointPlanner:
    """Whole command() of the point_env planner on the oracle (halton-spline or simple mode).

    Mirrors MPPI.command (mppi.py:211-264) + M3P2I update (m3p2i.py:66-92) with the oracle
    dynamics in place of Isaac Gym.  State kept between calls: means, best trajs, U, beta,
    pending suction forces -- the reference's warm-start state (mppi.py:148-153,134,186).
    Supports sharding (rank, world_size) so gloo tests can exercise the N>1 host logic.
    """

    def __init__(self, cfg: Cfg, delta=None, scene=None, seed=0, update_cov=False):
        self.cfg = cfg
        # mppi.py:201-203, :508-516 (single-mode halton-spline only: the multi-modal update has no such branch)
        self.update_cov = bool(update_cov) and not cfg.multi_modal and not cfg.mode_simple
        self.cov_action = np.array([np.float32(cfg.scale_tril[j]) ** 2 for j in range(cfg.nu)], np.float32)
        self.sc = scene or default_scene()
        K, T, nu = cfg.K, cfg.T, cfg.nu
        self.delta = None if delta is None else f32(delta).copy()
        if self.delta is not None:
            self.delta[-1] = 0.0
        z = lambda: np.zeros((T, nu), np.float32)
        self.mean, self.mean1, self.mean2 = z(), z(), z()
        self.best, self.best1, self.best2 = z(), z(), z()
        self.U = z()
        self.beta = 1.0
        self.pend = np.zeros((K, 4), np.float32)
        self.seed, self.calls = seed, 0
        self.last = {}

    def command(self, world0):
        cfg = self.cfg
        K, T, nu = cfg.K, cfg.T, cfg.nu
        if cfg.mode_simple:
            return self._command_simple(world0)
        self.mean = shift(self.mean)
        if cfg.multi_modal:
            self.mean1, self.mean2 = shift(self.mean1), shift(self.mean2)
            self.best1, self.best2 = shift(self.best1), shift(self.best2)
        if self.delta is None:  # sampling_method == 'random' (mppi.py:386-387,481; quirk Q4)
            delta = noise_fill(cfg, self.seed, self.calls, K)
            delta[-1] = 0.0      # mppi.py:392
        else:
            delta = self.delta
        act = assemble_actions(cfg, delta, self.mean, self.mean1, self.mean2, self.best1,
                               self.best2)
        r = self._rollout(world0, act)
        w, w1, w2, info = update_weights(cfg, r["J"], self.beta)
        self.beta = info.beta
        ps = partial_sums(cfg, w, w1, w2, r["actions"])
        self.mean = mean_update(cfg, self.mean, ps[0])
        if cfg.multi_modal:
            self.mean1, self.mean2 = ps[1].copy(), ps[2].copy()
            self.best1 = r["actions"][info.best_idx_1].copy()
            self.best2 = r["actions"][K // 2 + info.best_idx_2].copy()
        else:
            self.best = r["actions"][info.best_idx].copy()
        if self.update_cov:
            d = (r["actions"] - self.mean[None]).astype(np.float64)      # delta of mppi.py:506 (new mean)
            upd = ((w.astype(np.float64)[:, None, None] * d * d).sum(axis=0).mean(axis=0)).astype(np.float32)
            f = np.float32
            self.cov_action = (f(1.0 - 0.7) * self.cov_action + f(0.7) * upd).astype(f)   # :514
            self.cov_action = (self.cov_action + f(0.005)).astype(f)                      # :515
            for j in range(nu):
                cfg.scale_tril[j] = float(np.sqrt(self.cov_action[j]))                    # :516
        action = self.mean.copy()
        top_idx, top_val = topk(w, min(20, K))
        top_trajs = r["states"][top_idx][:, :, [0, 2]]
        if cfg.filter_u:
            action = savgol9(action)
        self.calls += 1
        self.last = dict(r, w=w, w1=w1, w2=w2, info=info, top_idx=top_idx, top_trajs=top_trajs,
                         act=act, action=action)
        return action

    def _rollout(self, world0, act):
        return point_rollout(self.cfg, self.sc, world0, act, self.pend)

    def _command_simple(self, world0):
        cfg = self.cfg
        K, T, nu = cfg.K, cfg.T, cfg.nu
        self.U = np.roll(self.U, -1, axis=0)  # mppi.py:221
        noise = noise_fill(cfg, self.seed, self.calls, K)  # mppi.py:340
        lo = np.array([cfg.u_min[j] for j in range(nu)], np.float32)
        hi = np.array([cfg.u_max[j] for j in range(nu)], np.float32)
        act = np.maximum(np.minimum(self.U[None] + noise, hi), lo).astype(np.float32)
        if cfg.env_type == 1 and cfg.gripper_cmd:                      # mppi.py:346-350
            act[:, :, 7:] = 1.5 if cfg.gripper_cmd == 1 else -1.5
        r = self._rollout(world0, act)
        self.U, ct, w = simple_update(cfg, r["S"], r["actions"], self.U)
        action = self.U[:cfg.u_per_command].copy()
        top_idx, _ = topk(w, min(20, K))
        if cfg.filter_u:
            action = savgol9(action)
        self.calls += 1
        self.last = dict(r, w=w, cost_total=ct, top_idx=top_idx, act=act, action=action,
                         top_trajs=r["states"][top_idx][:, :, [0, 2]])
        return action

    def pull_preference(self):
        info = self.last["info"]
        return int(info.wsum_pull > info.wsum_push)
