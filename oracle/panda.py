"""CPU ORACLE (test infrastructure) -- ctypes bindings of the panda_env part of libm3oracle.so
(oracle/panda_chain.c): chain spec v1 dynamics + the reference's reach/pick/place costs."""
from __future__ import annotations

import ctypes as C

import numpy as np

import oracle as O

WORLD_FLOATS = 84
# offsets inside one world row (m3o_panda_world): q9 qd9 | cubeA13 cubeB13 obs13 (pos3 quat4 vel3 angvel3) | held |
# rel_p3 rel_q4 | awake[cubeA, cubeB] | f_table3 f_shelf3 f_cubeB3
W_Q, W_QD, W_CUBEA, W_CUBEB, W_OBS, W_HELD, W_RELP, W_RELQ, W_AWAKE, W_FT, W_FS, W_FB, W_WARM = 0, 9, 18, 31, 44, 57, 58, 61, 65, 67, 70, 73, 76
RAW_FLOATS = 57     # what the engine's set_world_panda_raw takes: q9 qd9 cubeA13 cubeB13 obs13


def raw57(world):
    """the first 57 floats of a world row: the state the wrapper's tensors carry (no held / sleep flags, no forces)"""
    return np.asarray(world, np.float32).reshape(-1)[:RAW_FLOATS].copy()

OBS_FLOATS = 30
LINK_NAMES = ["panda_link0", "panda_link1", "panda_link2", "panda_link3", "panda_link4", "panda_link5",
              "panda_link6", "panda_link7", "panda_hand", "panda_leftfinger", "panda_rightfinger"]


class PandaScene(C.Structure):
    _fields_ = [("dt", C.c_float), ("substeps", C.c_int), ("g", C.c_float), ("base", C.c_float * 3),
                ("drive_damping", C.c_float), ("inertia", C.c_float * 9), ("effort", C.c_float * 9),
                ("vlim", C.c_float * 9), ("qlo", C.c_float * 9), ("qhi", C.c_float * 9),
                ("table", C.c_float * 6), ("shelf", C.c_float * 6), ("cube_half", C.c_float),
                ("cube_m", C.c_float), ("cube_mu", C.c_float), ("grasp_z", C.c_float),
                ("grasp_dx", C.c_float), ("grasp_dz", C.c_float), ("grasp_align", C.c_float),
                ("grasp_tol", C.c_float), ("k_contact", C.c_float), ("tip_z", C.c_float),
                ("tip_r", C.c_float), ("hand_z", C.c_float), ("hand_r", C.c_float),
                ("iters", C.c_int), ("contact_offset", C.c_float), ("slop", C.c_float), ("baumgarte", C.c_float),
                ("max_bias", C.c_float), ("act_margin", C.c_float), ("mu", C.c_float), ("obs_half", C.c_float * 3),
                ("obs_m", C.c_float), ("sleep_v", C.c_float), ("sleep_w", C.c_float), ("rest_gap", C.c_float)]


_bound = False


def lib():
    global _bound
    l = O.load()
    if not _bound:
        FP = C.POINTER(C.c_float)
        l.m3o_sincos.argtypes = [C.c_float, FP, FP]
        l.m3o_panda_scene_default.argtypes = [C.POINTER(PandaScene)]
        l.m3o_panda_world_init.argtypes = [FP, C.c_int]
        l.m3o_panda_fk.argtypes = [C.POINTER(PandaScene), FP, FP]
        l.m3o_panda_step_batch.argtypes = [C.POINTER(PandaScene), FP, C.c_int, FP]
        l.m3o_panda_observe.argtypes = [C.POINTER(PandaScene), FP, FP]
        l.m3o_panda_cost_obs_batch.argtypes = [C.POINTER(O.Cfg), FP, C.c_int, C.c_int, FP]
        l.m3o_panda_infer_held.argtypes = [C.POINTER(PandaScene), FP]
        l.m3o_panda_last_rows.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
        l.m3o_panda_rollout.argtypes = [C.POINTER(O.Cfg), C.POINTER(PandaScene), FP, FP, C.c_int, C.c_int,
                                        FP, FP, FP, FP]
        _bound = True
    return l


def default_scene() -> PandaScene:
    sc = PandaScene()
    lib().m3o_panda_scene_default(C.byref(sc))
    return sc


def init_world(n=1, cube_on_shelf=False) -> np.ndarray:
    w = np.zeros(WORLD_FLOATS, np.float32)
    lib().m3o_panda_world_init(O._fp(w), int(cube_on_shelf))
    return np.tile(w, (n, 1)).astype(np.float32)


def infer_state(sc, world):
    """held + the cubes' sleep flags from the geometry, as a world load does (in place on a float32 row)"""
    assert world.dtype == np.float32 and world.flags.c_contiguous
    lib().m3o_panda_infer_held(C.byref(sc), O._fp(world))
    return world


def last_rows():
    """(gripper contacts, corner contacts) of the last substep of this thread's last single step"""
    a, b = C.c_int(), C.c_int()
    lib().m3o_panda_last_rows(C.byref(a), C.byref(b))
    return a.value, b.value


def sincos(x):
    s, c = C.c_float(), C.c_float()
    lib().m3o_sincos(float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def fk(sc, q):
    """dict(pos[11,3], quat[11,4], ax, ay, az [11,3])"""
    buf = np.zeros(11 * 16, np.float32)
    lib().m3o_panda_fk(C.byref(sc), O._fp(O.f32(q)), O._fp(buf))
    return dict(pos=buf[0:33].reshape(11, 3), quat=buf[33:77].reshape(11, 4),
                ax=buf[77:110].reshape(11, 3), ay=buf[110:143].reshape(11, 3),
                az=buf[143:176].reshape(11, 3))


def step_batch(sc, worlds, u):
    assert worlds.dtype == np.float32 and worlds.flags.c_contiguous and worlds.shape[1] == WORLD_FLOATS
    lib().m3o_panda_step_batch(C.byref(sc), O._fp(worlds), worlds.shape[0], O._fp(O.f32(u)))


def observe(sc, world):
    o = np.zeros(OBS_FLOATS, np.float32)
    lib().m3o_panda_observe(C.byref(sc), O._fp(O.f32(world)), O._fp(o))
    return o


def make_obs(left, left_q, right, cube, cube_q, cube0, cube_q_half0, f_table, f_shelf, f_cubeB):
    n = left.shape[0]
    o = np.zeros((n, OBS_FLOATS), np.float32)
    o[:, 0:3], o[:, 3:7], o[:, 7:10] = left, left_q, right
    o[:, 10:13], o[:, 13:17] = cube, cube_q
    o[:, 17:20], o[:, 20:24] = cube0, cube_q_half0
    o[:, 24:26], o[:, 26:28], o[:, 28:30] = f_table, f_shelf, f_cubeB
    return o


def cost_obs(cfg, obs, k0=0):
    obs = O.f32(obs)
    c = np.zeros(obs.shape[0], np.float32)
    lib().m3o_panda_cost_obs_batch(C.byref(cfg), O._fp(obs), obs.shape[0], k0, O._fp(c))
    return c


def make_cfg(K, T, multi_modal=False, task="reach", goal=(0, 0, 0, 0, 0, 0, 1), gripper_cmd=1, **kw):
    return O.make_cfg(K, T, 9, multi_modal=multi_modal, env_type="panda_env", task=task, goal=goal,
                      u_min=[-2.0] * 7 + [-1.5] * 2, u_max=[2.0] * 7 + [1.5] * 2,
                      noise_sigma_diag=[10.0] * 7 + [0.8] * 2, gripper_cmd=gripper_cmd,
                      pre_height_diff=0.05, **kw)


def rollout(cfg, sc, world0, act, k0=0, k1=None):
    k1 = cfg.K if k1 is None else k1
    n = k1 - k0
    states = np.zeros((n, cfg.T, 4), np.float32)
    actions = np.zeros((n, cfg.T, 9), np.float32)
    cost_h = np.zeros((n, cfg.T), np.float32)
    J = np.zeros(n, np.float32)
    w0 = O.f32(world0).reshape(-1)[:WORLD_FLOATS].copy()
    lib().m3o_panda_rollout(C.byref(cfg), C.byref(sc), O._fp(w0), O._fp(O.f32(act)), k0, k1,
                            O._fp(states), O._fp(actions), O._fp(cost_h), O._fp(J))
    return dict(states=states, actions=actions, cost_h=cost_h, J=J)


class OraclePandaPlanner(O.OraclePointPlanner):
    """command() for panda_env on the oracle: OraclePointPlanner (mppi.py:211-264 + m3p2i.py:66-92, halton-spline
    with a noise table or the in-kernel stream, simple mode, update_cov) with the chain rollout in place of the
    planar one."""

    def __init__(self, cfg, delta=None, scene=None, seed=0, update_cov=False):
        super().__init__(cfg, delta, scene or default_scene(), seed=seed, update_cov=update_cov)

    def _rollout(self, world0, act):
        r = rollout(self.cfg, self.sc, world0, act)
        s = np.zeros(self.cfg.K, np.float32)
        for t in range(self.cfg.T):          # mppi.py:309: cost_samples += c, in order
            s = (s + r["cost_h"][:, t]).astype(np.float32)
        r["S"] = s
        return r
