"""The product's DEVICE code of the point_env dynamics (m3p2i_aip_amd/csrc/planar_dyn.hpp -- what every rollout lane and
the step-mode kernel execute) compiled by g++ for the host, one lane per "wavefront", and run against the oracle on
random worlds: identical bits, without a GPU.  (`-m gpu` repeats this with the real kernels, tests/test_hip_parity_point.py;
here the device SOURCE is what is checked -- both of its paths: the general instance of step mode and the rollout's
dispatch to the leanest substep instance.)"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests.native_flags import host_flags

HERE = os.path.dirname(os.path.abspath(__file__))
HOST_FLAGS = ["-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-fno-fast-math"]


def fma_flag():
    """The spec's fused multiply-adds (`fmaf`) as one instruction where the host CPU has them (same bits as libm's)."""
    try:
        return ["-mfma"] if " fma " in open("/proc/cpuinfo").read() else []
    except OSError:
        return []


@pytest.fixture(scope="module")
def host_lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("planar_dyn") / "libplanar_dyn_host.so")
    subprocess.check_call(["g++"] + host_flags(HOST_FLAGS) + fma_flag() + ["-I" + os.path.join(HERE, "native", "shim"),
                           os.path.join(HERE, "native", "planar_dyn_host.cpp"), "-o", out])
    lib = C.CDLL(out)
    lib.pdh_step.argtypes = [C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    return lib


def random_worlds(O, n, rng):
    """Robot, box and dyn-obs anywhere in the arena -- overlapping each other, the obstacle, the walls and corners,
    rotated, moving, with pending suction forces (the generator of the GPU fuzz test, vectorised)."""
    w = O.init_world(n)

    def place(k):
        p = rng.uniform(-3.8, 3.8, (k, 2))
        spot = rng.integers(0, 4, k)
        for i in range(k):
            if spot[i] == 1:
                p[i, rng.integers(0, 2)] = rng.choice([-1, 1]) * rng.uniform(3.4, 3.85)
                if rng.random() < 0.5:
                    p[i] = rng.choice([-1, 1], 2) * rng.uniform(3.3, 3.8, 2)
            elif spot[i] == 2:
                p[i] = np.array([2.0, 2.0]) + rng.uniform(-0.6, 0.6, 2)
            elif spot[i] == 3:
                p[i] = np.array([0.0, 1.0]) + rng.uniform(-0.8, 0.8, 2)
        return p

    w[:, 0:2] = place(n)
    w[:, 4:6] = rng.normal(0, 1, (n, 2))
    for base in (O.W_B, O.W_D):
        yaw = rng.uniform(-np.pi, np.pi, n)
        w[:, base:base + 2] = place(n)
        w[:, base + 2], w[:, base + 3] = np.cos(yaw), np.sin(yaw)
        mv = rng.random(n) < 0.5
        w[mv, base + 4:base + 6] = rng.normal(0, 0.5, (int(mv.sum()), 2))
        w[mv, base + 6] = rng.normal(0, 1, int(mv.sum()))
    w[:, O.W_FEXT_B:O.W_FEXT_B + 2] = rng.normal(0, 100, (n, 2)) * (rng.random((n, 1)) < 0.3)
    w[:, O.W_FEXT_R:O.W_FEXT_R + 2] = rng.normal(0, 100, (n, 2)) * (rng.random((n, 1)) < 0.3)
    return w.astype(np.float32)


def compare(O, lib, seed, mode, n=1500, steps=30, dt=0.05, substeps=2, iters=6):
    rng = np.random.default_rng(seed)
    a = random_worlds(O, n, rng)
    b = a.copy()
    sc = O.default_scene()
    sc.dt, sc.substeps, sc.iters = dt, substeps, iters
    # robot (c, s, w) do not exist on the device; the rollout's path forms the dyn-obs contact force only
    cols = [c for c in range(25) if c not in (2, 3, 6)] + ([25, 26, 27, 28] if mode == 0 else []) + [29, 30]
    contacts = 0
    for t in range(steps):
        u = rng.uniform(-3, 3, (n, 2)).astype(np.float32)
        O.step_batch(sc, a, u)
        lib.pdh_step(dt, substeps, iters, b.ctypes.data, n, u.ctypes.data, mode)
        neq = a[:, cols].view(np.uint32) != b[:, cols].view(np.uint32)
        if neq.any():
            r, c = np.argwhere(neq)[0]
            raise AssertionError(f"seed {seed} mode {mode} step {t} world {r} column {cols[c]}: oracle {a[r, cols[c]]!r} "
                                 f"device-source {b[r, cols[c]]!r} ({int(neq.sum())} values differ)")
        contacts += int((np.abs(a[:, 29:31]).sum(1) > 0).sum())
    return contacts


@pytest.mark.parametrize("mode", [0, 1], ids=["step_mode_general_instance", "rollout_instance_dispatch"])
@pytest.mark.parametrize("seed", range(3))
def test_device_header_on_the_host_equals_the_oracle(oracle, host_lib, seed, mode):
    assert compare(oracle, host_lib, seed, mode) > 1000      # (and the dyn-obs really is in contact often)


@pytest.mark.parametrize("dt,substeps,iters,mode", [(0.04, 3, 4, 1), (0.05, 1, 6, 0), (0.05, 2, 8, 1), (0.02, 2, 1, 0)])
def test_device_header_on_the_host_with_other_solver_settings(oracle, host_lib, dt, substeps, iters, mode):
    compare(oracle, host_lib, 100 + substeps + iters, mode, n=600, steps=20, dt=dt, substeps=substeps, iters=iters)


# ================================================================ panda_env: csrc/panda_dyn.hpp on the host
@pytest.fixture(scope="module")
def panda_host_lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("panda_dyn") / "libpanda_dyn_host.so")
    subprocess.check_call(["g++"] + host_flags(HOST_FLAGS) + fma_flag() + ["-Wno-unknown-pragmas", "-I" + os.path.join(HERE, "native", "shim"),
                           os.path.join(HERE, "native", "panda_dyn_host.cpp"), "-o", out])
    lib = C.CDLL(out)
    VP = C.c_void_p
    lib.pnh_step.argtypes = [C.c_float, C.c_int, VP, C.c_int, VP, VP, C.c_int, VP, VP]
    lib.pnh_infer_held.argtypes = [C.c_float, C.c_int, VP, C.c_int, VP]
    return lib


@pytest.fixture(scope="module")
def P():
    import oracle.panda as P
    P.lib()
    return P


def random_panda_worlds(P, sc, n, rng):
    """Random joint configurations and velocities; cubeA on the table, near the hand or falling onto the shelf (the
    generator of the GPU fuzz test); cubeB on the table, next to / under / on top of cubeA, tilted or moving; the plate
    where it floats or near the hand; every seventh world holds the cube / has the open gripper around it / just off it;
    gripper poses with the finger tips at the table, at a cube, at the plate."""
    from tests.panda_worlds import grasp_world
    w = P.init_world(n)
    w[:, P.W_CUBEA + 2] = 1.05; w[:, P.W_CUBEB + 2] = 1.05          # resting on the table
    qlo, qhi = np.array(sc.qlo), np.array(sc.qhi)
    w[:, P.W_Q:P.W_Q + 9] = qlo + rng.uniform(0.05, 0.95, (n, 9)) * (qhi - qlo)
    w[:, P.W_QD:P.W_QD + 9] = rng.normal(0, 0.3, (n, 9)) * (rng.random((n, 1)) < 0.5)

    def quat(axis, ang):
        axis = np.asarray(axis, float) / np.linalg.norm(axis)
        return np.concatenate([axis * np.sin(ang / 2), [np.cos(ang / 2)]])

    for i in range(n):
        hand = P.fk(sc, w[i, P.W_Q:P.W_Q + 9].astype(np.float32))["pos"][8]
        k = i % 6
        if k == 0:
            w[i, P.W_CUBEA:P.W_CUBEA + 2] = rng.uniform(-0.5, 0.5, 2)
        elif k == 1:
            w[i, P.W_CUBEA:P.W_CUBEA + 3] = hand + rng.uniform(-0.12, 0.12, 3)
        elif k == 2:
            w[i, P.W_CUBEA:P.W_CUBEA + 3] = (0.5 + rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), 1.6)
        elif k == 3:      # stacked on cubeB (offsets up to beyond the edge), resting or dropped from a few mm
            w[i, P.W_CUBEA:P.W_CUBEA + 3] = w[i, P.W_CUBEB:P.W_CUBEB + 3] + (rng.uniform(-0.035, 0.035), rng.uniform(-0.035, 0.035), 0.05 + rng.choice([0.0, 0.0005, 0.004]))
            w[i, P.W_CUBEA + 3:P.W_CUBEA + 7] = quat((0, 0, 1), rng.uniform(-0.5, 0.5))
        elif k == 4:      # tumbling in the air above the table / next to cubeB
            w[i, P.W_CUBEA:P.W_CUBEA + 3] = w[i, P.W_CUBEB:P.W_CUBEB + 3] + (rng.uniform(-0.07, 0.07), rng.uniform(-0.07, 0.07), rng.uniform(0.0, 0.08))
            w[i, P.W_CUBEA + 3:P.W_CUBEA + 7] = quat(rng.normal(0, 1, 3), rng.uniform(0, 3))
            w[i, P.W_CUBEA + 7:P.W_CUBEA + 13] = rng.normal(0, 1, 6) * (0.3, 0.3, 0.3, 3, 3, 3)
        else:             # cubeB pushed / near the hand; the plate near the hand
            w[i, P.W_CUBEB:P.W_CUBEB + 3] = hand + rng.uniform(-0.1, 0.1, 3)
            w[i, P.W_CUBEB + 7:P.W_CUBEB + 10] = rng.normal(0, 0.3, 3)
            w[i, P.W_OBS:P.W_OBS + 3] = hand + rng.uniform(-0.15, 0.15, 3)
        if i % 11 == 0:   # gripper pointing down with the finger tips at the table / at cubeB's height
            z = rng.choice([1.025 + 0.1154 + rng.uniform(-0.004, 0.02), 1.05 + 0.1034 + rng.uniform(-0.01, 0.01)])
            tgt = np.array([rng.uniform(0.1, 0.4), rng.uniform(-0.3, 0.3), z])
            if rng.random() < 0.5:
                tgt[:2] = w[i, P.W_CUBEB:P.W_CUBEB + 2] + rng.uniform(-0.06, 0.06, 2)
            w[i, P.W_Q:P.W_Q + 9] = _ik_down(P, sc, tgt)
            w[i, P.W_Q + 7:P.W_Q + 9] = rng.uniform(0, 0.04, 2)
    special = [grasp_world(P, sc), grasp_world(P, sc, close_gripper=False), grasp_world(P, sc, close_gripper=False, offset=(0.0, 0.012))]
    for i in range(0, n, 7):
        w[i] = special[(i // 7) % 3]
    return w.astype(np.float32)


def _ik_down(P, sc, target):
    """joint angles that put the hand's origin at `target` with the gripper pointing straight down (damped least squares
    from the elbow-up pose the grasp helper uses; test data only)"""
    q = np.array([0, 0.3, 0, -2.2, 0, 2.5, 0.785, 0.04, 0.04], np.float64)

    def feat(L):
        return np.concatenate([L["pos"][8], 0.3 * L["az"][8], 0.3 * L["ay"][8]]).astype(np.float64)

    want = np.concatenate([target, 0.3 * np.array([0, 0, -1.0]), 0.3 * np.array([0, 1.0, 0])])
    lo, hi = np.array(sc.qlo)[:7] + 0.02, np.array(sc.qhi)[:7] - 0.02
    for _ in range(400):
        L = P.fk(sc, q.astype(np.float32))
        e = want - feat(L)
        if np.linalg.norm(e) < 1e-5:
            break
        Jm = np.zeros((9, 7))
        for j in range(7):
            dq = q.copy(); dq[j] += 1e-3
            Jm[:, j] = (feat(P.fk(sc, dq.astype(np.float32))) - feat(L)) / 1e-3
        step = Jm.T @ np.linalg.solve(Jm @ Jm.T + 1e-4 * np.eye(9), e)
        q[:7] = np.clip(q[:7] + np.clip(step, -0.2, 0.2), lo, hi)
    return q.astype(np.float32)


@pytest.mark.parametrize("mode", [0, 1, 2], ids=["step_mode", "pick_rollout_lazy_fk", "reach_rollout_lazy_fk_no_forces"])
@pytest.mark.parametrize("seed", range(2))
def test_panda_device_header_on_the_host_equals_the_oracle(P, panda_host_lib, seed, mode):
    sc = P.default_scene()
    rng = np.random.default_rng(40 + seed)
    n, steps = 330, 25
    a = random_panda_worlds(P, sc, n, rng)
    b = a.copy()
    hp, trav, obs = np.zeros((n, 3), np.float32), np.zeros(n, np.float32), np.zeros((n, 10), np.float32)
    for i in range(n):          # a world is loaded: the grasp and sleep states are inferred from the geometry
        row = np.ascontiguousarray(a[i])
        P.infer_state(sc, row)
        a[i] = row
    panda_host_lib.pnh_infer_held(0.01, 2, b.ctypes.data, n, hp.ctypes.data)
    np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
    assert a[:, P.W_HELD].sum() >= n // 21 and 0 < a[:, P.W_AWAKE].sum() < 2 * n
    # (not on the device: the plate's orientation and angular velocity; the reach rollout forms no contact forces)
    cols = [c for c in range(P.WORLD_FLOATS) if not (P.W_OBS + 3 <= c < P.W_OBS + 7) and not (P.W_OBS + 10 <= c < P.W_OBS + 13)
            and not (mode == 2 and P.W_FT <= c < P.W_FT + 9)]
    grip = rng.integers(0, 3, n)
    rows_seen = np.zeros(2, int)
    for t in range(steps):
        u = rng.uniform(-2, 2, (n, 9)).astype(np.float32)
        u[:, 7:] = rng.uniform(-1.5, 1.5, (n, 2))
        u[grip == 1, 7:] = 1.5          # gripper override open / close (m3p2i.py:10-14) for a third of the worlds each
        u[grip == 2, 7:] = -1.5
        for i in range(n):              # (one world at a time: the oracle's row counters are per call)
            row = a[i:i + 1]
            P.step_batch(sc, row, u[i:i + 1])
            rows_seen += np.minimum(P.last_rows(), 1)
        panda_host_lib.pnh_step(0.01, 2, b.ctypes.data, n, u.ctypes.data, obs.ctypes.data, mode, hp.ctypes.data, trav.ctypes.data)
        neq = a[:, cols].view(np.uint32) != b[:, cols].view(np.uint32)
        if neq.any():
            r, c = np.argwhere(neq)[0]
            raise AssertionError(f"seed {seed} mode {mode} step {t} world {r} column {cols[c]}: oracle {a[r, cols[c]]!r} "
                                 f"device-source {b[r, cols[c]]!r} ({int(neq.sum())} values differ)")
        for i in range(0, n, 13):       # what the costs read: the finger links at the final joint values
            Lk = P.fk(sc, a[i, :9])
            want = np.concatenate([Lk["pos"][9], Lk["quat"][9], Lk["pos"][10]]).astype(np.float32)
            np.testing.assert_array_equal(want.view(np.uint32), obs[i].view(np.uint32))
    assert rows_seen[0] > 200 and rows_seen[1] > 1000, rows_seen     # gripper contacts and cube contacts really occur
