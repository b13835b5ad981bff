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
    subprocess.check_call(["g++"] + HOST_FLAGS + fma_flag() + ["-I" + os.path.join(HERE, "native", "shim"),
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
