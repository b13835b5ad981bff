"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the
header declares, the init-time sampler reproduces the reference's values, scene tables."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd import build
    build.build()
    hdr = open(os.path.join(ROOT, "include", "m3p2i_hip.h")).read()
    declared = set(re.findall(r"\b(m3_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"m3_status", "m3_env", "m3_task"}
    lib = L.load()            # raises if the .so is missing -- no fallback
    bound = {n for n, _, _ in L.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    raw = ctypes.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert lib.m3_abi_version() == L.ABI_VERSION


def test_config_struct_layout_matches_header():
    """ctypes mirror of m3_config vs the C definition: defaults written by the library land in
    the right Python fields (no GPU needed: m3_default_config is host-only)."""
    from m3p2i_aip_amd import _lib as L
    lib = L.load()
    c = L.Config()
    lib.m3_default_config(ctypes.byref(c), L.ENV_POINT)
    assert (c.nu, c.T, c.K_global, c.substeps, c.solver_iters) == (2, 15, 200, 2, 6)
    assert c.dt == pytest.approx(0.05) and c.kp_suction == 400 and c.step_size_mean == pytest.approx(0.98)
    assert list(c.u_max)[:2] == [3.0, 3.0] and list(c.noise_sigma_diag)[:2] == [3.0, 3.0]
    lib.m3_default_config(ctypes.byref(c), L.ENV_PANDA)
    assert (c.nu, c.T) == (9, 12) and c.dt == pytest.approx(0.01)
    assert list(c.u_max)[6:9] == [2.0, 1.5, 1.5] and c.noise_sigma_diag[7] == pytest.approx(0.8)
    assert c.sim_only == 0 and c.seed == 0


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd.engine import HipEngine, make_config
    with pytest.raises(L.M3Error):
        HipEngine(make_config(K=64, T=12))


@pytest.mark.parametrize("shape", [(64, 12, 2), (64, 30, 2), (64, 20, 9)])
def test_g8_sampler_reproduces_reference_values(golden, shape):
    from m3p2i_aip_amd import sampling as S
    K, T, nu = shape
    kn = S.halton_gaussian(K, (T // 4) * nu).numpy()
    np.testing.assert_array_equal(kn, golden[f"g8_knots_{K}_{T}_{nu}"])
    d = S.halton_spline_delta(K, T, nu).numpy()
    np.testing.assert_allclose(d, golden[f"g8_delta_{K}_{T}_{nu}"], atol=1e-6)
    # sharded generation = rows of the global set
    np.testing.assert_array_equal(S.halton_spline_delta(K, T, nu, k0=16, k1=32).numpy(), d[16:32])
    assert S.first_primes(20) == list(golden["g8_primes"])


def test_sampler_rejects_short_horizon():
    from m3p2i_aip_amd import sampling as S
    with pytest.raises(ValueError):
        S.halton_spline_delta(8, 10, 2)   # n_knots = 2 <= degree (reference: splrep raises)


def test_scene_tables():
    from m3p2i_aip_amd import scenes
    assert len(scenes.POINT_ENV) == 11 and scenes.num_bodies("point_env") == 13
    assert len(scenes.PANDA_ENV) == 7 and scenes.num_bodies("panda_env") == 17
    assert scenes.POINT_ENV[-1].type == "robot"       # skill_utils.py:89-90 needs robot last
    assert scenes.body_index("point_env", "point_robot", "link_y") == 12
    assert scenes.body_index("point_env", "box", "box") == scenes.actor_index("point_env", "box") == 6
    assert scenes.body_index("panda_env", "panda", "panda_leftfinger") == 15
    assert scenes.body_index("panda_env", "cubeA", "box") == 4


def test_objective_goal_host_copy_is_cached_until_the_goal_changes():
    """Objective.goal_list() feeds the C-ABI by value; it must not re-read the goal tensor at
    every command (a device read is a stream sync) yet must follow every way a caller can change
    the goal: a new tensor, a new list, or an in-place write into the same tensor."""
    import types
    import torch
    from m3p2i_aip_amd.cost_functions import Objective
    cfg = types.SimpleNamespace(mppi=types.SimpleNamespace(device="cpu", num_samples=64), multi_modal=False, env_type="point_env",
                                kp_suction=400, suction_active=True, pre_height_diff=0.0, task="push", goal=[0, 0],
                                cube_on_shelf=False)
    o = Objective(cfg)
    g = torch.tensor([1.0, 2.0])
    o.update_objective("push", g)
    assert o.goal_list() == [1.0, 2.0]
    first = o._goal_host
    o.update_objective("push", g)                 # same tensor again (every tick of reactive_tamp.py)
    assert o.goal_list() == [1.0, 2.0] and o._goal_host is first
    g[0] = 5.0                                    # in-place edit: torch bumps the version counter
    assert o.goal_list() == [5.0, 2.0]
    o.update_objective("pull", torch.tensor([7.0, 8.0]))
    assert o.goal_list() == [7.0, 8.0]
    o.update_objective("pull", [3.0, 4.0])        # python list
    assert o.goal_list() == [3.0, 4.0]


def test_non_diagonal_noise_sigma_uses_its_diagonal_in_halton_spline_mode():
    """mppi.py:175-176: scale_tril = sqrt(diagonal(noise_sigma)) -- the reference's halton-spline path never
    reads the off-diagonal entries, so a full matrix plans exactly like its diagonal; the whole matrix travels to the
    library (noise_sigma_full) for the modes that sample from MultivariateNormal(noise_sigma)."""
    from types import SimpleNamespace
    import pytest
    import torch
    from m3p2i_aip_amd import planner as P
    from tests.oracle_engine import OracleEngine
    old = P.ENGINE_CLS
    P.ENGINE_CLS = OracleEngine
    try:
        def make(sigma, **kw):
            m = P.MPPIConfig(num_samples=64, horizon=12, nx=4, device="cpu", u_min=[-3.0, -3.0], u_max=[3.0, 3.0],
                             noise_sigma=sigma, u_per_command=12, sample_null_action=True, filter_u=True, fused=True, **kw)
            return P.M3P2I(SimpleNamespace(env_type="point_env", multi_modal=False, suction_active=False, kp_suction=0,
                                           pre_height_diff=0.0, task="push", goal=[0.0, 0.0], cube_on_shelf=False, mppi=m))
        full, diag = make([[3.0, 0.7], [0.7, 2.0]]), make([[3.0, 0.0], [0.0, 2.0]])
        assert torch.equal(full.scale_tril, diag.scale_tril)
        c = full._engine.cfg
        assert list(c.noise_sigma_diag)[:2] == [3.0, 2.0] and c.full_sigma == 1 and diag._engine.cfg.full_sigma == 0
        assert [round(v, 6) for v in list(c.noise_sigma_full)[:4]] == [3.0, 0.7, 0.7, 2.0]
    finally:
        P.ENGINE_CLS = old
