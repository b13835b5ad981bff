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


def test_library_in_tree_is_built_from_this_tree():
    """m3_build_id() -- compiled into the library -- equals the hash of the kernel sources + compiler flags of THIS checkout: the
    .so that travels to the GPU box is not a stale one, and the committed PMC profiles (profiles/r06/mix_*.json, keyed by the same
    id; bench.py's roofline_valu) can be matched to it across rebuilds."""
    import json
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd import build
    build.build()
    bid = L.load().m3_build_id().decode()
    assert len(bid) == 16 and bid == build.source_hash()
    mix = json.load(open(os.path.join(ROOT, "profiles", "r06", "mix_push.json")))
    assert len(mix["build_id"]) == 16          # (whether it is THIS build's is what bench.py reports: `stale` otherwise)


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


def test_faure_permutations_known_answers_and_shape():
    """MPPIConfig.halton_scramble='faure': Faure's (1992) recursive digit permutations.  Known answers from the
    construction (pi_2 = (0 1); even b: (2 pi_{b/2}, 2 pi_{b/2} + 1); odd b = 2k + 1: pi_{2k} with values >= k
    raised by one and k inserted in the middle)."""
    from m3p2i_aip_amd import sampling as S
    assert S.faure_permutation(2) == [0, 1]
    assert S.faure_permutation(3) == [0, 1, 2]
    assert S.faure_permutation(4) == [0, 2, 1, 3]
    assert S.faure_permutation(5) == [0, 3, 2, 1, 4]
    assert S.faure_permutation(6) == [0, 2, 4, 1, 3, 5]
    assert S.faure_permutation(7) == [0, 2, 5, 3, 1, 4, 6]
    assert S.faure_permutation(8) == [0, 4, 2, 6, 1, 5, 3, 7]
    for b in S.first_primes(100):
        p = S.faure_permutation(b)
        assert sorted(p) == list(range(b)) and p[0] == 0      # a permutation that keeps digit 0 (trailing zeros stay zeros)


def test_generalized_halton_keeps_the_one_dimensional_stratification():
    """Permuting digits permutes the points of every complete block: the first b^m values of the base-b coordinate
    are exactly {j / b^m} -- with index 0 (value 0) replaced by index b^m, whose only non-zero digit maps to pi(1)/b^(m+1)."""
    from m3p2i_aip_amd import sampling as S
    for j, b in enumerate(S.first_primes(6)):
        m = 2
        n = b ** m
        u = S.halton_uniform(n - 1, j + 1, "faure")[:, j].double().numpy()     # indices 1 .. b^m - 1
        got = np.sort(np.round(u * n).astype(int))
        assert np.array_equal(got, np.arange(1, n)), (b, got)
    # default stays the reference's in-tree plain sequence (golden G8 pins it elsewhere): Appendix A row 1
    np.testing.assert_allclose(S.halton_gaussian(8, 4)[0].numpy(), [0.0, -0.4307, -0.8416, -1.0676], atol=1e-4)


def test_scrambled_halton_decorrelates_the_high_dimensions():
    """The panda_env knots are 45-dimensional (nu 9 x T//4 = 5 knots: primes up to 197).  At K = 4000 the plain
    sequence's neighbouring high dimensions are strongly correlated; Faure's permutations remove that
    (the bound a set of 4000 independent uniforms meets is ~4.5 / sqrt(K) = 0.07)."""
    from m3p2i_aip_amd import sampling as S
    K, nd = 4000, 45
    worst = {}
    for sc in ("none", "faure"):
        u = S.halton_uniform(K, nd, sc).double().numpy()
        c = np.corrcoef(u[:, 29:].T)                      # dimensions 30 .. 45
        np.fill_diagonal(c, 0.0)
        worst[sc] = float(np.abs(c).max())
        assert abs(u.mean() - 0.5) < 5e-3 and u.min() > 0 and u.max() < 1
    assert worst["none"] > 0.4, worst            # what the plain set does
    assert worst["faure"] < 0.1, worst           # the bound the scrambled set passes and the plain one fails
    with pytest.raises(ValueError):
        S.halton_uniform(8, 2, "sobol")


def test_objective_takes_a_goal_tensor_with_its_host_values_attached():
    """PLANNER_AIF_PANDA hands its goal over as a tensor that carries the host values it was made from (`_m3_host`): goal_list()
    uses them -- no read-back of the tensor in command() -- and still follows an in-place write into that tensor."""
    import types
    import torch
    from m3p2i_aip_amd.cost_functions import Objective
    from m3p2i_aip_amd import task_planner as tp
    cfg = types.SimpleNamespace(mppi=types.SimpleNamespace(device="cpu", num_samples=64), multi_modal=False, env_type="panda_env",
                                kp_suction=400, suction_active=False, pre_height_diff=0.05, task="pick", goal=[0] * 7,
                                cube_on_shelf=False)
    pl = tp.set_task_planner(cfg)
    g = pl._device_tensor(np.array([0.2, 0.2, 1.115, 0, 0, 0, 1], np.float32))
    assert g.dtype == torch.float32 and g._m3_host == [float(np.float32(x)) for x in (0.2, 0.2, 1.115, 0, 0, 0, 1)]
    o = Objective(cfg)
    o.update_objective("pick", g)
    assert o._goal_host[0] is g._m3_host and o.goal_list() == g._m3_host
    g[2] = 1.5                                    # a caller writes into it: the version counter invalidates the attached values
    assert o.goal_list()[2] == 1.5
    #  ... also when the write happens BEFORE the hand-over: the attached values are stale then and must not be used
    from m3p2i_aip_amd.cost_functions import attached_host_values
    g2 = pl._device_tensor(np.array([0.2, 0.2, 1.115, 0, 0, 0, 1], np.float32))
    assert attached_host_values(g2) is g2._m3_host
    g2[2] += 0.25
    assert attached_host_values(g2) is None
    o.update_objective("pick", g2)
    assert o.goal_list()[2] == pytest.approx(1.365, abs=1e-6)
    pl.task, pl.curr_goal = "place", g2          # check_task_success reads the goal the same way
    pl._poses = lambda sim: [np.array([0.2, 0.2, 1.0], np.float32)] * 4
    assert pl.check_task_success(None) is True


def test_new_entry_points_refuse_a_null_handle():
    from m3p2i_aip_amd import _lib as L
    lib = L.load()
    assert lib.m3_panda_lanes_per_sample_used(None) == 0
    assert lib.m3_panda_near_share(None) == -1
    assert lib.m3_set_panda_reach_cost_kernel(None, 1) < 0
    assert lib.m3_set_panda_lanes_per_sample(None, 16) < 0
