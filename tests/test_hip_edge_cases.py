"""GPU edge cases for the point_env path: ragged sample counts (K not a multiple of the
wavefront), minimum K (torch.topk(20)), odd K in multi-modal mode (half_K = int(K/2)), minimum
horizon for the filter, no filter with a very short horizon, u_scale != 1, no null action,
degenerate cost inputs (robot exactly on the box -> NaN cos_theta handled like torch), and a
large-K sanity run.  HIP vs the CPU oracle on identical inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def raw_world(w31):
    w = np.asarray(w31, np.float32)
    return np.concatenate([w[[0, 1, 4, 5]], w[7:14], w[14:21]])


CASES = [
    dict(K=20, T=12, task="push"),                                   # minimum K
    dict(K=65, T=12, task="pull", goal=(0.0, 0.0)),                  # one full wave + 1 lane
    dict(K=127, T=30, task="push_pull", multi_modal=True),           # odd K: halves 63 / 64
    dict(K=130, T=9, task="navigation", goal=(-3.0, 3.0)),           # T = filter window
    dict(K=64, T=3, task="push", filter_u=False),                    # short horizon, no filter
    dict(K=96, T=12, task="push", u_scale=0.5),
    dict(K=96, T=12, task="pull", goal=(0.0, 0.0), sample_null_action=False),
    dict(K=256, T=12, task="push", gamma=1.0),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items() if k != "goal"))
def test_hip_equals_oracle_on_edge_configs(oracle, case):
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd.engine import HipEngine, make_config
    c = dict(case)
    K, T, task = c.pop("K"), c.pop("T"), c.pop("task")
    goal = c.pop("goal", (-1.0, -1.0))
    mm = c.get("multi_modal", False)
    rng = np.random.default_rng(K * 31 + T)
    delta = (rng.standard_normal((K, T, 2)) * 1.3).astype(np.float32)
    w0 = oracle.init_world(1)[0]
    w0[0:2] = (0.05, 1.55)
    ocfg = oracle.make_cfg(K, T, 2, task=task, goal=goal, multi_modal=mm, filter_u=c.get("filter_u", True),
                           u_scale=c.get("u_scale", 1.0), gamma=c.get("gamma", 0.95),
                           sample_null_action=c.get("sample_null_action", True))
    opl = oracle.OraclePointPlanner(ocfg, delta)
    eng = HipEngine(make_config(K=K, T=T, nu=2, multi_modal=mm, filter_u=c.get("filter_u", True),
                                u_scale=c.get("u_scale", 1.0), gamma=c.get("gamma", 0.95),
                                sample_null_action=c.get("sample_null_action", True),
                                u_min=[-3, -3], u_max=[3, 3], noise_sigma_diag=[3, 3]))
    eng.set_objective(task, goal)
    eng.set_noise(delta)
    eng.set_world_point_raw(raw_world(w0))
    for call in range(3):
        a = eng.command(sync_host=True)
        b = opl.command(w0)
        if call == 0:
            np.testing.assert_array_equal(eng.states.cpu().numpy(), opl.last["states"])
            np.testing.assert_array_equal(eng.actions.cpu().numpy(), opl.last["actions"])
            np.testing.assert_array_equal(eng.cost_horizon.cpu().numpy(), opl.last["cost_h"])
        np.testing.assert_allclose(a, b, atol=1e-3)
        w = eng.buffer(L.BUF_WEIGHTS).cpu().numpy()
        np.testing.assert_allclose(w, opl.last["w"], rtol=2e-3, atol=1e-6)
        assert abs(w.sum() - 1.0) < 1e-4
        # top-20 by weight: when fewer than 20 weights are non-zero, torch.topk may return ANY of
        # the zero-weight samples (the library orders those by cost), so compare the weights
        top_h = eng.buffer(L.BUF_TOP_IDX).cpu().numpy()
        np.testing.assert_allclose(np.sort(opl.last["w"][top_h]), np.sort(opl.last["w"][opl.last["top_idx"]]),
                                   atol=1e-6)
    eng.close()


def test_degenerate_cost_inputs_match_torch_semantics(oracle):
    """Robot exactly on the box centre / box exactly on the goal: cos_theta is NaN in the
    reference and the align term then contributes 0 (`cos_theta > 0` is False for NaN)."""
    from m3p2i_aip_amd.engine import HipEngine, make_config
    K, T = 64, 12
    delta = np.zeros((K, T, 2), np.float32)
    for task, goal in (("push", (0.0, 2.0)), ("pull", (0.0, 2.0))):
        w0 = oracle.init_world(1)[0]
        w0[0:2] = (0.0, 2.0)      # robot on the box centre, box on the goal
        opl = oracle.OraclePointPlanner(oracle.make_cfg(K, T, 2, task=task, goal=goal), delta)
        eng = HipEngine(make_config(K=K, T=T, nu=2, u_min=[-3, -3], u_max=[3, 3], noise_sigma_diag=[3, 3]))
        eng.set_objective(task, goal)
        eng.set_noise(delta)
        eng.set_world_point_raw(raw_world(w0))
        a = eng.command(sync_host=True)
        b = opl.command(w0)
        ch = eng.cost_horizon.cpu().numpy()
        assert np.isfinite(ch).all()
        np.testing.assert_array_equal(ch, opl.last["cost_h"])
        np.testing.assert_allclose(a, b, atol=1e-3)
        eng.close()


@pytest.mark.parametrize("K", [3000, 9000])
def test_topk_with_massive_ties(oracle, K):
    """All samples identical (zero noise) -> every cost ties: the threshold filter of the top-k
    selection overflows its LDS list and the argmin-round fallback runs; ties resolve towards the
    lower sample index, and the weights are uniform."""
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd.engine import HipEngine, make_config
    T = 12
    eng = HipEngine(make_config(K=K, T=T, nu=2, u_min=[-3, -3], u_max=[3, 3], noise_sigma_diag=[3, 3],
                                sample_null_action=False))
    eng.set_objective("navigation", (-3.0, 3.0))
    eng.set_noise(np.zeros((K, T, 2), np.float32))
    eng.set_world_point_raw(raw_world(oracle.init_world(1)[0]))
    eng.command(sync_host=True)
    J = eng.buffer(L.BUF_TRAJ_COST).cpu().numpy()
    assert (J == J[0]).all()
    np.testing.assert_array_equal(eng.buffer(L.BUF_TOP_IDX).cpu().numpy(), np.arange(20))
    np.testing.assert_allclose(eng.buffer(L.BUF_WEIGHTS).cpu().numpy(), 1.0 / K, rtol=1e-4)
    tt = eng.buffer(L.BUF_TOP_TRAJS).cpu().numpy()
    st = eng.states.cpu().numpy()
    np.testing.assert_array_equal(tt, st[:20][:, :, [0, 2]])
    eng.close()


def test_large_k_sanity():
    """K = 131072 samples x T = 30 on one GPU: finite plan, normalised weights, sorted top-k."""
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd.engine import HipEngine, make_config
    K, T = 131072, 30
    eng = HipEngine(make_config(K=K, T=T, nu=2, sampling_random=True, u_min=[-3, -3], u_max=[3, 3],
                                noise_sigma_diag=[3, 3], seed=3))
    eng.set_objective("push", (-1.0, -1.0))
    for _ in range(2):
        a = eng.command(sync_host=True)
    assert np.isfinite(a).all() and np.abs(a).max() <= 4.0   # the savgol filter may overshoot u_max
    w = eng.buffer(L.BUF_WEIGHTS)
    assert abs(w.sum().item() - 1.0) < 1e-3
    J = eng.buffer(L.BUF_TRAJ_COST)
    top = eng.buffer(L.BUF_TOP_IDX).to(torch.int64)
    ref = torch.topk(-J, 20).indices
    assert torch.equal(torch.sort(J[top]).values, torch.sort(J[ref]).values)
    st = eng.states
    assert st.shape == (K, T, 4) and torch.isfinite(st).all()
    eng.close()


def _engine(**kw):
    from m3p2i_aip_amd.engine import HipEngine, make_config
    return HipEngine(make_config(**kw))


def _noise(K, T, nu, seed=3):
    g = torch.Generator().manual_seed(seed)
    knots = torch.randn(K, nu, T // 4, generator=g)
    return torch.nn.functional.interpolate(knots, size=T, mode="linear", align_corners=True).permute(0, 2, 1).contiguous().numpy()


def _three_commands(eng, setters, full=False):
    """Three warm-started commands of `eng` after applying `setters` (engine method name -> argument): what the update
    decided and produced, as host values."""
    from m3p2i_aip_amd import _lib as L
    for name, value in setters.items():
        getattr(eng, name)(value)
    out = []
    for _ in range(3):
        a = eng.command(sync_host=True)
        i = eng.info()
        d = dict(iters=[i.iters, i.iters_1, i.iters_2], eta=[i.eta, i.eta_1, i.eta_2], action=a.copy(),
                 w=eng.buffer(L.BUF_WEIGHTS).cpu().numpy().copy())
        if full:
            d.update(best=[i.best_idx, i.best_idx_1, i.best_idx_2], pref=i.pull_preference,
                     top=eng.buffer(L.BUF_TOP_IDX).cpu().numpy().tolist(), w1=eng.buffer(L.BUF_WEIGHTS_1).cpu().numpy().copy(),
                     m1=eng.buffer(L.BUF_MEAN_1).cpu().numpy().copy(), m2=eng.buffer(L.BUF_MEAN_2).cpu().numpy().copy())
        out.append(d)
    eng.close()
    return out


def test_ladder_exchange_give_up_branch_makes_the_same_decisions():
    """k_update_small (multi-modal, K <= 8192): the T column workgroups share the beta-ladder evaluations
    through memory and wait for each other with a BOUNDED spin; a workgroup whose wait runs out (other
    kernels occupying the CUs) runs all its search passes itself.  m3_set_ladder_spins(h, 0) forces that branch
    in every workgroup: iteration counts, weights and plan must equal the normal run's.  (Until round 5 the switch was an
    environment variable of the library and every run a process of its own.)"""
    K, T = 4000, 30
    delta = _noise(K, T, 2)

    def run(setters):
        eng = _engine(K=K, T=T, nu=2, multi_modal=True, u_min=[-3, -3], u_max=[3, 3], noise_sigma_diag=[3, 3])
        eng.set_objective("push_pull", (-3.75, -3.75))
        eng.set_noise(delta)
        eng.set_world_point_raw(np.array([0.0, 1.5, 0, 0, 0, 2, 1, 0, 0, 0, 0, -2, 2, 1, 0, 0, 0, 0], np.float32))
        return _three_commands(eng, setters)

    normal, gave_up = run({}), run({"set_ladder_spins": 0})
    for a, b in zip(normal, gave_up):
        assert a["iters"] == b["iters"] and min(a["iters"]) > 1
        np.testing.assert_allclose(a["eta"], b["eta"], rtol=1e-6)
        np.testing.assert_allclose(a["w"], b["w"], rtol=1e-5, atol=1e-12)
        np.testing.assert_allclose(a["action"], b["action"], atol=1e-6)


@pytest.mark.parametrize("env", ["point", "panda"])
def test_three_launch_update_equals_the_five_launch_one_and_its_give_up_branch(env):
    """Unsharded multi-modal command() with K beyond k_update_small's range: k_ladder_search (ladder workgroups + ONE search
    workgroup that waits for their flags; the minima from the rows the rollout's workgroups left behind) ->
    k_regen_part<false> -> k_regen_done<false>, against round 3's five launches (m3_set_update_launches(h, 5): k_mins,
    k_ladder, k_search, k_apply_weights, k_wsum) and against its own give-up branch (m3_set_ladder_spins(h, 0): the search
    workgroup does not wait and runs the reference's iterative passes over the costs): same pass counts and best samples,
    weights / plan to rounding -- the partial tables are added in another order, the sums in 2048-sample chunks."""
    if env == "point":
        K, T, nu = 20000, 30, 2
    else:
        K, T, nu = 6000, 20, 9
    delta = _noise(K, T, nu)

    def run(setters):
        if env == "point":
            eng = _engine(K=K, T=T, nu=2, multi_modal=True, u_min=[-3, -3], u_max=[3, 3], noise_sigma_diag=[3, 3])
            eng.set_objective("push_pull", (-3.75, -3.75))
            eng.set_world_point_raw(np.array([0.0, 1.5, 0, 0, 0, 2, 1, 0, 0, 0, 0, -2, 2, 1, 0, 0, 0, 0], np.float32))
        else:
            eng = _engine(K=K, T=T, nu=9, env_type="panda_env", multi_modal=True, u_min=[-1.2] * 9, u_max=[1.2] * 9,
                          noise_sigma_diag=[10.0] * 7 + [0.8, 0.8], lambda_=0.05, pre_height_diff=0.05, dt=0.01)
            eng.set_objective("reach", [0.2, 0.2, 1.115, 0, 0, 0, 1], gripper_cmd=1)
        eng.set_noise(delta)
        return _three_commands(eng, setters, full=True)

    three, five, gave_up = run({}), run({"set_update_launches": 5}), run({"set_ladder_spins": 0})
    for other in (five, gave_up):
        # (the first command: the same costs bit for bit; later ones start from plans that agree to rounding)
        a, b = three[0], other[0]
        assert a["iters"] == b["iters"] and a["best"] == b["best"] and a["pref"] == b["pref"] and a["top"] == b["top"]
        np.testing.assert_allclose(a["eta"], b["eta"], rtol=1e-5)
        np.testing.assert_allclose(a["w"], b["w"], rtol=2e-5, atol=1e-12)
        np.testing.assert_allclose(a["w1"], b["w1"], rtol=2e-5, atol=1e-12)
        for key in ("action", "m1", "m2"):
            np.testing.assert_allclose(a[key], b[key], atol=2e-5)
        for a, b in zip(three[1:], other[1:]):
            np.testing.assert_allclose(a["action"], b["action"], atol=1e-3)
    assert min(three[0]["iters"]) >= 1 and (env == "panda" or max(three[0]["iters"]) > 1)   # (the point searches walk their ladders)


@pytest.mark.parametrize("dt,substeps,iters", [(0.04, 3, 4), (0.05, 1, 6), (0.05, 2, 8), (0.02, 2, 1)])
def test_rollout_bit_exact_with_other_solver_settings(oracle, dt, substeps, iters):
    """isaacgym/point.yaml dt, isaacgym_wrapper.py:10 substeps and :28 solver iterations are run-time settings of the
    library (m3_config.dt / substeps / solver_iters): the kernels specialise the reference's values (six passes,
    unrolled) and keep generic loops for everything else -- both against the oracle, bit for bit."""
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd.engine import HipEngine, make_config
    K, T = 512, 20
    rng = np.random.default_rng(int(dt * 1000) + 10 * substeps + iters)
    delta = (rng.standard_normal((K, T, 2)) * 1.2).astype(np.float32)
    sc = oracle.default_scene()
    sc.dt, sc.substeps, sc.iters = dt, substeps, iters
    w = oracle.init_world(1)[0]
    w[0:2] = (0.1, 1.45)                         # next to the box: contacts from the first step
    for task, goal, mm in (("push", (-1.0, 3.0), False), ("push_pull", (-3.75, -3.75), True)):
        opl = oracle.OraclePointPlanner(oracle.make_cfg(K, T, 2, task=task, goal=goal, multi_modal=mm), delta, scene=sc)
        opl.command(w)
        eng = HipEngine(make_config(K=K, T=T, nu=2, multi_modal=mm, u_min=[-3, -3], u_max=[3, 3], noise_sigma_diag=[3, 3],
                                    dt=dt, substeps=substeps, solver_iters=iters))
        eng.set_objective(task, goal)
        eng.set_noise(delta)
        eng.set_world_point_raw(np.concatenate([w[[0, 1, 4, 5]], w[7:14], w[14:21]]))
        eng.command()
        np.testing.assert_array_equal(eng.states.cpu().numpy(), opl.last["states"])
        np.testing.assert_array_equal(eng.cost_horizon.cpu().numpy(), opl.last["cost_h"])
        np.testing.assert_array_equal(eng.buffer(L.BUF_TRAJ_COST).cpu().numpy(), opl.last["J"])
        eng.close()
