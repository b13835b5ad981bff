"""GPU parity tests for the point_env hot path: HIP kernels (through the C-ABI of
libm3p2i_hip.so) vs the CPU oracle and vs golden traces produced by the reference's own
planner code (tests/golden/make_golden.py).  Run with ``pytest -m gpu``.

Bars (BASELINE.json north_star): 1e-3 on trajectory cost and control output.  The rollout
itself is checked far tighter: the dynamics are specified as a fixed sequence of IEEE f32
operations, so HIP and oracle states must agree bit-for-bit.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _engine(**kw):
    from m3p2i_aip_amd.engine import HipEngine, make_config
    return HipEngine(make_config(**kw))


def raw_world(w31):
    """oracle world row (31 floats) -> library internal 18 floats."""
    w = np.asarray(w31, np.float32)
    return np.concatenate([w[[0, 1, 4, 5]], w[7:14], w[14:21]])


G9 = {
    "push": dict(K=256, T=30, task="push", goal=(-1.0, -1.0)),
    "pushc": dict(K=256, T=30, task="push", goal=(-1.0, 3.0)),
    "pull": dict(K=256, T=30, task="pull", goal=(0.0, 0.0)),
    "hybrid": dict(K=256, T=30, task="push_pull", goal=(-3.75, -3.75), multi_modal=True),
    "nav": dict(K=100, T=10, task="navigation", goal=(-3.0, 3.0), mode_simple=True,
                u_per_command=10, lambda_=0.5),
    "navr": dict(K=128, T=12, task="navigation", goal=(-3.0, 3.0)),
    # the MPPIConfig switches no shipped config turns on (reference traces: make_golden.py g11)
    "opt_uscale": dict(K=256, T=30, task="push", goal=(-1.0, 3.0), u_scale=0.5),
    "opt_cov": dict(K=256, T=30, task="push", goal=(-1.0, 3.0), update_cov=True),
    "opt_abs": dict(K=100, T=10, task="navigation", goal=(-3.0, 3.0), mode_simple=True, u_per_command=10,
                    lambda_=0.5, u_scale=0.8, noise_mu=[0.3, -0.2], noise_sigma=[[3.0, 1.0], [1.0, 2.0]],
                    noise_abs_cost=True),
    "opt_navr": dict(K=128, T=12, task="navigation", goal=(-3.0, 3.0), noise_mu=[0.3, -0.2],
                     noise_sigma=[[3.0, 1.0], [1.0, 2.0]]),
}


def make_pair(oracle, golden, tag, seed=7):
    """(HIP engine, oracle planner) configured identically for golden trace `tag`."""
    kw = dict(G9[tag])
    task, goal = kw.pop("task"), kw.pop("goal")
    K, T = kw.pop("K"), kw.pop("T")
    delta = golden[f"g9_{tag}_delta"] if f"g9_{tag}_delta" in golden else None
    update_cov = kw.pop("update_cov", False)
    ocfg = oracle.make_cfg(K, T, 2, task=task, goal=goal, **kw)
    opl = oracle.OraclePointPlanner(ocfg, delta, seed=seed, update_cov=update_cov)
    eng = _engine(K=K, T=T, nu=2, multi_modal=kw.get("multi_modal", False),
                  mode_simple=kw.get("mode_simple", False), sampling_random=delta is None,
                  u_per_command=kw.get("u_per_command"), lambda_=kw.get("lambda_", 1.0),
                  u_min=[-3, -3], u_max=[3, 3], noise_sigma_diag=[3, 3], seed=seed,
                  u_scale=kw.get("u_scale", 1.0), noise_mu=kw.get("noise_mu"), noise_sigma=kw.get("noise_sigma"),
                  noise_abs_cost=kw.get("noise_abs_cost", False), update_cov=update_cov)
    eng.set_objective(task, goal)
    if delta is not None:
        eng.set_noise(delta)
    return eng, opl


@pytest.mark.parametrize("tag", list(G9))
def test_command_traces_vs_reference_and_oracle(golden, oracle, tag):
    eng, opl = make_pair(oracle, golden, tag)
    worlds = golden[f"g9_{tag}_world"]
    from m3p2i_aip_amd import _lib as L
    for call in range(worlds.shape[0]):
        eng.set_world_point_raw(raw_world(worlds[call]))
        a_hip = eng.command(sync_host=True)
        a_orc = opl.command(worlds[call])
        a_ref = golden[f"g9_{tag}_action"][call]
        rows = a_ref.shape[0]
        # control output: HIP vs reference-generated golden and vs oracle (bar 1e-3)
        np.testing.assert_allclose(a_hip[:rows], a_ref, atol=1e-3, err_msg=f"{tag} call {call} vs reference")
        np.testing.assert_allclose(a_hip[:rows], a_orc, atol=1e-3, err_msg=f"{tag} call {call} vs oracle")
        w_hip = eng.buffer(L.BUF_WEIGHTS).cpu().numpy()
        from tests.conftest import assert_close_but_few
        # (the golden weights are the REFERENCE's, torch arithmetic: one or two of the 256 rollouts may take another
        # contact history from the second command on -- conftest.assert_close_but_few)
        assert_close_but_few(w_hip, golden[f"g9_{tag}_weights"][call], rtol=2e-3, atol=1e-6, frac=0.0 if call == 0 else 0.01,
                             cap=1e-3, err_msg=f"{tag} call {call} weights")
        if call == 0:
            # identical inputs on the first call: rollout must be bit-identical to the oracle
            st = eng.states.cpu().numpy()
            ac = eng.actions.cpu().numpy()
            if f"g9_{tag}_delta" in golden:
                np.testing.assert_array_equal(ac, opl.last["actions"])
                np.testing.assert_array_equal(st, opl.last["states"])
                np.testing.assert_array_equal(eng.cost_horizon.cpu().numpy(), opl.last["cost_h"])
                np.testing.assert_array_equal(eng.buffer(L.BUF_TRAJ_COST).cpu().numpy(),
                                              opl.last["J"])
            else:
                # in-kernel noise: logf/sinf/cosf of the device library and glibc differ in
                # the last bit, so the sampled actions agree to ~1 ulp, not bit-for-bit
                np.testing.assert_allclose(ac, opl.last["actions"], atol=2e-6)
                np.testing.assert_allclose(st, opl.last["states"], atol=1e-4)
                np.testing.assert_allclose(eng.cost_horizon.cpu().numpy(), opl.last["cost_h"],
                                           rtol=1e-5, atol=1e-4)
        if f"g9_{tag}_extra" in golden:     # update_cov: scale_tril after the call (mppi.py:516)
            np.testing.assert_allclose(eng.buffer(L.BUF_COV).cpu().numpy()[1], golden[f"g9_{tag}_extra"][call], rtol=1e-4)
            np.testing.assert_allclose(eng.buffer(L.BUF_COV).cpu().numpy()[1], [opl.cfg.scale_tril[j] for j in range(2)],
                                       rtol=1e-4)
        info = eng.info()
        if G9[tag].get("multi_modal"):
            assert info.pull_preference == int(golden[f"g9_{tag}_pref"][call])
            oi = opl.last["info"]
            assert (info.iters_1, info.iters_2, info.iters) == (oi.iters_1, oi.iters_2, oi.iters)
    eng.close()


@pytest.mark.parametrize("task,goal,mm,avoid", [("push", (-1, -1), False, False), ("pull", (0, 0), False, False),
                                                 ("push_pull", (-3.75, -3.75), True, False),
                                                 ("navigation", (-3, 3), False, False),
                                                 ("push", (-1, -1), False, True), ("pull", (0, 0), False, True),
                                                 ("push_pull", (-3.75, -3.75), True, True)])
def test_rollout_bit_exact_with_contacts(oracle, task, goal, mm, avoid):
    """Dense-contact stress: robot starts between box, dyn-obs, obstacle and a wall corner;
    large random controls.  Every state / action / cost must equal the oracle's bit-for-bit.
    avoid: the extension m3_set_avoid_dyn_obs (push / pull with get_motion_cost, as the reference's logged
    `case2_*_coll` experiments evidently ran): states identical to the plain task, costs + 1000 per step in contact."""
    from m3p2i_aip_amd import _lib as L
    K, T = 1024, 30
    rng = np.random.default_rng(11)
    delta = (rng.standard_normal((K, T, 2)) * 1.5).astype(np.float32)
    worlds = []
    w = oracle.init_world(1)[0]
    worlds.append(w.copy())
    w2 = w.copy()
    w2[0:2] = (-1.2, 2.0)            # between box (0,2) and dyn-obs (-2,2)
    w2[oracle.W_B:oracle.W_B + 2] = (-0.7, 2.05)
    worlds.append(w2)
    w3 = w.copy()
    w3[0:2] = (3.2, 3.2)             # wall corner, obstacle at (2,2) close by
    w3[oracle.W_B:oracle.W_B + 4] = (3.4, 2.6, np.cos(0.4), np.sin(0.4))
    w3[oracle.W_D:oracle.W_D + 4] = (2.6, 3.5, np.cos(-0.9), np.sin(-0.9))
    worlds.append(w3)
    ocfg = oracle.make_cfg(K, T, 2, task=task, goal=goal, multi_modal=mm)
    ocfg.avoid_dyn_obs = int(avoid)
    eng = _engine(K=K, T=T, nu=2, multi_modal=mm, u_min=[-3, -3], u_max=[3, 3],
                  noise_sigma_diag=[3, 3])
    eng.set_objective(task, goal)
    eng.set_avoid_dyn_obs(avoid)
    eng.set_noise(delta)
    penalised = 0
    for w0 in worlds:
        opl = oracle.OraclePointPlanner(ocfg, delta)
        eng.reset()
        eng.set_world_point_raw(raw_world(w0))
        eng.command(sync_host=True)
        opl.command(w0)
        st = eng.states.cpu().numpy()
        np.testing.assert_array_equal(eng.actions.cpu().numpy(), opl.last["actions"])
        bad = np.argwhere(st != opl.last["states"])
        assert bad.size == 0, f"first mismatch at (k,t,c)={bad[0]} of {len(bad)}"
        np.testing.assert_array_equal(eng.cost_horizon.cpu().numpy(), opl.last["cost_h"])
        np.testing.assert_array_equal(eng.buffer(L.BUF_TRAJ_COST).cpu().numpy(), opl.last["J"])
        np.testing.assert_allclose(eng.buffer(L.BUF_WEIGHTS).cpu().numpy(), opl.last["w"],
                                   rtol=2e-3, atol=1e-6)
        np.testing.assert_allclose(eng.buffer(L.BUF_MEAN).cpu().numpy(), opl.mean, atol=1e-4)
        np.testing.assert_allclose(eng.buffer(L.BUF_PENDING_FORCE).cpu().numpy().T,
                                   opl.pend, atol=0)
        penalised += int((opl.last["cost_h"] >= 1000.0).sum())
    assert (penalised > 100) == (avoid or task == "navigation")     # (the penalty really fired / is really absent)
    eng.close()


def test_api_errors_are_loud(oracle):
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd.engine import HipEngine, make_config
    with pytest.raises(L.M3Error):
        HipEngine(make_config(K=10, T=30))            # K < 20: topk(20)
    with pytest.raises(L.M3Error):
        HipEngine(make_config(K=64, T=8))             # filter window 9 > T
    eng = HipEngine(make_config(K=64, T=12))
    with pytest.raises(L.M3Error):
        eng.command()                                  # no noise set
    with pytest.raises(L.M3Error):
        eng.set_objective("push_pull", (0, 0))         # needs multi_modal
    with pytest.raises(L.M3Error):
        eng.set_objective("reach", (0, 0))             # not a point_env task
    eng.close()
    # noise_sigma_full must be a covariance matrix with the configured diagonal
    with pytest.raises(L.M3Error):
        HipEngine(make_config(K=64, T=12, sampling_random=True, noise_sigma=[[3.0, 4.0], [4.0, 3.0]]))   # not positive definite
    c = make_config(K=64, T=12, sampling_random=True, noise_sigma=[[3.0, 1.0], [1.0, 2.0]])
    c.noise_sigma_full[1] = 0.5                                                                         # not symmetric
    with pytest.raises(L.M3Error):
        HipEngine(c)
    c = make_config(K=64, T=12, sampling_random=True, noise_sigma=[[3.0, 1.0], [1.0, 2.0]])
    c.noise_sigma_diag[0] = 2.0                                                                         # diagonals disagree
    with pytest.raises(L.M3Error):
        HipEngine(c)
    with pytest.raises(L.M3Error):   # update_cov on a sharded single-mode handle would need one more reduction
        HipEngine(make_config(K=128, T=12, K_local=64, k_offset=0, update_cov=True))
    # ... while the modes in which the reference ignores the flag take it silently (m3p2i.py:66-92 has no such branch)
    HipEngine(make_config(K=128, T=12, K_local=64, k_offset=0, update_cov=True, multi_modal=True)).close()


@pytest.mark.parametrize("task,goal,mm", [("push", (-1, -1), False), ("push_pull", (-3.75, -3.75), True)])
def test_sharded_handles_equal_unsharded(golden, oracle, task, goal, mm):
    """Two shard handles (rank 0 / rank 1 of world_size 2) on ONE GPU with the two collectives
    done by hand (concatenate J, add the REDUCE buffers) must reproduce the unsharded handle:
    checks the kernels' global-index logic (k_offset, specials at k=0, K/2, K-1, owner-only
    rows) that the RCCL path relies on."""
    from m3p2i_aip_amd import _lib as L
    K, T = 256, 30
    delta = golden["g9_push_delta"]
    kw = dict(T=T, nu=2, multi_modal=mm, u_min=[-3, -3], u_max=[3, 3], noise_sigma_diag=[3, 3])
    full = _engine(K=K, **kw)
    shards = [_engine(K=K, K_local=K // 2, k_offset=r * (K // 2), **kw) for r in range(2)]
    w0 = oracle.init_world(1)[0]
    w0[0:2] = (0.0, 1.5)
    for e, d in [(full, delta)] + [(shards[r], delta[r * 128:(r + 1) * 128]) for r in range(2)]:
        e.set_objective(task, goal)
        e.set_noise(d)
        e.set_world_point_raw(raw_world(w0))
    for call in range(4):
        full.command()
        for e in shards:
            e.rollout()
        J = torch.cat([e.buffer(L.BUF_TRAJ_COST) for e in shards])           # all_gather
        for e in shards:
            e.buffer(L.BUF_TRAJ_COST_ALL).copy_(J)
            e.update()
        red = shards[0].buffer(L.BUF_REDUCE) + shards[1].buffer(L.BUF_REDUCE)  # all_reduce(sum)
        for e in shards:
            e.buffer(L.BUF_REDUCE).copy_(red)
            e.finalize()
        torch.cuda.synchronize()
        for e in shards:
            for b in (L.BUF_ACTION_OUT, L.BUF_MEAN, L.BUF_MEAN_1, L.BUF_MEAN_2, L.BUF_BEST,
                      L.BUF_BEST_1, L.BUF_BEST_2, L.BUF_TOP_TRAJS):
                np.testing.assert_allclose(e.buffer(b).cpu().numpy(), full.buffer(b).cpu().numpy(),
                                           atol=2e-5, err_msg=f"call {call} buffer {b}")
            np.testing.assert_allclose(e.buffer(L.BUF_WEIGHTS).cpu().numpy(),
                                       full.buffer(L.BUF_WEIGHTS).cpu().numpy(), rtol=1e-3, atol=1e-8)
            np.testing.assert_array_equal(e.buffer(L.BUF_TOP_IDX).cpu().numpy(),
                                          full.buffer(L.BUF_TOP_IDX).cpu().numpy())
        st = torch.cat([e.states for e in shards]).cpu().numpy()
        from tests.conftest import assert_close_but_few
        if call == 0:
            np.testing.assert_array_equal(st, full.states.cpu().numpy())      # same inputs: same bits
        else:   # (the means of the sharded and the unsharded update agree to 2e-5: a rollout in contact may amplify that)
            assert_close_but_few(st, full.states.cpu().numpy(), atol=1e-4, frac=1e-3, cap=0.05, err_msg=f"call {call} states")
        assert shards[0].info().pull_preference == full.info().pull_preference
    for e in shards + [full]:
        e.close()


@pytest.mark.parametrize("world,K,mode", [(2, 256, "halton"), (4, 512, "halton"), (8, 16000, "halton"),
                                          (2, 256, "simple")])
def test_one_collective_shard_mix_equals_unsharded(golden, oracle, world, K, mode):
    """cfg.shard_mix: `world` shard handles on ONE GPU, the single collective done by hand
    (stack the ranks' RECORD buffers into RECORDS_ALL).  The mixture of per-rank softmins must
    reproduce the unsharded command() (same plan, means, best trajectory, top-k, eta, weights)
    up to f32 rounding, and every rank must produce the same plan bit for bit."""
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd import sampling
    T = 30
    simple = mode == "simple"
    delta = golden["g9_push_delta"] if K == 256 else sampling.halton_spline_delta(K, T, 2).numpy()
    kw = dict(T=T, nu=2, u_min=[-3, -3], u_max=[3, 3], noise_sigma_diag=[3, 3], mode_simple=simple,
              lambda_=0.5 if simple else 1.0)
    Kl = K // world
    full = _engine(K=K, **kw)
    shards = [_engine(K=K, K_local=Kl, k_offset=r * Kl, shard_mix=True, **kw) for r in range(world)]
    w0 = oracle.init_world(1)[0]
    w0[0:2] = (0.0, 1.5)
    for e, d in [(full, delta)] + [(shards[r], delta[r * Kl:(r + 1) * Kl]) for r in range(world)]:
        e.set_objective("push", (-1, -1))
        e.set_noise(d)
        e.set_world_point_raw(raw_world(w0))
    for call in range(4):
        full.command()
        for e in shards:
            e.rollout()
            e.update()
        allrec = torch.stack([e.buffer(L.BUF_RECORD) for e in shards])        # all_gather
        for e in shards:
            e.buffer(L.BUF_RECORDS_ALL).copy_(allrec)
            e.finalize()
        torch.cuda.synchronize()
        fi = full.info()
        for r, e in enumerate(shards):
            for b in (L.BUF_ACTION_OUT, L.BUF_MEAN, L.BUF_BEST, L.BUF_TOP_TRAJS):
                np.testing.assert_allclose(e.buffer(b).cpu().numpy(), full.buffer(b).cpu().numpy(),
                                           atol=3e-5, err_msg=f"call {call} rank {r} buffer {b}")
                assert torch.equal(e.buffer(b), shards[0].buffer(b))
            np.testing.assert_array_equal(e.buffer(L.BUF_TOP_IDX).cpu().numpy(),
                                          full.buffer(L.BUF_TOP_IDX).cpu().numpy())
            np.testing.assert_allclose(e.buffer(L.BUF_WEIGHTS)[r * Kl:(r + 1) * Kl].cpu().numpy(),
                                       full.buffer(L.BUF_WEIGHTS)[r * Kl:(r + 1) * Kl].cpu().numpy(),
                                       rtol=1e-3, atol=1e-8)
            # ... plus the global top-k samples' weights wherever they live: planner.top_values
            ti = e.buffer(L.BUF_TOP_IDX).long()
            np.testing.assert_allclose(e.buffer(L.BUF_WEIGHTS)[ti].cpu().numpy(), full.buffer(L.BUF_WEIGHTS)[ti].cpu().numpy(),
                                       rtol=1e-3, atol=1e-8)
            i = e.info()
            assert i.best_idx == fi.best_idx
            assert abs(i.eta - fi.eta) <= 1e-4 * fi.eta
            assert abs(i.wsum_push - fi.wsum_push) < 1e-4 and abs(i.wsum_pull - fi.wsum_pull) < 1e-4
    for e in shards + [full]:
        e.close()


def test_shard_mix_multi_modal_is_refused_without_a_noise_table():
    """(the one-collective multi-modal protocol itself: tests/test_c5_sharded_gpu.py)"""
    from m3p2i_aip_amd import _lib as L
    with pytest.raises(L.M3Error):
        _engine(K=256, K_local=128, k_offset=0, shard_mix=True, T=30, nu=2, multi_modal=True, sampling_random=True,
                u_min=[-3, -3], u_max=[3, 3], noise_sigma_diag=[3, 3])


@pytest.mark.parametrize("seed", range(48))
def test_rollout_bit_exact_on_random_worlds(oracle, seed):
    """Fuzz: robot, box and dyn-obs anywhere in the arena -- overlapping each other, the obstacle or a
    wall, rotated, moving and spinning -- with a random task; strong random controls.  Exercises the
    heavy substep instances (all 19 slots, walls, box-box) that planned rollouts rarely reach.
    States / actions / costs must equal the oracle's bit-for-bit, with the coherent wavefront order
    (default) as well as by index."""
    from m3p2i_aip_amd import _lib as L
    rng = np.random.default_rng(1000 + seed)
    K, T = 256, 30
    task, goal, mm = [("push", (-1, -1), False), ("pull", (0, 0), False), ("push_pull", (-3.75, -3.75), True),
                      ("navigation", (-3, 3), False)][seed % 4]
    delta = (rng.standard_normal((K, T, 2)) * 1.5).astype(np.float32)
    w = oracle.init_world(1)[0]

    def place(lo=-3.8, hi=3.8):
        spot = rng.integers(0, 4)
        if spot == 0:   # anywhere
            return rng.uniform(lo, hi, 2)
        if spot == 1:   # against a wall / in a corner
            p = rng.uniform(lo, hi, 2)
            p[rng.integers(0, 2)] = rng.choice([-1, 1]) * rng.uniform(3.4, 3.85)
            if rng.random() < 0.5:
                p[:] = rng.choice([-1, 1], 2) * rng.uniform(3.3, 3.8, 2)
            return p
        if spot == 2:   # at the obstacle (2, 2)
            return np.array([2.0, 2.0]) + rng.uniform(-0.6, 0.6, 2)
        return np.array([0.0, 1.0]) + rng.uniform(-0.8, 0.8, 2)   # in the middle, near each other

    w[0:2] = place()
    w[4:6] = rng.normal(0, 1.0, 2)                      # robot velocity
    for base in (oracle.W_B, oracle.W_D):
        yaw = rng.uniform(-np.pi, np.pi)
        w[base:base + 2] = place()
        w[base + 2:base + 4] = (np.cos(yaw), np.sin(yaw))
        if rng.random() < 0.5:
            w[base + 4:base + 6] = rng.normal(0, 0.5, 2)
            w[base + 6] = rng.normal(0, 1.0)
    w = w.astype(np.float32)
    ocfg = oracle.make_cfg(K, T, 2, task=task, goal=goal, multi_modal=mm)
    opl = oracle.OraclePointPlanner(ocfg, delta)
    opl.command(w)
    for order in (True, False):
        eng = _engine(K=K, T=T, nu=2, multi_modal=mm, u_min=[-3, -3], u_max=[3, 3], noise_sigma_diag=[3, 3])
        eng.set_wave_order(order)
        eng.set_objective(task, goal)
        eng.set_noise(delta)
        eng.set_world_point_raw(raw_world(w))
        eng.command(sync_host=True)
        st = eng.states.cpu().numpy()
        assert np.isfinite(st).all()
        np.testing.assert_array_equal(eng.actions.cpu().numpy(), opl.last["actions"])
        bad = np.argwhere(st != opl.last["states"])
        assert bad.size == 0, f"seed {seed}: first mismatch at (k,t,c)={bad[0]} of {len(bad)}"
        np.testing.assert_array_equal(eng.cost_horizon.cpu().numpy(), opl.last["cost_h"])
        np.testing.assert_array_equal(eng.buffer(L.BUF_TRAJ_COST).cpu().numpy(), opl.last["J"])
        eng.close()
