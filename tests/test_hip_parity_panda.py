"""GPU parity tests for the panda_env hot path: fused HIP rollout + update (through the C-ABI)
vs the CPU oracle (Panda chain spec v1 + the reference's reach/pick/place costs, the latter
pinned by golden group G6b).  Run with ``pytest -m gpu``."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

UMIN = [-2.0] * 7 + [-1.5] * 2
UMAX = [2.0] * 7 + [1.5] * 2
SIG = [10.0] * 7 + [0.8] * 2


def raw31(P, world):
    """what set_world_panda_raw takes (now 57 floats: q9 qd9 cubeA13 cubeB13 dyn-obs13 -- the wrapper's tensors' content)"""
    return P.raw57(world)


from tests.panda_worlds import grasp_world  # noqa: E402


@pytest.mark.parametrize("task,mm,grip,held", [("reach", False, 1, False), ("reach", True, 1, False),
                                               ("pick", False, 2, True), ("pick", False, 2, False),
                                               ("place", False, 1, True),
                                               # open gripper around the cube / 3, 8, 15 cm above it:
                                               # rollouts grasp during the horizon, or pass the bound of
                                               # the lazy kinematics (panda_step LAZY_FK) closely
                                               ("pick", False, 2, "open0"), ("pick", False, 2, "open3"),
                                               ("pick", False, 2, "open8"), ("reach", False, 2, "open15"),
                                               # spec v1.1 (pad channel): open gripper at the grasp height, displaced
                                               # along the pads' closing direction (swept + held) / their width
                                               ("pick", False, 2, "offy12"), ("pick", False, 2, "offy30"),
                                               ("pick", False, 2, "offx20"), ("pick", False, 2, "offx30")])
def test_panda_command_matches_oracle(oracle, task, mm, grip, held):
    import oracle.panda as P
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd.engine import HipEngine, make_config
    sc = P.default_scene()
    K, T = 256, 20
    rng = np.random.default_rng(5)
    delta = rng.standard_normal((K, T, 9)).astype(np.float32)
    goal = np.array([0.2, 0.2, 1.115, 0, 0, 0, 1], np.float32)
    if isinstance(held, str) and held.startswith("off"):
        d = 0.001 * int(held[4:])
        w0 = grasp_world(P, sc, close_gripper=False, offset=(d, 0.0) if held[3] == "x" else (0.0, d))
        held = False
    elif isinstance(held, str):
        w0 = grasp_world(P, sc, close_gripper=False, lift=0.01 * int(held[4:]))
        held = False
    else:
        w0 = grasp_world(P, sc) if held else P.init_world(1)[0]
    cfg = P.make_cfg(K, T, multi_modal=mm, task=task, goal=goal, gripper_cmd=grip)
    opl = P.OraclePandaPlanner(cfg, delta, sc)
    eng = HipEngine(make_config(K=K, T=T, nu=9, env_type="panda_env", multi_modal=mm, u_min=UMIN, u_max=UMAX,
                                noise_sigma_diag=SIG, lambda_=0.05, pre_height_diff=0.05, dt=0.01))
    eng.set_objective(task, goal, gripper_cmd=grip)
    eng.set_noise(delta)
    eng.set_world_panda_raw(raw31(P, w0))
    for call in range(3):
        a_hip = eng.command(sync_host=True)
        a_orc = opl.command(w0)
        st, ac = eng.states.cpu().numpy(), eng.actions.cpu().numpy()
        ch, J = eng.cost_horizon.cpu().numpy(), eng.buffer(L.BUF_TRAJ_COST).cpu().numpy()
        if call == 0:  # identical inputs: the rollout must agree bit-for-bit
            np.testing.assert_array_equal(ac, opl.last["actions"])
            np.testing.assert_array_equal(st, opl.last["states"])
            bad = np.argwhere(ch != opl.last["cost_h"])
            assert bad.size == 0, f"{len(bad)} cost mismatches, first {bad[0]}: {ch[tuple(bad[0])]} vs {opl.last['cost_h'][tuple(bad[0])]}"
            np.testing.assert_array_equal(J, opl.last["J"])
        # later calls: the two updates' means agree to ~1e-6, not to the bit, and with spec v2 a rollout in contact can
        # turn such a difference into another contact history (unilateral contacts are not continuous): all but a few
        # rollouts agree, the plan agrees
        same = np.isclose(ch, opl.last["cost_h"], rtol=1e-4, atol=1e-3).all(axis=1)
        assert same.mean() > 0.90, f"call {call}: {int((~same).sum())} of {K} rollouts differ"
        np.testing.assert_allclose(a_hip, a_orc, atol=1e-3, err_msg=f"call {call}")
        wd = np.isclose(eng.buffer(L.BUF_WEIGHTS).cpu().numpy(), opl.last["w"], rtol=2e-3, atol=1e-6)
        assert wd.mean() > 0.90 and np.abs(eng.buffer(L.BUF_WEIGHTS).cpu().numpy() - opl.last["w"]).max() < 2e-3
        info = eng.info()
        assert info.beta == pytest.approx(opl.beta, rel=1e-5)   # panda adapts beta (mppi.py:446-454)
    if held and task == "pick":
        assert np.ptp(ch) > 0.05       # the held cube really moves with the hand in the rollouts
    eng.close()


import functools  # noqa: E402


@functools.lru_cache(maxsize=4)
def _fuzz_batch(b):
    import oracle.panda as P
    from tests.test_device_dynamics_on_host import random_panda_worlds
    return random_panda_worlds(P, P.default_scene(), 42, np.random.default_rng(900 + b))


@pytest.mark.parametrize("lps", [1, 8, 16])
@pytest.mark.parametrize("seed", range(36))
def test_panda_rollout_bit_exact_on_random_worlds(oracle, seed, lps):
    """Fuzz: a random world per seed -- random joint configuration and velocities; cubeA on the table / near the hand /
    falling onto the shelf / stacked on cubeB (resting, dropped, beyond the edge) / tumbling in the air next to cubeB;
    cubeB and the plate near the hand, moving; the gripper pointing down with its finger tips at the table or at cubeB;
    the cube held / the open gripper around it (tests/test_device_dynamics_on_host.random_panda_worlds) --, random
    gripper command and task, strong random controls.  The rollout must equal the oracle's bit for bit: the contact
    detection (wave-level rejects, lazy kinematics), the gripper rows, the cubes' manifolds in LDS, sleeping and waking
    all decide per wave or per lane what to evaluate -- random worlds put everything at every distance.  In all three forms
    of the kernel (world spec v3): a lane per sample, and eight / sixteen lanes sharing a sample with the contact rows across
    them (DPP butterflies, row shifts between the joints' and the cubes' vectors, rows in registers / lane-distributed LDS)."""
    import oracle.panda as P
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd.engine import HipEngine, make_config
    rng = np.random.default_rng(500 + seed)
    sc = P.default_scene()
    K, T = 128, 20
    task, grip = [("reach", 1), ("pick", 2), ("place", 1), ("reach", 2)][seed % 4]
    # (the generator cycles through its world kinds by index: take world `seed` of a batch)
    w0 = _fuzz_batch(seed // 42)[seed % 42].copy().astype(np.float32)
    goal = np.array([0.2, 0.2, 1.115, 0, 0, 0, 1], np.float32)
    delta = rng.standard_normal((K, T, 9)).astype(np.float32)
    cfg = P.make_cfg(K, T, multi_modal=False, task=task, goal=goal, gripper_cmd=grip)
    opl = P.OraclePandaPlanner(cfg, delta, sc)
    eng = HipEngine(make_config(K=K, T=T, nu=9, env_type="panda_env", u_min=UMIN, u_max=UMAX,
                                noise_sigma_diag=SIG, lambda_=0.05, pre_height_diff=0.05, dt=0.01))
    eng.set_objective(task, goal, gripper_cmd=grip)
    eng.set_panda_lanes_per_sample(lps)
    eng.set_panda_reach_cost_kernel((seed // 4) % 2 == 0)   # reach: the cost kernel behind a rollout without shadow slots / the shadows
    eng.set_noise(delta)
    eng.set_world_panda_raw(raw31(P, w0))
    eng.command(sync_host=True)
    opl.command(w0)
    st = eng.states.cpu().numpy()
    assert np.isfinite(st).all()
    np.testing.assert_array_equal(eng.actions.cpu().numpy(), opl.last["actions"])
    np.testing.assert_array_equal(st, opl.last["states"])
    ch = eng.cost_horizon.cpu().numpy()
    bad = np.argwhere(ch != opl.last["cost_h"])
    assert bad.size == 0, f"seed {seed}: {len(bad)} cost mismatches, first {bad[0]}"
    np.testing.assert_array_equal(eng.buffer(L.BUF_TRAJ_COST).cpu().numpy(), opl.last["J"])
    eng.close()


@pytest.mark.parametrize("tag,task,mm,grip", [("panda_reach", "reach", False, 1), ("panda_reachmm", "reach", True, 1),
                                              ("panda_pick", "pick", False, 2)])
def test_panda_command_traces_vs_reference_golden(golden, tag, task, mm, grip):
    """G9 (panda): the HIP command() against five consecutive calls of the REFERENCE's own M3P2I + Objective
    (tests/golden/make_golden.py g9_panda: the reference's planner + panda costs driven through their plugin
    API; C4-shaped at K = 256, T = 20).  Bar: 1e-3 on the control output and the weights; the adapted,
    persistent beta of the panda env (mppi.py:446-454) must follow the reference's call by call."""
    import oracle.panda as P
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd.engine import HipEngine, make_config
    K, T = 256, 20
    goal = np.array([0.2, 0.2, 1.115, 0, 0, 0, 1], np.float32)
    eng = HipEngine(make_config(K=K, T=T, nu=9, env_type="panda_env", multi_modal=mm, u_min=UMIN, u_max=UMAX,
                                noise_sigma_diag=SIG, lambda_=0.05, pre_height_diff=0.05, dt=0.01))
    eng.set_objective(task, goal, gripper_cmd=grip)
    eng.set_noise(golden[f"g9_{tag}_delta"])
    for call, w in enumerate(golden[f"g9_{tag}_world"]):
        eng.set_world_panda_raw(raw31(P, w))
        a = eng.command(sync_host=True)
        np.testing.assert_allclose(a, golden[f"g9_{tag}_action"][call], atol=1e-3, err_msg=f"{tag} call {call}")
        np.testing.assert_allclose(eng.buffer(L.BUF_WEIGHTS).cpu().numpy(), golden[f"g9_{tag}_weights"][call], rtol=2e-3, atol=1e-6)
        np.testing.assert_allclose(eng.buffer(L.BUF_MEAN).cpu().numpy(), golden[f"g9_{tag}_mean"][call], atol=1e-3)
        np.testing.assert_allclose(eng.buffer(L.BUF_TOP_TRAJS).cpu().numpy()[:5], golden[f"g9_{tag}_top_trajs"][call][:5],
                                   atol=1e-3)
        info = eng.info()
        if not mm:
            assert info.beta == pytest.approx(float(golden[f"g9_{tag}_beta"][call]), rel=1e-5)
        else:
            assert info.pull_preference == int(golden[f"g9_{tag}_pref"][call])
    # (all but a few rollouts: a rollout in contact amplifies the 1e-6 difference of the controls -- test_oracle_panda.py)
    bad = np.abs(eng.states.cpu().numpy() - golden[f"g9_{tag}_states_last"]).max(axis=(1, 2)) > 1e-3
    assert bad.mean() < 0.08, f"{int(bad.sum())} of {bad.size} rollouts differ"
    np.testing.assert_allclose(eng.actions.cpu().numpy(), golden[f"g9_{tag}_actions_last"], atol=1e-3)
    eng.close()


@pytest.mark.parametrize("tag,mm", [("panda_reach_touch", False), ("panda_reachmm_touch", True)])
def test_reach_cost_reads_environment_0_cube_as_the_reference(golden, oracle, tag, mm):
    """Quirk Q8 with world spec v2 (cost_functions.py:97 `cube_state[0, :3]`, skill_utils.py:274): the open gripper stands
    around cubeA and the rollouts -- sample 0's too -- push it; the reference measures EVERY rollout's reach cost against
    environment 0's cube (tilted mode: + the orientation of the first environment of the second half).  The rollout kernel
    carries those samples as shadow lanes of every wavefront: bit-identical to the oracle (which simulates them first), and
    within 1e-3 of the reference's own planner + cost code on the first call (later calls: test_oracle_panda.py)."""
    import oracle.panda as P
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd.engine import HipEngine, make_config
    K, T = 256, 20
    goal = np.array([0.2, 0.2, 1.115, 0, 0, 0, 1], np.float32)
    cfg = P.make_cfg(K, T, multi_modal=mm, task="reach", goal=goal, gripper_cmd=1)
    opl = P.OraclePandaPlanner(cfg, golden[f"g9_{tag}_delta"])
    w = golden[f"g9_{tag}_world"][0]
    opl.command(w)
    # the quirk is live: with each sample's own cube the costs are other numbers
    act = opl.last["actions"] / cfg.u_scale
    own = np.concatenate([P.rollout(cfg, opl.sc, w, act[:K // 2], 0, K // 2)["J"], P.rollout(cfg, opl.sc, w, act[K // 2:], K // 2, K)["J"]])
    assert np.abs(own - opl.last["J"]).max() > 1.0
    for lanes in (0, 16):      # (the narrow-wave knob keeps the shadows: lanes per wavefront are clipped to 64 - shadows)
        eng = HipEngine(make_config(K=K, T=T, nu=9, env_type="panda_env", multi_modal=mm, u_min=UMIN, u_max=UMAX,
                                    noise_sigma_diag=SIG, lambda_=0.05, pre_height_diff=0.05, dt=0.01))
        eng.set_rollout_lanes(lanes)
        eng.set_objective("reach", goal, gripper_cmd=1)
        eng.set_noise(golden[f"g9_{tag}_delta"])
        eng.set_world_panda_raw(raw31(P, w))
        a = eng.command(sync_host=True)
        np.testing.assert_array_equal(eng.states.cpu().numpy(), opl.last["states"])
        np.testing.assert_array_equal(eng.cost_horizon.cpu().numpy(), opl.last["cost_h"])
        np.testing.assert_array_equal(eng.buffer(L.BUF_TRAJ_COST).cpu().numpy(), opl.last["J"])
        np.testing.assert_allclose(a, golden[f"g9_{tag}_action"][0], atol=1e-3)
        np.testing.assert_allclose(eng.buffer(L.BUF_WEIGHTS).cpu().numpy(), golden[f"g9_{tag}_weights"][0], rtol=2e-3, atol=1e-6)
        np.testing.assert_allclose(eng.buffer(L.BUF_MEAN).cpu().numpy(), golden[f"g9_{tag}_mean"][0], atol=1e-3)
        eng.close()


@pytest.mark.parametrize("tag", ["panda_opt_cov", "panda_opt_rand", "panda_opt_simple"])
def test_panda_option_traces_vs_reference_golden(golden, oracle, tag):
    """The MPPIConfig switches no shipped config turns on, on the panda_env (make_golden.py g11): update_cov;
    sampling_method='random' with a noise mean and a non-diagonal noise_sigma; mppi_mode='simple' with
    noise_abs_cost and u_scale != 1 -- HIP command() vs the reference's own planner and vs the oracle."""
    import oracle.panda as P
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd.engine import HipEngine, make_config
    from tests.test_oracle_panda import PANDA_OPT, panda_opt_planner
    kw = dict(PANDA_OPT[tag])
    ocfg, opl = panda_opt_planner(golden, tag)
    goal = np.array([0.2, 0.2, 1.115, 0, 0, 0, 1], np.float32)
    delta = golden[f"g9_{tag}_delta"] if f"g9_{tag}_delta" in golden else None
    simple = bool(kw.get("mode_simple"))
    eng = HipEngine(make_config(K=kw["K"], T=kw["T"], nu=9, env_type="panda_env", u_min=UMIN, u_max=UMAX,
                                noise_sigma_diag=SIG, lambda_=0.05, pre_height_diff=0.05, dt=0.01, seed=7,
                                mode_simple=simple, sampling_random=delta is None, u_per_command=kw.get("u_per_command"),
                                u_scale=kw.get("u_scale", 1.0), noise_mu=kw.get("noise_mu"), noise_sigma=kw.get("noise_sigma"),
                                noise_abs_cost=kw.get("noise_abs_cost", False), update_cov=kw.get("update_cov", False)))
    eng.set_objective("reach", goal, gripper_cmd=1)
    if delta is not None:
        eng.set_noise(delta)
    for call, w in enumerate(golden[f"g9_{tag}_world"]):
        if simple and call:
            # lambda_ = 0.05 makes the softmin of simple mode (mppi.py:226: exp(-(J - min) / lambda_)) amplify one
            # f32 ulp of a trajectory cost of ~60 into 1e-4 of a weight; left alone, two correct implementations
            # drift apart by ~1e-3 per warm-started call (the reference itself carries cost_total + mean(S), quirk
            # Q1).  Every call therefore starts from the REFERENCE's previous U: four independent comparisons.
            eng.set_plan(L.BUF_MEAN, golden[f"g9_{tag}_mean"][call - 1])
            opl.U = golden[f"g9_{tag}_mean"][call - 1].copy()
        eng.set_world_panda_raw(raw31(P, w))
        a = eng.command(sync_host=True)
        a_orc = opl.command(w)
        rows = golden[f"g9_{tag}_action"][call].shape[0]
        np.testing.assert_allclose(a[:rows], golden[f"g9_{tag}_action"][call], atol=1e-3, err_msg=f"{tag} call {call}")
        np.testing.assert_allclose(a[:rows], a_orc, atol=1e-3)
        np.testing.assert_allclose(eng.buffer(L.BUF_WEIGHTS).cpu().numpy(), golden[f"g9_{tag}_weights"][call], rtol=2e-3, atol=1e-6)
        np.testing.assert_allclose(eng.buffer(L.BUF_MEAN).cpu().numpy(), golden[f"g9_{tag}_mean"][call], atol=1e-3)
        if not simple:
            assert eng.info().beta == pytest.approx(float(golden[f"g9_{tag}_beta"][call]), rel=1e-5)
        if f"g9_{tag}_extra" in golden:
            np.testing.assert_allclose(eng.buffer(L.BUF_COV).cpu().numpy()[1], golden[f"g9_{tag}_extra"][call], rtol=1e-4)
        if call == 0 and delta is not None:     # same inputs: the rollout is bit-identical to the oracle's
            np.testing.assert_array_equal(eng.states.cpu().numpy(), opl.last["states"])
            np.testing.assert_array_equal(eng.actions.cpu().numpy(), opl.last["actions"])
    np.testing.assert_allclose(eng.states.cpu().numpy(), golden[f"g9_{tag}_states_last"], atol=1e-3)
    np.testing.assert_allclose(eng.actions.cpu().numpy() / np.float32(ocfg.u_scale), golden[f"g9_{tag}_actions_last"], atol=1e-3)
    eng.close()


@pytest.mark.parametrize("lps", [1, 8, 16, -8, -16])
@pytest.mark.parametrize("K,task,mm", [(203, "pick", False), (61, "reach", False), (90, "reach", True), (22, "place", False)])
def test_panda_ragged_sample_counts_in_every_kernel_form(oracle, K, task, mm, lps):
    """Sample counts that fill no wavefront evenly -- a last wavefront with one, two or three of its four (eight) sample slots
    used, fewer samples than one wavefront of the one-lane form, the shadow slots of quirk Q8 (reach: one, multi-modal: two) taking
    their share -- in the three forms of the kernel: every state, action, cost and trajectory cost equals the oracle's bit for
    bit, nothing is written past the K rows.  (lps < 0: that many lanes with the reach cost's shadow slots instead of the cost
    kernel behind a rollout without them -- the default for lps = 8 / 16.)"""
    import oracle.panda as P
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd.engine import HipEngine, make_config
    if lps < 0 and task != "reach":
        pytest.skip("the switch concerns the reach cost only")
    shadows, lps = lps < 0, abs(lps)
    sc = P.default_scene()
    T = 20
    rng = np.random.default_rng(900 + K)
    delta = rng.standard_normal((K, T, 9)).astype(np.float32)
    goal = np.array([0.2, 0.2, 1.115, 0, 0, 0, 1], np.float32)
    grip = 2 if task == "pick" else 1
    w0 = grasp_world(P, sc) if task in ("pick", "place") else grasp_world(P, sc, close_gripper=False, lift=0.0)
    cfg = P.make_cfg(K, T, multi_modal=mm, task=task, goal=goal, gripper_cmd=grip)
    opl = P.OraclePandaPlanner(cfg, delta, sc)
    eng = HipEngine(make_config(K=K, T=T, nu=9, env_type="panda_env", multi_modal=mm, u_min=UMIN, u_max=UMAX,
                                noise_sigma_diag=SIG, lambda_=0.05, pre_height_diff=0.05, dt=0.01))
    eng.set_objective(task, goal, gripper_cmd=grip)
    eng.set_panda_lanes_per_sample(lps)
    eng.set_panda_reach_cost_kernel(not shadows)
    eng.set_noise(delta)
    eng.set_world_panda_raw(raw31(P, w0))
    eng.command(sync_host=True)
    opl.command(w0)
    assert eng.panda_lanes_per_sample_used() == lps
    np.testing.assert_array_equal(eng.actions.cpu().numpy(), opl.last["actions"])
    np.testing.assert_array_equal(eng.states.cpu().numpy(), opl.last["states"])
    np.testing.assert_array_equal(eng.cost_horizon.cpu().numpy(), opl.last["cost_h"])
    np.testing.assert_array_equal(eng.buffer(L.BUF_TRAJ_COST).cpu().numpy(), opl.last["J"])
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cost_kernel", [False, True])
def test_reach_kernel_form_follows_what_the_rollouts_meet(oracle, cost_kernel):
    """The automatic form of the REACH command (m3_set_panda_lanes_per_sample 0).  With the cost kernel available and one round of
    sixteen-lane wavefronts fitting (K <= 4096; round 6: that form is at least as fast in every scene of an episode): sixteen lanes
    per sample without shadow slots + the cost kernel, always.  Where it is not (m3_set_panda_reach_cost_kernel 0, as beyond
    K = 8192): one lane per sample (with quirk Q8's shadow slots) while the gripper is within reach of a box in few of the rollouts'
    (sample, substep) pairs, eight lanes with them from the command after the kernel reported many (m3_panda_near_share).  The
    plans do not depend on it: the same commands with the form forced to one lane give the same bits."""
    many = 16 if cost_kernel else 8
    quiet = 16 if cost_kernel else 1          # the form in a scene with next to nothing near anything
    import oracle.panda as P
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd.engine import HipEngine, make_config
    sc = P.default_scene()
    K, T = 512, 20
    delta = np.random.default_rng(9).standard_normal((K, T, 9)).astype(np.float32)
    goal = np.array([0.2, 0.2, 1.115, 0, 0, 0, 1], np.float32)

    def engine(lps):
        eng = HipEngine(make_config(K=K, T=T, nu=9, env_type="panda_env", u_min=UMIN, u_max=UMAX, noise_sigma_diag=SIG,
                                    lambda_=0.05, pre_height_diff=0.05, dt=0.01))
        eng.set_objective("reach", goal, gripper_cmd=1)
        eng.set_noise(delta)
        eng.set_panda_lanes_per_sample(lps)
        eng.set_panda_reach_cost_kernel(cost_kernel)
        return eng
    # arm up, the cubes at rest on the table (the initial scene after its cubes have landed): few pairs near anything, no cube
    # awake -> the quiet form, command after command
    rest = P.init_world(1)
    for _ in range(30):
        P.step_batch(sc, rest, np.zeros((1, 9), np.float32))
    rest = rest[0].copy()
    far = engine(0)
    assert far.panda_near_share() == -1
    far.set_world_panda_raw(raw31(P, rest))
    used, shares = [], []
    for _ in range(4):
        far.command(sync_host=True)
        used.append(far.panda_lanes_per_sample_used())
        shares.append(far.panda_near_share())
    assert used == [quiet] * 4 and 0 <= max(shares) < 260, (used, shares)
    # open gripper 3 cm above the cube: the rollouts are next to it all the time
    near = grasp_world(P, sc, close_gripper=False, lift=0.03)
    auto, one = engine(0), engine(1)
    used = []
    for call in range(4):
        for e in (auto, one):
            e.set_world_panda_raw(raw31(P, near))
        a, b = auto.command(sync_host=True), one.command(sync_host=True)
        used.append(auto.panda_lanes_per_sample_used())
        assert np.array_equal(np.asarray(a), np.asarray(b))
        for buf in (L.BUF_TRAJ_COST, L.BUF_MEAN):
            assert torch.equal(auto.buffer(buf), one.buffer(buf))
    assert used == [quiet, many, many, many] and auto.panda_near_share() >= 300, (used, auto.panda_near_share())
    # back with the arm up and the cubes at rest: the quiet form again from the command after the first report from there
    used = []
    for _ in range(3):
        auto.set_world_panda_raw(raw31(P, rest))
        auto.command(sync_host=True)
        used.append(auto.panda_lanes_per_sample_used())
    assert used == [many, quiet, quiet], used
    # the configured initial scene, whose cubes start 1 cm above the table and land: awake cubes count like a near gripper
    for _ in range(2):
        auto.set_world_panda_raw(raw31(P, P.init_world(1)[0]))
        auto.command(sync_host=True)
    assert auto.panda_lanes_per_sample_used() == many and auto.panda_near_share() >= 260


@pytest.mark.gpu
@pytest.mark.parametrize("mm", [False, True])
def test_reach_cost_kernel_equals_the_shadow_slots(oracle, mm):
    """Quirk Q8 two ways: every wavefront re-simulates samples 0 (and K / 2) in shadow slots and reads their cube after each step,
    or -- the default up to K = 8192 -- the rollout runs without shadow slots in the sixteen-lane form, leaves the 17 floats per
    (step, sample) the reach cost reads, and k_panda_reach_cost forms the costs.  Three warm-started commands from a scene in
    which rollouts push the cube around (so that environment 0's cube is NOT the sample's own): every buffer of the command equal
    bit for bit, and equal to the oracle's on the first call."""
    import oracle.panda as P
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd.engine import HipEngine, make_config
    sc = P.default_scene()
    K, T = 300, 20
    delta = np.random.default_rng(21).standard_normal((K, T, 9)).astype(np.float32)
    goal = np.array([0.2, 0.2, 1.115, 0, 0, 0, 1], np.float32)
    w0 = grasp_world(P, sc, close_gripper=False, lift=0.0)
    cfg = P.make_cfg(K, T, multi_modal=mm, task="reach", goal=goal, gripper_cmd=2)
    opl = P.OraclePandaPlanner(cfg, delta, sc)

    def engine(deferred):
        eng = HipEngine(make_config(K=K, T=T, nu=9, env_type="panda_env", multi_modal=mm, u_min=UMIN, u_max=UMAX,
                                    noise_sigma_diag=SIG, lambda_=0.05, pre_height_diff=0.05, dt=0.01))
        eng.set_objective("reach", goal, gripper_cmd=2)
        eng.set_noise(delta)
        eng.set_panda_reach_cost_kernel(deferred)
        eng.set_panda_lanes_per_sample(16 if deferred else 0)
        eng.set_world_panda_raw(raw31(P, w0))
        return eng
    a, b = engine(True), engine(False)
    bufs = [L.BUF_TRAJ_COST, L.BUF_COST_HORIZON, L.BUF_STATES, L.BUF_ACTIONS, L.BUF_MEAN, L.BUF_WEIGHTS, L.BUF_TOP_TRAJS]
    if mm:
        bufs += [L.BUF_MEAN_1, L.BUF_MEAN_2, L.BUF_BEST_1, L.BUF_BEST_2]
    for call in range(3):
        pa, pb = a.command(sync_host=True), b.command(sync_host=True)
        assert a.panda_lanes_per_sample_used() == 16 and b.panda_lanes_per_sample_used() in (1, 8)
        assert np.array_equal(np.asarray(pa), np.asarray(pb))
        for buf in bufs:
            assert torch.equal(a.buffer(buf), b.buffer(buf)), (call, buf)
        if call == 0:
            opl.command(w0)
            np.testing.assert_array_equal(a.cost_horizon.cpu().numpy(), opl.last["cost_h"])
            np.testing.assert_array_equal(a.buffer(L.BUF_TRAJ_COST).cpu().numpy(), opl.last["J"])
    # the cubes did move in some rollouts: the quirk is exercised
    ch = a.cost_horizon.cpu().numpy()
    assert np.isfinite(ch).all() and ch.std() > 0
