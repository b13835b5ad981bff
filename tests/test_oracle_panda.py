"""CPU tests of the panda_env oracle: reference cost goldens (G6b), FK known answers from the
URDF, spec sin/cos accuracy, chain/grasp behaviour."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def P():
    import oracle.panda as P
    P.lib()
    return P


def test_spec_sincos_accuracy(P):
    xs = np.linspace(-4.0, 4.0, 4001)
    err = max(max(abs(P.sincos(x)[0] - np.sin(np.float32(x))), abs(P.sincos(x)[1] - np.cos(np.float32(x))))
              for x in xs)
    assert err < 2.5e-7


def test_fk_known_answers_appendix_b(P):
    """Init pose q=[0,0,0,-2,0,1.8675,0,.02,.02], base (-0.45,0,1.125): values computed from the
    URDF with plain numpy in the survey (Appendix B)."""
    sc = P.default_scene()
    L = P.fk(sc, [0, 0, 0, -2, 0, 1.8675, 0, 0.02, 0.02])
    np.testing.assert_allclose(L["pos"][7], [0.1032, 0, 1.6776], atol=2e-4)     # link7
    np.testing.assert_allclose(L["pos"][8], [0.0891, 0, 1.5715], atol=2e-4)     # hand
    np.testing.assert_allclose(L["pos"][9], [0.0954, -0.0141, 1.5118], atol=2e-4)   # left finger
    np.testing.assert_allclose(L["pos"][10], [0.0674, 0.0141, 1.5155], atol=2e-4)   # right finger
    ee = (L["pos"][9] + L["pos"][10]) / 2
    np.testing.assert_allclose(ee, [0.0814, 0, 1.5136], atol=2e-4)
    np.testing.assert_allclose(L["az"][9], [-0.1321, 0, -0.9912], atol=2e-4)
    np.testing.assert_allclose(L["ay"][9], [0.7009, -0.7071, -0.0934], atol=2e-4)
    # initial 10*reach term with cubeA on the table: 4.6583 (cost_functions.py:97-99,114)
    goal = np.array([0.2, -0.2, 1.06 + 0.05])
    assert 10 * np.linalg.norm(ee - goal) == pytest.approx(4.6583, abs=2e-3)
    # quaternions are unit and reproduce the axes
    q = L["quat"][9]
    assert abs(np.linalg.norm(q) - 1) < 1e-6
    x, y, z, w = q
    np.testing.assert_allclose([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)],
                               L["az"][9], atol=1e-6)


@pytest.mark.parametrize("mm", [0, 1])
@pytest.mark.parametrize("task", ["reach", "pick", "place"])
def test_g6b_panda_costs_match_reference(golden, oracle, P, task, mm):
    left, right, cubeA = golden["g6p_left"], golden["g6p_right"], golden["g6p_cubeA"]
    K = left.shape[0]
    half0 = cubeA[K // 2, 3:7] if mm else cubeA[0, 3:7]
    obs = P.make_obs(left[:, :3], left[:, 3:7], right[:, :3], cubeA[:, :3], cubeA[:, 3:7],
                     np.tile(cubeA[0, :3], (K, 1)), np.tile(half0, (K, 1)),
                     golden["g6p_f_table"][:, :2], golden["g6p_f_shelf_stand"][:, :2],
                     golden["g6p_f_cubeB"][:, :2])
    cfg = P.make_cfg(K, 20, multi_modal=bool(mm), task=task, goal=golden["g6p_goal7"])
    c = P.cost_obs(cfg, obs)
    np.testing.assert_allclose(c, golden[f"g6p_cost_{task}_{mm}"], rtol=3e-6, atol=2e-5)


def test_chain_tracks_velocity_targets_and_limits(P):
    sc = P.default_scene()
    w = P.init_world(1)
    u = np.array([[1.0, -0.5, 0.3, 2.0, 1.0, -1.0, 0.5, 1.5, 1.5]], np.float32)
    for _ in range(120):
        P.step_batch(sc, w, u)
    qd = w[0, P.W_QD:P.W_QD + 9]
    np.testing.assert_allclose(qd[:3], u[0, :3], atol=1e-3)          # servo converged
    assert w[0, P.W_Q + 3] == pytest.approx(-0.0698, abs=1e-6) and qd[3] == 0   # joint-4 upper limit
    assert w[0, P.W_Q + 7] == pytest.approx(0.04) and w[0, P.W_Q + 8] == pytest.approx(0.04)  # fingers open
    assert abs(qd[5]) <= 2.61 + 1e-6


def test_cube_settles_on_table_and_can_be_grasped_and_lifted(P):
    sc = P.default_scene()
    w = P.init_world(1)
    for _ in range(30):
        P.step_batch(sc, w, np.zeros((1, 9), np.float32))
    assert w[0, P.W_CUBEA + 2] == pytest.approx(1.025 + 0.025, abs=1e-6)      # resting on the table top
    assert np.all(w[0, P.W_CUBEA + 7:P.W_CUBEA + 13] == 0)
    # put the open gripper around the cube: solve q by crude numeric IK on the oracle FK
    target = w[0, P.W_CUBEA:P.W_CUBEA + 3] + np.array([0, 0, sc.grasp_z])     # hand above the cube
    q = np.array([0, 0.3, 0, -2.2, 0, 2.5, 0.785, 0.04, 0.04], np.float32)
    def feat(L):  # hand position, hand z pointing down, hand y along world y (pads face the cube)
        return np.concatenate([L["pos"][8], 0.3 * L["az"][8], 0.3 * L["ay"][8]])

    want = np.concatenate([target, 0.3 * np.array([0, 0, -1.0]), 0.3 * np.array([0, 1.0, 0])])
    for it in range(600):
        L = P.fk(sc, q)
        e = want - feat(L)
        if np.linalg.norm(e) < 1e-4:
            break
        Jm = np.zeros((9, 7))
        for j in range(7):
            dq = q.copy(); dq[j] += 1e-3
            Jm[:, j] = (feat(P.fk(sc, dq)) - feat(L)) / 1e-3
        q[:7] += (np.linalg.pinv(Jm, rcond=1e-3) @ e * 0.5).astype(np.float32)
        q[:7] = np.clip(q[:7], np.array(sc.qlo)[:7], np.array(sc.qhi)[:7])
    assert np.linalg.norm(e) < 1e-3, "IK did not converge"
    w[0, P.W_Q:P.W_Q + 9] = q
    w[0, P.W_QD:P.W_QD + 9] = 0
    close = np.zeros((1, 9), np.float32); close[0, 7:] = -1.5
    for _ in range(40):
        P.step_batch(sc, w, close)
    assert w[0, P.W_HELD] == 1.0
    assert w[0, P.W_Q + 7] + w[0, P.W_Q + 8] == pytest.approx(0.05, abs=2.1e-3)
    lift = close.copy(); lift[0, 3] = 0.5     # bend the elbow: hand moves, cube follows
    z0 = w[0, P.W_CUBEA + 2]
    for _ in range(50):
        P.step_batch(sc, w, lift)
    assert w[0, P.W_HELD] == 1.0 and abs(w[0, P.W_CUBEA + 2] - z0) > 0.01
    release = np.zeros((1, 9), np.float32); release[0, 7:] = 1.5
    for _ in range(100):
        P.step_batch(sc, w, release)
    assert w[0, P.W_HELD] == 0.0 and w[0, P.W_CUBEA + 2] == pytest.approx(1.05, abs=2e-4)  # fell back (asleep on the table)
    assert w[0, P.W_AWAKE] == 0.0


def test_pad_channel_spec_v11(P):
    """Chain spec v1.1 -- the pad channel: a cube whose centre lies between the two pad faces, inside the pads'
    footprint (|x| <= 2.5 cm, |z - 0.1034| <= 2.5 cm in the hand frame) and aligned with them is SWEPT to the pads'
    centre line by closing fingers (it slides on the table) and held when they meet it; open fingers do not move it;
    a cube outside the footprint, or beyond a pad face, is not captured."""
    from tests.panda_worlds import grasp_world
    sc = P.default_scene()
    close = np.zeros((1, 9), np.float32); close[0, 7:] = -1.5
    opened = np.zeros((1, 9), np.float32); opened[0, 7:] = 1.5

    def run(offset, u, steps=40):
        w = grasp_world(P, sc, close_gripper=False, offset=offset).reshape(1, -1).copy()
        c0 = w[0, P.W_CUBEA:P.W_CUBEA + 3].copy()
        for _ in range(steps):
            P.step_batch(sc, w, u)
        return w[0], c0

    # hand 1.2 cm off the cube along the pads' closing direction: swept to the centre line, then held
    w, c0 = run((0.0, 0.012), close)
    hand_y = c0[1] + 0.012
    assert w[P.W_HELD] == 1.0
    assert w[P.W_CUBEA + 1] == pytest.approx(hand_y, abs=2e-4) and w[P.W_CUBEA] == pytest.approx(c0[0], abs=1e-4)
    assert w[P.W_Q + 7] + w[P.W_Q + 8] == pytest.approx(0.05, abs=2.1e-3)
    assert abs(w[P.W_RELP + 1]) < 1e-6 and w[P.W_CUBEA + 2] == pytest.approx(c0[2], abs=1e-6)   # slid on the table
    # the same pose with the fingers commanded open: nothing touches the cube
    w, c0 = run((0.0, 0.012), opened)
    assert w[P.W_HELD] == 0.0 and np.array_equal(w[P.W_CUBEA:P.W_CUBEA + 3], c0)
    # 2 cm off along the pads' width (inside the footprint): held off-centre, not moved
    w, c0 = run((0.02, 0.0), close)
    assert w[P.W_HELD] == 1.0 and abs(abs(w[P.W_RELP]) - 0.02) < 1e-3 and np.allclose(w[P.W_CUBEA:P.W_CUBEA + 2], c0[:2], atol=1e-4)
    # 3 cm off along the width: outside the pads' footprint -- not captured.  (Spec v2.1: the pads still MEET the cube's
    # side faces there -- 1 cm of pad overlaps them -- and stop at its width; they hold nothing and do not move it.)
    w, c0 = run((0.03, 0.0), close)
    assert w[P.W_HELD] == 0.0 and np.allclose(w[P.W_CUBEA:P.W_CUBEA + 3], c0, atol=1e-4)
    assert w[P.W_Q + 7] + w[P.W_Q + 8] == pytest.approx(2 * 0.025, abs=1e-4)
    # 4 cm off along the width: beyond the pads altogether (but inside the capture volume, in which the tips' spheres do
    # not act on cubeA): the fingers close past the cube's side, which stays where it is
    w, c0 = run((0.04, 0.0), close)
    assert w[P.W_HELD] == 0.0 and np.allclose(w[P.W_CUBEA:P.W_CUBEA + 3], c0, atol=1e-4)
    assert w[P.W_Q + 7] + w[P.W_Q + 8] < 0.01
    # 4.5 cm off along the closing direction: the cube's centre is beyond a pad face -- not captured (pushed aside
    # by the closing finger instead)
    w, c0 = run((0.0, 0.045), close)
    assert w[P.W_HELD] == 0.0 and np.allclose(w[P.W_CUBEA:P.W_CUBEA + 3], c0, atol=3e-2)
    # 3 cm above the grasp height: out of reach of the pads
    w = grasp_world(P, sc, close_gripper=False, lift=0.03).reshape(1, -1).copy()
    for _ in range(40):
        P.step_batch(sc, w, close)
    assert w[0, P.W_HELD] == 0.0


PANDA_TRACES = {"panda_reach": ("reach", False, 1), "panda_reachmm": ("reach", True, 1), "panda_pick": ("pick", False, 2),
                # quirk Q8: open gripper around cubeA, the rollouts (environment 0's too) push it; every rollout's reach
                # cost is measured against environment 0's cube (cost_functions.py:97, skill_utils.py:274)
                "panda_reach_touch": ("reach", False, 1), "panda_reachmm_touch": ("reach", True, 1)}


@pytest.mark.parametrize("tag", list(PANDA_TRACES))
def test_g9_panda_command_traces_vs_reference(golden, oracle, tag):
    """G9 (panda): five consecutive command() calls of the REFERENCE's M3P2I + Objective (imported by
    tests/golden/make_golden.py, the oracle's chain dynamics behind its wrapper API) vs the oracle's own
    planner from the same worlds and noise: returned plan, weights, top trajectories, the persistent beta
    the panda env adapts (mppi.py:446-454), gripper override (mppi.py:412-416)."""
    import oracle.panda as P
    task, mm, grip = PANDA_TRACES[tag]
    K, T = 256, 20
    goal = np.array([0.2, 0.2, 1.115, 0, 0, 0, 1], np.float32)
    cfg = P.make_cfg(K, T, multi_modal=mm, task=task, goal=goal, gripper_cmd=grip)
    opl = P.OraclePandaPlanner(cfg, golden[f"g9_{tag}_delta"])
    touch = tag.endswith("_touch")
    for call, w in enumerate(golden[f"g9_{tag}_world"]):
        a = opl.command(w)
        # In the touch traces EVERY rollout's cost hangs on environment 0's cube, whose flight after the finger's blow is
        # one contact history: from the second call on the 1e-7 between torch's controls and the oracle's decides
        # which way it tumbles, and the plans part (observed: 6e-5 / 2e-3 at call 1, 0.05 at call 3).  The first call pins
        # the semantics: with each sample's own cube instead, J differs by up to 15 and the plan by far more than 1e-3.
        tol = 1e-3 if (call == 0 or not touch) else 0.15
        np.testing.assert_allclose(a, golden[f"g9_{tag}_action"][call], atol=tol, err_msg=f"{tag} call {call}")
        np.testing.assert_allclose(opl.last["w"], golden[f"g9_{tag}_weights"][call], atol=tol)
        np.testing.assert_allclose(opl.mean, golden[f"g9_{tag}_mean"][call], atol=tol)
        if not mm and not (touch and call):
            assert opl.beta == pytest.approx(float(golden[f"g9_{tag}_beta"][call]), rel=1e-5)
        top = opl.last["states"][opl.last["top_idx"]][:, :, [0, 2]]
        if not (touch and call):
            np.testing.assert_allclose(top[:5], golden[f"g9_{tag}_top_trajs"][call][:5], atol=1e-3)
        if touch and call == 0:      # the quirk is live in this trace: own-cube costs would be different numbers
            act = opl.last["actions"] / cfg.u_scale
            own = np.concatenate([P.rollout(cfg, opl.sc, w, act[:K // 2], 0, K // 2)["J"],
                                  P.rollout(cfg, opl.sc, w, act[K // 2:], K // 2, K)["J"]])
            assert np.abs(own - opl.last["J"]).max() > 1.0
        # the gripper override: fingers commanded open (reach) / closed (pick) in every sample
        assert np.all(opl.last["actions"][:-1, :, 7:] == (1.5 if grip == 1 else -1.5))
    if touch:
        return
    np.testing.assert_allclose(opl.last["actions"], golden[f"g9_{tag}_actions_last"], atol=1e-3)
    # the rollouts themselves: equal for all but a few samples -- the reference forms its controls in torch (1e-7
    # apart from the oracle's), and with spec v2 a rollout that pushes the held cube into the table or sweeps a finger
    # past cubeB turns such a difference into a different contact history (unilateral contacts are not continuous)
    bad = np.abs(opl.last["states"] - golden[f"g9_{tag}_states_last"]).max(axis=(1, 2)) > 1e-3
    assert bad.mean() < 0.08, f"{int(bad.sum())} of {bad.size} rollouts differ"      # (observed: 11 of 256 in the pick trace, controls 4e-6 apart)


PANDA_SIG = [[0.0] * 9 for _ in range(9)]
for _i in range(7):
    PANDA_SIG[_i][_i] = 10.0
PANDA_SIG[7][7] = PANDA_SIG[8][8] = 0.8
PANDA_SIG[0][1] = PANDA_SIG[1][0] = 4.0
PANDA_SIG[2][5] = PANDA_SIG[5][2] = -3.0
PANDA_MU = [0.2, -0.1, 0.0, 0.1, 0.0, 0.0, -0.2, 0.0, 0.0]
# the MPPIConfig switches no shipped config turns on, on the panda_env (reference traces: make_golden.py g11)
PANDA_OPT = {
    "panda_opt_cov": dict(K=256, T=20, update_cov=True),
    "panda_opt_rand": dict(K=128, T=12, noise_mu=PANDA_MU, noise_sigma=PANDA_SIG),
    "panda_opt_simple": dict(K=128, T=12, mode_simple=True, u_per_command=12, lambda_=0.05, noise_mu=PANDA_MU,
                             noise_sigma=PANDA_SIG, noise_abs_cost=True, u_scale=0.9),
}


def panda_opt_planner(golden, tag, seed=7):
    import oracle.panda as P
    kw = dict(PANDA_OPT[tag])
    update_cov = kw.pop("update_cov", False)
    goal = np.array([0.2, 0.2, 1.115, 0, 0, 0, 1], np.float32)
    cfg = P.make_cfg(kw.pop("K"), kw.pop("T"), task="reach", goal=goal, gripper_cmd=1, **kw)
    delta = golden[f"g9_{tag}_delta"] if f"g9_{tag}_delta" in golden else None
    return cfg, P.OraclePandaPlanner(cfg, delta, seed=seed, update_cov=update_cov)


@pytest.mark.parametrize("tag", list(PANDA_OPT))
def test_g11_panda_option_traces_vs_reference(golden, oracle, tag):
    """update_cov; sampling_method='random' with a noise mean and a non-diagonal noise_sigma; mppi_mode='simple'
    with noise_abs_cost and u_scale != 1 -- the reference's planner on the panda_env vs the oracle's."""
    cfg, opl = panda_opt_planner(golden, tag)
    for call, w in enumerate(golden[f"g9_{tag}_world"]):
        a = opl.command(w)
        np.testing.assert_allclose(a, golden[f"g9_{tag}_action"][call], atol=1e-3, err_msg=f"{tag} call {call}")
        np.testing.assert_allclose(opl.last["w"], golden[f"g9_{tag}_weights"][call], atol=1e-3)
        np.testing.assert_allclose(opl.U if cfg.mode_simple else opl.mean, golden[f"g9_{tag}_mean"][call], atol=1e-3)
        if cfg.mode_simple:
            np.testing.assert_allclose(opl.last["cost_total"], golden[f"g9_{tag}_J"][call], rtol=1e-5, atol=1e-3)
        else:
            assert opl.beta == pytest.approx(float(golden[f"g9_{tag}_beta"][call]), rel=1e-5)
        if f"g9_{tag}_extra" in golden:
            np.testing.assert_allclose([cfg.scale_tril[j] for j in range(9)], golden[f"g9_{tag}_extra"][call], rtol=1e-4)
    np.testing.assert_allclose(opl.last["states"], golden[f"g9_{tag}_states_last"], atol=1e-3)
    np.testing.assert_allclose(opl.last["actions"] / np.float32(cfg.u_scale), golden[f"g9_{tag}_actions_last"], atol=1e-3)


def test_lane_butterflies_reproduce_the_spec_summation_trees(oracle):
    """World spec v3 defines a gripper row's velocity as SUM16 and a manifold row's as SUM8 -- fixed pairwise trees over rounded
    products (oracle/panda_chain.c) -- because the product's kernel forms them ACROSS LANES: a butterfly of DPP steps that pair
    lanes symmetrically (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), so that every lane of a sample ends with the same
    bits.  Emulated here step by step in binary32: all sixteen (eight) lanes equal the oracle's tree bit for bit, for sixteen
    lanes per sample, for eight (two coordinates per lane: the halves are added last) and for the cubes' half-row sums with their
    two +0 pads; signed zeros included."""
    import ctypes as C
    lib = oracle.load()
    lib.m3o_sum16.restype = C.c_float
    lib.m3o_sum16.argtypes = [C.POINTER(C.c_float)]
    lib.m3o_sum8_6.restype = C.c_float
    lib.m3o_sum8_6.argtypes = [C.POINTER(C.c_float)]
    f32 = np.float32
    rng = np.random.default_rng(11)

    def butterfly(v, steps):
        v = v.astype(f32).copy()
        n = len(v)
        lanes = np.arange(n)
        perms = {"xor1": lanes ^ 1, "xor2": lanes ^ 2, "half_mirror": (lanes & ~7) | (7 - (lanes & 7)), "mirror": 15 - lanes}
        for st in steps:
            v = (v + v[perms[st]]).astype(f32)          # v_add_f32_dpp: every lane adds its partner's value
        return v

    def bits(x):
        return np.asarray(x, f32).view(np.uint32)

    for trial in range(2000):
        x = (rng.standard_normal(16) * 10.0 ** rng.integers(-6, 4, 16)).astype(f32)
        if trial % 5 == 0:
            x[rng.integers(0, 16, 6)] = f32(0.0) * rng.choice([-1.0, 1.0], 6).astype(f32)     # signed zeros
        if trial % 7 == 0:
            x[9:] = 0.0                                                                     # a row without a free target
        want = lib.m3o_sum16(x.ctypes.data_as(C.POINTER(C.c_float)))
        got16 = butterfly(x, ["xor1", "xor2", "half_mirror", "mirror"])
        assert (bits(got16) == bits(want)).all(), (trial, x)
        # eight lanes per sample: element 0 = coordinates 0-7, element 1 = 8-15; three steps each, then the two elements
        e0, e1 = butterfly(x[:8], ["xor1", "xor2", "half_mirror"]), butterfly(x[8:], ["xor1", "xor2", "half_mirror"])
        assert (bits((e0 + e1).astype(f32)) == bits(want)).all(), (trial, x)
        # a cube's half row: six products and two pads of +0
        h = np.concatenate([x[:6], np.zeros(2, f32)])
        want8 = lib.m3o_sum8_6(h[:6].ctypes.data_as(C.POINTER(C.c_float)))
        assert (bits(butterfly(h, ["xor1", "xor2", "half_mirror"])) == bits(want8)).all(), (trial, h)
