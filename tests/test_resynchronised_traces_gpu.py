"""Every call of a warm-started trace checked at BIT level.

The multi-call parity tests compare later calls loosely by construction: two correct implementations of the update agree to
~1e-6 in the new means, a rollout in contact turns that into another contact history, so from call 1 on only 'all but a few
rollouts' can be asserted -- and a wrong warm-start shift that perturbs less than that would pass.  Here the two sides are
RE-SYNCHRONISED after every call: the oracle's warm-start state (means, best trajectories, beta) is written into the handle
through the product's own restore API (m3_set_plan / m3_set_beta: what a caller saving and restoring a planner uses), so call
c + 1 starts from identical bits on both sides and its shift, action assembly (best-trajectory rows, null action, gripper
override), rollout, costs and trajectory costs must be IDENTICAL -- at every call, not just the first.  The pending suction
forces carry over inside each side (bit-equal because the rollouts are).

Panda: the kernel form is forced per call and cycles 1 / 8 / 16 lanes per sample (and, for reach, the deferred cost kernel
against the shadow slots) within one trace -- the forms must be interchangeable call by call, J and cost_horizon bit-equal."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from tests.test_hip_parity_point import G9, make_pair, raw_world  # noqa: E402


def resync(eng, opl, L):
    for buf, val in ((L.BUF_MEAN, opl.mean), (L.BUF_MEAN_1, opl.mean1), (L.BUF_MEAN_2, opl.mean2), (L.BUF_BEST, opl.best),
                     (L.BUF_BEST_1, opl.best1), (L.BUF_BEST_2, opl.best2)):
        eng.set_plan(buf, val)
    eng.set_beta(opl.beta)


def assert_rollout_bits(eng, opl, L, where):
    np.testing.assert_array_equal(eng.actions.cpu().numpy(), opl.last["actions"], err_msg=where + " actions")
    np.testing.assert_array_equal(eng.states.cpu().numpy(), opl.last["states"], err_msg=where + " states")
    np.testing.assert_array_equal(eng.cost_horizon.cpu().numpy(), opl.last["cost_h"], err_msg=where + " cost_horizon")
    np.testing.assert_array_equal(eng.buffer(L.BUF_TRAJ_COST).cpu().numpy(), opl.last["J"], err_msg=where + " J")


@pytest.mark.parametrize("tag", ["push", "pushc", "pull", "hybrid", "opt_uscale"])
def test_point_trace_is_bit_exact_at_every_call_when_resynchronised(golden, oracle, tag):
    from m3p2i_aip_amd import _lib as L
    eng, opl = make_pair(oracle, golden, tag)
    worlds = golden[f"g9_{tag}_world"]
    mm = bool(G9[tag].get("multi_modal"))
    for call in range(worlds.shape[0]):
        eng.set_world_point_raw(raw_world(worlds[call]))
        a_hip = eng.command(sync_host=True)
        a_orc = opl.command(worlds[call])
        assert_rollout_bits(eng, opl, L, f"{tag} call {call}")
        # the update on identical costs: weights of every sample, the plan, and what the searches decided
        np.testing.assert_allclose(eng.buffer(L.BUF_WEIGHTS).cpu().numpy(), opl.last["w"], rtol=2e-3, atol=1e-6)
        np.testing.assert_allclose(a_hip, a_orc, atol=1e-4, err_msg=f"{tag} call {call}")
        i, oi = eng.info(), opl.last["info"]
        if mm:
            assert (i.iters_1, i.iters_2, i.iters) == (oi.iters_1, oi.iters_2, oi.iters)
            # (m3_info holds GLOBAL sample indices, the oracle's second-mode index counts from the start of that half)
            assert (i.best_idx_1, i.best_idx_2) == (oi.best_idx_1, oi.best_idx_2 + opl.cfg.K // 2)
            assert i.pull_preference == opl.pull_preference()
        else:
            assert i.best_idx == oi.best_idx
        resync(eng, opl, L)
    eng.close()


PANDA_UMIN, PANDA_UMAX, PANDA_SIG = [-2.0] * 7 + [-1.5] * 2, [2.0] * 7 + [1.5] * 2, [10.0] * 7 + [0.8] * 2


@pytest.mark.parametrize("tag,task,mm,grip", [("panda_reach", "reach", False, 1), ("panda_reachmm", "reach", True, 1),
                                              ("panda_pick", "pick", False, 2), ("panda_reach_touch", "reach", False, 1),
                                              ("panda_reachmm_touch", "reach", True, 1)])
def test_panda_trace_is_bit_exact_at_every_call_in_a_cycle_of_kernel_forms(golden, oracle, tag, task, mm, grip):
    import oracle.panda as P
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd.engine import HipEngine, make_config
    K, T = 256, 20
    goal = np.array([0.2, 0.2, 1.115, 0, 0, 0, 1], np.float32)
    cfg = P.make_cfg(K, T, multi_modal=mm, task=task, goal=goal, gripper_cmd=grip)
    opl = P.OraclePandaPlanner(cfg, golden[f"g9_{tag}_delta"])
    eng = HipEngine(make_config(K=K, T=T, nu=9, env_type="panda_env", multi_modal=mm, u_min=PANDA_UMIN, u_max=PANDA_UMAX,
                                noise_sigma_diag=PANDA_SIG, lambda_=0.05, pre_height_diff=0.05, dt=0.01))
    eng.set_objective(task, goal, gripper_cmd=grip)
    eng.set_noise(golden[f"g9_{tag}_delta"])
    # (lanes per sample, reach cost kernel): every form the library can launch, one per call, round and round
    forms = [(16, True), (1, True), (8, False), (16, False), (8, True), (1, False)]
    worlds = golden[f"g9_{tag}_world"]
    used = set()
    for call, w in enumerate(worlds):
        lps, deferred = forms[call % len(forms)]
        eng.set_panda_lanes_per_sample(lps)
        eng.set_panda_reach_cost_kernel(deferred)
        eng.set_world_panda_raw(P.raw57(w))
        a_hip = eng.command(sync_host=True)
        a_orc = opl.command(w)
        assert eng.lib.m3_panda_lanes_per_sample_used(eng._h) == lps
        used.add(lps)
        assert_rollout_bits(eng, opl, L, f"{tag} call {call} lps {lps} deferred {deferred}")
        np.testing.assert_allclose(eng.buffer(L.BUF_WEIGHTS).cpu().numpy(), opl.last["w"], rtol=2e-3, atol=1e-6)
        np.testing.assert_allclose(a_hip, a_orc, atol=1e-4, err_msg=f"{tag} call {call}")
        i, oi = eng.info(), opl.last["info"]
        if not mm:
            assert i.best_idx == oi.best_idx and i.beta == pytest.approx(opl.beta, rel=1e-6)
        else:
            assert (i.iters_1, i.iters_2, i.iters) == (oi.iters_1, oi.iters_2, oi.iters)
        resync(eng, opl, L)
    assert used == {1, 8, 16} or len(worlds) < 3
    eng.close()
