"""The update phase alone (m3_update on costs written into TRAJ_COST by the test) against a
python restatement of the reference's weight computation, for cost distributions the rollouts never
produce: spreads of 1e-5 and 1e+8 drive the multi-modal beta search (m3p2i.py:24-64: start at 1,
x0.9 while eta > 10, x1.2 while eta < 3) far beyond the precomputed ladders (0.9^63, 1.2^32), so the
iterative fallback passes run -- in the one-workgroup kernel (K <= 8192) and in the split
k_search / k_apply_weights path (K > 8192).  path="fused" drives m3_update_finalize instead, i.e. what
m3_command runs after its rollout: for K <= 4096 the one-launch k_update_small, whose searches are
iterative from the start."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def search(J):
    """m3p2i.py:24-64 (_multi_modal_exp_util) in float64."""
    J = J.astype(np.float64)
    m, beta, it = J.min(), 1.0, 0
    while True:
        e = np.exp(-(J - m) / beta)
        eta = e.sum()
        it += 1
        if eta > 10:
            beta *= 0.9
        elif eta < 3:
            beta *= 1.2
        else:
            return e / eta, eta, beta, it
        assert it < 5000


@pytest.mark.parametrize("path", ["split", "fused"])
@pytest.mark.parametrize("K", [1500, 4000, 6000, 64000])
@pytest.mark.parametrize("scale", [1e-5, 1.0, 1e8])
def test_multi_modal_weights_on_synthetic_costs(K, scale, path):
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd.engine import HipEngine, make_config
    eng = HipEngine(make_config(K=K, T=12, nu=2, multi_modal=True, u_min=[-3, -3], u_max=[3, 3],
                                noise_sigma_diag=[3, 3]))
    rng = np.random.default_rng(int(K + np.log10(scale) * 7))
    J = (scale * np.abs(rng.standard_normal(K))).astype(np.float32)   # distinct values: the search
    # does not terminate (in the reference either) when > 10 samples tie with the minimum
    eng.buffer(L.BUF_TRAJ_COST).copy_(torch.from_numpy(J))
    if path == "fused":
        eng.update_finalize()
    else:
        eng.update()
    torch.cuda.synchronize()
    info = eng.info()
    half = K // 2
    for buf, JJ, eta, beta, iters in ((L.BUF_WEIGHTS, J, info.eta, None, info.iters),
                                      (L.BUF_WEIGHTS_1, J[:half], info.eta_1, info.beta_1, info.iters_1),
                                      (L.BUF_WEIGHTS_2, J[half:], info.eta_2, info.beta_2, info.iters_2)):
        w_ref, eta_ref, beta_ref, it_ref = search(JJ)
        w = eng.buffer(buf).cpu().numpy()
        assert 3.0 <= eta <= 10.0
        # the pass counts agree unless eta grazes a bound of the window in f32
        assert abs(iters - it_ref) <= 1, (iters, it_ref)
        if iters == it_ref:
            if beta is not None:
                assert abs(beta - beta_ref) <= 2e-5 * beta_ref
            np.testing.assert_allclose(w, w_ref, rtol=5e-3, atol=1e-7)
            assert abs(eta - eta_ref) <= 5e-3 * eta_ref
        assert abs(w.sum() - 1.0) < 1e-4
    if scale != 1.0:   # beyond the shrink ladder (64 entries) / the grow ladder (32): the fallback ran
        assert max(info.iters, info.iters_1, info.iters_2) > (65 if scale < 1 else 34)
    assert info.best_idx == int(np.argmin(J))
    assert info.best_idx_1 == int(np.argmin(J[:half])) and info.best_idx_2 == half + int(np.argmin(J[half:]))
    eng.close()


@pytest.mark.parametrize("K,mm", [(1500, False), (1500, True), (4000, False), (4000, True), (20000, True), (64000, True), (131072, True)])
def test_one_launch_update_equals_the_phases(K, mm):
    """k_update_small (m3_update_finalize, K <= 4096) -- and, multi-modal beyond its range, the three-launch update
    (k_ladder_search, k_regen_part<false>, k_regen_done<false>) -- against m3_update + m3_finalize (the multi-launch
    phases) on the same costs and actions: single mode -> the same bits everywhere (same reductions, same order); the
    multi-modal searches sum eta in a different order than the ladder kernels, so their weights
    agree to rounding and the pass counts unless eta grazes a bound."""
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd.engine import HipEngine, make_config
    rng = np.random.default_rng(K + int(mm))
    T = 12
    J = (30.0 * np.abs(rng.standard_normal(K))).astype(np.float32)
    A = rng.uniform(-3, 3, (T, K, 2)).astype(np.float32)
    mean0 = rng.uniform(-1, 1, (T, 2)).astype(np.float32)
    outs = []
    for path in ("phases", "one"):
        eng = HipEngine(make_config(K=K, T=T, nu=2, multi_modal=mm, u_min=[-3, -3], u_max=[3, 3], noise_sigma_diag=[3, 3]))
        eng.buffer(L.BUF_TRAJ_COST).copy_(torch.from_numpy(J))
        eng.buffer(L.BUF_ACTIONS).copy_(torch.from_numpy(A))
        eng.buffer(L.BUF_STATES).copy_(torch.from_numpy(rng.standard_normal((T, K, 4)).astype(np.float32) * 0 + 1.0))
        eng.buffer(L.BUF_MEAN).copy_(torch.from_numpy(mean0))
        if path == "one":
            eng.update_finalize()
        else:
            eng.update()
            eng.finalize()
        torch.cuda.synchronize()
        info = eng.info()
        bufs = [L.BUF_WEIGHTS, L.BUF_MEAN, L.BUF_ACTION_OUT, L.BUF_TOP_IDX, L.BUF_TOP_TRAJS]
        bufs += [L.BUF_WEIGHTS_1, L.BUF_WEIGHTS_2, L.BUF_MEAN_1, L.BUF_MEAN_2, L.BUF_BEST_1, L.BUF_BEST_2] if mm else [L.BUF_BEST]
        outs.append((info, {b: eng.buffer(b).clone() for b in bufs}))
        eng.close()
    (ia, a), (ib, b) = outs
    assert ia.best_idx == ib.best_idx and ia.best_idx_1 == ib.best_idx_1 and ia.best_idx_2 == ib.best_idx_2
    assert ia.pull_preference == ib.pull_preference
    if not mm:
        assert ia.eta == ib.eta and ia.beta == ib.beta
        for k in a:
            assert torch.equal(a[k], b[k]), k
    else:
        assert (ia.iters, ia.iters_1, ia.iters_2) == (ib.iters, ib.iters_1, ib.iters_2)
        for k in a:
            if a[k].dtype == torch.int32:
                assert torch.equal(a[k], b[k])
            else:
                np.testing.assert_allclose(a[k].cpu().numpy(), b[k].cpu().numpy(), rtol=2e-5, atol=1e-6)
