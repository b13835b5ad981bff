"""Full-size checks (BASELINE.json configs C2..C5 on one GPU) through properties that do not
need the CPU oracle to run at that size: every relation below is re-derived with torch / scipy
from the library's own buffers after command(), so it is an independent implementation of the
reference's formulas (file:line given per check).

  J        = sum_t gamma^t cost_horizon[k, t]                     mppi_utils.py:106-113, mppi.py:313
  weights  = softmin(J) with the reported beta / eta              mppi.py:430-456 / m3p2i.py:24-64
  sum w    = 1 (per weight set), eta in the search window [3, 10] (multi-modal)
  top_idx  = the 20 smallest costs, ascending; top_trajs = their states    mppi.py:248-254
  mean     = (1 - 0.98) * shift(mean_old) + 0.98 * sum_k w_k a_k   mppi.py:266-273,497-503
  mean_1/2 = sum_k w1_k a_k / sum_k w2_k a_k; best_1/2 = actions of the per-mode argmax  m3p2i.py:77-87
  action   = savgol_filter(mean, 9, 2)                             mppi.py:257-263 (scipy's own filter)
  actions within [u_min, u_max]; the null-action sample; finite states
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

CONFIGS = {
    "C2_push_K2000_T30": dict(K=2000, T=30, nu=2, env="point_env", task="push", goal=(-1.0, -1.0), mm=False),
    "C3_hybrid_K4000_T30": dict(K=4000, T=30, nu=2, env="point_env", task="push_pull", goal=(-3.75, -3.75), mm=True),
    "C4_panda_K4000_T20": dict(K=4000, T=20, nu=9, env="panda_env", task="reach",
                               goal=(0.2, 0.2, 1.115, 0.0, 0.0, 0.0, 1.0), mm=False),
    "C5_hybrid_K64000_T30": dict(K=64000, T=30, nu=2, env="point_env", task="push_pull", goal=(-3.75, -3.75), mm=True),
    "push_K6000_T30": dict(K=6000, T=30, nu=2, env="point_env", task="push", goal=(-1.0, -1.0), mm=False),
    "hybrid_K6000_T30": dict(K=6000, T=30, nu=2, env="point_env", task="push_pull", goal=(-3.75, -3.75), mm=True),
    "push_K40000_T30": dict(K=40000, T=30, nu=2, env="point_env", task="push", goal=(-1.0, -1.0), mm=False),
    "northstar_push_K10000_T30": dict(K=10000, T=30, nu=2, env="point_env", task="push", goal=(-1.0, -1.0), mm=False),
    # the throughput-bound regime: the two-waves-per-SIMD build of the rollout kernel, the large-K update (local softmins
    # on many workgroups, 64 / 256 top-k candidate lists merged by stage B)
    "saturated_push_K262144_T30": dict(K=262144, T=30, nu=2, env="point_env", task="push", goal=(-1.0, -1.0), mm=False),
    "saturated_hybrid_K262144_T30": dict(K=262144, T=30, nu=2, env="point_env", task="push_pull", goal=(-3.75, -3.75), mm=True),
    "saturated_push_K1048576_T30": dict(K=1048576, T=30, nu=2, env="point_env", task="push", goal=(-1.0, -1.0), mm=False),
}


def build(c, seed=5):
    from m3p2i_aip_amd.engine import HipEngine, make_config
    if c["env"] == "point_env":
        kw = dict(u_min=[-3, -3], u_max=[3, 3], noise_sigma_diag=[3, 3])
    else:
        kw = dict(u_min=[-2.0] * 7 + [-1.5] * 2, u_max=[2.0] * 7 + [1.5] * 2, noise_sigma_diag=[10.0] * 7 + [0.8] * 2,
                  lambda_=0.05, dt=0.01)
    eng = HipEngine(make_config(K=c["K"], T=c["T"], nu=c["nu"], env_type=c["env"], multi_modal=c["mm"], **kw))
    eng.set_objective(c["task"], c["goal"], 1 if c["env"] == "panda_env" else 0)
    # synthetic smooth noise (a cheap stand-in for the Halton spline: the properties do not
    # depend on the sample values, only on their being distinct): random knots, linear interpolation
    g = torch.Generator().manual_seed(seed)
    nk = c["T"] // 4
    knots = torch.randn(c["K"], c["nu"], nk, generator=g)
    delta = torch.nn.functional.interpolate(knots, size=c["T"], mode="linear", align_corners=True)
    eng.set_noise(delta.permute(0, 2, 1).contiguous().numpy())
    return eng


def _compare_sampled_rows_with_the_oracle(eng, c, st, act, ch, J):
    """The saturated regime (two / three resident waves per SIMD: the occ2 / occ3 builds of the rollout kernel): 4096
    rows of the first command -- the first and last 1024 (samples 0 and K-1), 1024 around K/2 (the mode boundary) and 1024
    at a random place -- recomputed by the CPU oracle from the same noise rows; states, actions, step costs and J must
    be the kernel's bit for bit.  (First command: the means are zero, so a row's controls are its own noise row.)"""
    import oracle as O
    from m3p2i_aip_amd import _lib as L
    K, T, nu = c["K"], c["T"], c["nu"]
    cfg = O.make_cfg(K, T, nu, task=c["task"], goal=c["goal"], multi_modal=c["mm"])
    sc = O.default_scene()
    w0 = O.init_world(1)[0]
    z = np.zeros((T, nu), np.float32)
    noise = eng.buffer(L.BUF_NOISE)                      # [T, K, nu]
    rnd = int(np.random.default_rng(K).integers(2048, K // 2 - 2048))
    table = np.zeros((K, T, nu), np.float32)
    for k0 in (0, K // 2 - 512, rnd, K - 1024):
        k1 = k0 + 1024
        delta = noise[:, k0:k1].permute(1, 0, 2).contiguous().cpu().numpy()
        # controls of rows k0..k1: clamp(mean + delta * scale) with the special rows (best trajectories at 0 and K/2:
        # zero on the first command; the null action at K-1) -- the oracle's own assembly on a K-row table that
        # holds these rows at their global places
        table[k0:k1] = delta
        act_o = O.assemble_actions(cfg, table, z, z, z, z, z, k0, k1)
        r = O.point_rollout(cfg, sc, w0, act_o, None, k0, k1)
        for got, want, what in ((st[k0:k1], r["states"], "states"), (act[k0:k1], r["actions"], "actions"),
                                (ch[k0:k1], r["cost_h"], "cost_h"), (J[k0:k1], r["J"], "J")):
            g = got.cpu().numpy()
            assert np.array_equal(g.view(np.uint32), want.view(np.uint32)), f"rows {k0}..{k1}: {what} differ from the oracle"


@pytest.mark.parametrize("name", list(CONFIGS))
def test_command_properties_at_full_size(name):
    import scipy.signal
    from m3p2i_aip_amd import _lib as L
    c = CONFIGS[name]
    K, T, nu, mm = c["K"], c["T"], c["nu"], c["mm"]
    half = K // 2
    eng = build(c)
    f64 = torch.float64
    for call in range(3):
        mean_old = eng.buffer(L.BUF_MEAN).clone()
        eng.command()
        torch.cuda.synchronize()
        info = eng.info()
        J = eng.buffer(L.BUF_TRAJ_COST)
        ch = eng.cost_horizon                       # [K, T]
        st, act = eng.states, eng.actions           # [K, T, 4], [K, T, nu]
        assert torch.isfinite(st).all() and torch.isfinite(J).all()
        if K >= 262144 and call == 0:
            _compare_sampled_rows_with_the_oracle(eng, c, st, act, ch, J)
        # discounted accumulation
        gam = torch.tensor(0.95, dtype=f64, device=J.device) ** torch.arange(T, device=J.device, dtype=f64)
        np.testing.assert_allclose((ch.to(f64) * gam).sum(1).cpu().numpy(), J.to(f64).cpu().numpy(), rtol=2e-5)
        # bounds, null action (sample K-1: zero noise AND zero command, mppi.py:300-302,392)
        lo = torch.tensor(list(eng.cfg.u_min)[:nu], device=J.device)
        hi = torch.tensor(list(eng.cfg.u_max)[:nu], device=J.device)
        assert (act >= lo - 1e-6).all() and (act <= hi + 1e-6).all()
        assert (act[K - 1] == 0).all()
        # weights
        w = eng.buffer(L.BUF_WEIGHTS).to(f64)
        Jd = J.to(f64)
        assert abs(w.sum().item() - 1.0) < 1e-5
        if mm:
            w1, w2 = eng.buffer(L.BUF_WEIGHTS_1).to(f64), eng.buffer(L.BUF_WEIGHTS_2).to(f64)
            assert abs(w1.sum().item() - 1.0) < 1e-5 and abs(w2.sum().item() - 1.0) < 1e-5
            for eta in (info.eta, info.eta_1, info.eta_2):      # the search window, m3p2i.py:35-51
                assert 3.0 <= eta <= 10.0, (info.eta, info.eta_1, info.eta_2)
            for ww, JJ, beta, eta in ((w1, Jd[:half], info.beta_1, info.eta_1), (w2, Jd[half:], info.beta_2, info.eta_2)):
                ref = torch.exp(-(JJ - JJ.min()) / beta)
                np.testing.assert_allclose((ref / ref.sum()).cpu().numpy(), ww.cpu().numpy(), rtol=2e-3, atol=1e-9)
                assert abs(ref.sum().item() - eta) <= 2e-3 * eta
            assert info.pull_preference == int(w[half:].sum() > w[:half].sum())   # m3p2i.py:18-21
        else:
            beta_used = 1.0     # point_env never adapts beta; panda starts at 1 and adapts AFTER use
            if c["env"] == "panda_env" and call > 0:
                beta_used = beta_prev
            ref = torch.exp(-(Jd - Jd.min()) / beta_used)
            np.testing.assert_allclose((ref / ref.sum()).cpu().numpy(), w.cpu().numpy(), rtol=2e-3, atol=1e-9)
            assert abs(ref.sum().item() - info.eta) <= 2e-3 * info.eta
            if c["env"] == "panda_env":     # mppi.py:446-454
                want = beta_used * (0.9 if info.eta > 20 else 1.2 if info.eta < 10 else 1.0)
                assert abs(info.beta - want) < 1e-6 * want
            beta_prev = info.beta
        # top-k
        top = eng.buffer(L.BUF_TOP_IDX).to(torch.int64)
        order = torch.argsort(J, stable=True)[:20]
        assert torch.equal(J[top], J[order])
        np.testing.assert_array_equal(eng.buffer(L.BUF_TOP_TRAJS).cpu().numpy(),
                                      st[top][:, :, [0, 2]].cpu().numpy())
        # mean update (+ per-mode means / best trajectories)
        a64 = act.to(f64)
        shift = torch.cat([mean_old[1:], mean_old[-1:]]).to(f64)
        if mm:
            np.testing.assert_allclose(eng.buffer(L.BUF_MEAN_1).cpu().numpy(),
                                       torch.einsum("k,ktj->tj", w1, a64[:half]).cpu().numpy(), atol=2e-5)
            np.testing.assert_allclose(eng.buffer(L.BUF_MEAN_2).cpu().numpy(),
                                       torch.einsum("k,ktj->tj", w2, a64[half:]).cpu().numpy(), atol=2e-5)
            assert torch.equal(eng.buffer(L.BUF_BEST_1), act[int(torch.argmax(w1))])
            assert torch.equal(eng.buffer(L.BUF_BEST_2), act[half + int(torch.argmax(w2))])
            assert info.best_idx_1 == int(torch.argmin(J[:half])) and info.best_idx_2 == half + int(torch.argmin(J[half:]))
        else:
            assert torch.equal(eng.buffer(L.BUF_BEST), act[int(torch.argmin(J))])
            assert info.best_idx == int(torch.argmin(J))
        mean_new = 0.02 * shift + 0.98 * torch.einsum("k,ktj->tj", w, a64)
        np.testing.assert_allclose(eng.buffer(L.BUF_MEAN).cpu().numpy(), mean_new.cpu().numpy(), atol=3e-5)
        # returned plan = scipy's Savitzky-Golay filter of the new mean
        want = scipy.signal.savgol_filter(eng.buffer(L.BUF_MEAN).cpu().numpy().astype(np.float64), 9, 2, axis=0)
        np.testing.assert_allclose(eng.buffer(L.BUF_ACTION_OUT).cpu().numpy(), want, atol=1e-5)
    eng.close()


def test_rollout_is_deterministic_and_sample_local():
    """Same inputs -> same bits; and a sample's trajectory does not depend on which other samples
    are in the batch (the reference's envs are independent): rows of a K=4000 run equal the
    corresponding rows of a K=2000 run given the same noise rows and the same (zero) means."""
    c = dict(CONFIGS["C2_push_K2000_T30"])
    big = build(dict(c, K=4000), seed=9)
    big.command()
    a = big.states.clone()
    big.reset()
    big.command()
    assert torch.equal(a, big.states)
    from m3p2i_aip_amd import _lib as L
    small = build(c, seed=9)
    small.set_noise(big.buffer(L.BUF_NOISE).permute(1, 0, 2)[1000:3000].contiguous().cpu().numpy())
    small.command()
    # sample K-1 of each batch is the null-action sample: compare all but that one
    assert torch.equal(small.states[:-1], a[1000:2999])
    big.close()
    small.close()


@pytest.mark.parametrize("name", ["C2_push_K2000_T30", "C3_hybrid_K4000_T30"])
def test_wave_order_does_not_change_results(name):
    """m3_set_wave_order: the samples are sorted into spatially coherent wavefronts (default) or
    assigned by index.  Lane placement must not change a single bit of any output."""
    from m3p2i_aip_amd import _lib as L
    c = CONFIGS[name]
    outs = []
    for on in (True, False):
        eng = build(c, seed=11)
        eng.set_wave_order(on)
        for _ in range(3):
            eng.command()
        torch.cuda.synchronize()
        outs.append([eng.buffer(b).clone() for b in (L.BUF_TRAJ_COST, L.BUF_STATES, L.BUF_ACTIONS, L.BUF_COST_HORIZON,
                                                      L.BUF_WEIGHTS, L.BUF_MEAN, L.BUF_ACTION_OUT, L.BUF_TOP_IDX,
                                                      L.BUF_TOP_TRAJS, L.BUF_PENDING_FORCE)])
        eng.close()
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("name", ["C2_push_K2000_T30", "C3_hybrid_K4000_T30"])
def test_relabelled_samples_are_the_same_sample_set(name):
    """m3_relabel_samples permutes the noise rows once into wavefront order (within each mode's half,
    the rows of samples 0, K/2 and K-1 in place).  The sample SET must be unchanged: the trajectory
    costs of the relabelled run are a permutation of the others (bit for bit), the special samples
    keep theirs, the noise buffer holds the same rows, and the plan agrees to summation order."""
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd import sampling
    c = CONFIGS[name]
    K, T, half = c["K"], c["T"], c["K"] // 2
    knots = sampling.halton_knots(K, T, 2)
    runs = []
    for relabel in (False, True):
        eng = build(c)
        eng.set_wave_order(False)      # by index in both runs: only the labels differ
        eng.set_noise_knots(knots, 2, 0.5)
        eng.set_wave_order(True)
        if relabel:
            eng.relabel_samples()
        else:
            eng.set_wave_order(False)
        eng.command()
        torch.cuda.synchronize()
        runs.append(dict(J=eng.buffer(L.BUF_TRAJ_COST).clone(), noise=eng.buffer(L.BUF_NOISE).clone(),
                         plan=eng.buffer(L.BUF_ACTION_OUT).clone(), w=eng.buffer(L.BUF_WEIGHTS).clone(),
                         topJ=eng.buffer(L.BUF_TRAJ_COST)[eng.buffer(L.BUF_TOP_IDX).long()].clone()))
        eng.close()
    a, b = runs
    assert not torch.equal(a["J"], b["J"])                                   # labels did move
    for lo, hi in (((0, half), (half, K)) if c["mm"] else ((0, K),)):          # ... inside each mode's half only
        assert torch.equal(torch.sort(a["J"][lo:hi]).values, torch.sort(b["J"][lo:hi]).values)
    for k in (0, half, K - 1):
        assert a["J"][k] == b["J"][k]
        assert torch.equal(a["noise"][:, k], b["noise"][:, k])
    rows_a = {bytes(r.cpu().numpy().tobytes()) for r in a["noise"].permute(1, 0, 2)}
    rows_b = {bytes(r.cpu().numpy().tobytes()) for r in b["noise"].permute(1, 0, 2)}
    assert rows_a == rows_b
    assert torch.equal(a["topJ"], b["topJ"])                                 # the same 20 best trajectories
    np.testing.assert_allclose(a["plan"].cpu().numpy(), b["plan"].cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(torch.sort(a["w"]).values.cpu().numpy(), torch.sort(b["w"]).values.cpu().numpy(),
                               rtol=1e-4, atol=1e-9)
