"""Device-side Halton-spline sampler (m3_set_noise_knots) against the
host sampler that calls scipy's FITPACK exactly like the reference (sampling.halton_spline_delta,
pinned by golden G8): the noise must be bit-identical, for both environments' shapes and for a
shard of the global sample set."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("K,T,nu,k0,k1", [(2000, 30, 2, 0, 2000), (4000, 20, 9, 0, 4000), (1000, 12, 2, 0, 1000),
                                          (4096, 30, 2, 1024, 2048), (300, 64, 2, 0, 300)])
def test_device_sampler_equals_host_sampler(K, T, nu, k0, k1):
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd import sampling
    from m3p2i_aip_amd.engine import HipEngine, make_config
    env = "point_env" if nu == 2 else "panda_env"
    kw = dict(u_min=[-1.0] * nu, u_max=[1.0] * nu, noise_sigma_diag=[1.0] * nu)
    eng = HipEngine(make_config(K=K, K_local=k1 - k0, k_offset=k0, T=T, nu=nu, env_type=env, filter_u=T >= 9, **kw))
    eng.set_noise_knots(sampling.halton_knots(K, T, nu, k0=k0, k1=k1))
    torch.cuda.synchronize()
    dev = eng.buffer(L.BUF_NOISE).permute(1, 0, 2).cpu().numpy()          # [K_local, T, nu]
    ref = sampling.halton_spline_delta(K, T, nu, k0=k0, k1=k1, workers=1).numpy()
    np.testing.assert_array_equal(dev, ref)
    eng.close()


def test_planner_uses_the_device_sampler_and_it_is_fast():
    """K = 64000 x nu = 2 = 128000 fits: ~4 s through scipy on the host, milliseconds here."""
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd import sampling
    from m3p2i_aip_amd.engine import HipEngine, make_config
    K, T, nu = 64000, 30, 2
    eng = HipEngine(make_config(K=K, T=T, nu=nu, u_min=[-3, -3], u_max=[3, 3], noise_sigma_diag=[3, 3]))
    knots = sampling.halton_knots(K, T, nu)
    eng.set_noise_knots(knots)            # warm-up (first launch allocates scratch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.set_noise_knots(knots)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"device sampler: {K * nu} spline fits in {dt * 1e3:.1f} ms (upload included)")
    assert dt < 0.5
    # spot-check 500 rows against scipy
    ref = sampling.halton_spline_delta(K, T, nu, k0=31000, k1=31500, workers=1).numpy()
    dev = eng.buffer(L.BUF_NOISE).permute(1, 0, 2)[31000:31500].cpu().numpy()
    np.testing.assert_array_equal(dev, ref)
    eng.close()


def test_bad_knot_shapes_are_refused():
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd.engine import HipEngine, make_config
    eng = HipEngine(make_config(K=64, T=12, nu=2, u_min=[-1, -1], u_max=[1, 1], noise_sigma_diag=[1, 1]))
    with pytest.raises(L.M3Error):
        eng.set_noise_knots(np.zeros((64, 2, 2), np.float32))       # n_knots <= degree ("m > k must hold")
    with pytest.raises(L.M3Error):
        eng.set_noise_knots(np.zeros((64, 2, 65), np.float32))
    eng.close()


@pytest.mark.parametrize("scramble", ["none", "faure"])
@pytest.mark.parametrize("K,T,nu,k0,k1", [(2000, 30, 2, 0, 2000), (4000, 20, 9, 0, 4000), (4096, 30, 2, 1024, 2048)])
def test_whole_sampler_on_the_device_agrees_with_the_host_sampler(K, T, nu, k0, k1, scramble):
    """m3_set_noise_halton: Halton radical inverses + erfinv on the device as well (MPPIConfig.device_knots).
    The Halton uniforms are the host's bit for bit (the knot signs and magnitudes below would scatter
    otherwise); the Gaussian values go through the device's erff / expf / logf, so the final noise agrees with
    the host sampler -- the reference's arithmetic, golden G8 -- to ~1e-5 absolute, not bit for bit."""
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd import sampling
    from m3p2i_aip_amd.engine import HipEngine, make_config
    env = "point_env" if nu == 2 else "panda_env"
    kw = dict(u_min=[-1.0] * nu, u_max=[1.0] * nu, noise_sigma_diag=[1.0] * nu)
    eng = HipEngine(make_config(K=K, K_local=k1 - k0, k_offset=k0, T=T, nu=nu, env_type=env, **kw))
    eng.set_noise_halton(T // 4, scramble=scramble)      # "faure": generalized Halton, m3_set_noise_halton_scrambled
    torch.cuda.synchronize()
    dev = eng.buffer(L.BUF_NOISE).permute(1, 0, 2).cpu().numpy()
    ref = sampling.halton_spline_delta(K, T, nu, k0=k0, k1=k1, workers=1, scramble=scramble).numpy()
    assert np.isfinite(dev).all()
    np.testing.assert_allclose(dev, ref, atol=3e-5, rtol=1e-5)
    eng.close()


def test_planner_with_device_knots_plans_like_the_default_sampler():
    import bench
    outs = []
    for dk in (False, True):
        pl, sim, obj, cfg = bench.build_tamp("point_env", "push", (-1.0, -1.0), False, 2000, 0, 1, 30, "cuda:0")
        cfg.mppi.device_knots = dk
        a = [pl.command(sim._dof_state[0]).clone() for _ in range(3)][-1]
        outs.append(a.cpu().numpy())
        pl._engine.close()
    np.testing.assert_allclose(outs[1], outs[0], atol=2e-3)


@pytest.mark.parametrize("device_knots", [False, True])
def test_planner_with_scrambled_halton(device_knots):
    """MPPIConfig.halton_scramble='faure' (host knots and device knots): a different -- better spread -- sample set of
    the same distribution: the planner runs on it, the noise has the default's moments, the plan is of the same kind."""
    import bench
    stats = {}
    for sc in ("none", "faure"):
        pl, sim, obj, cfg = bench.build_tamp("panda_env", "reach", (0.0,) * 7, False, 4000, 0, 1, 20, "cuda:0")
        cfg.mppi.halton_scramble, cfg.mppi.device_knots = sc, device_knots
        a = [pl.command(sim._dof_state[0]).clone() for _ in range(3)][-1]
        d = pl.delta.contiguous().cpu().numpy()
        stats[sc] = (d.mean(), d.std(), a.cpu().numpy(), float(pl.cost_total.min()))
        assert np.isfinite(a.cpu().numpy()).all()
        pl._engine.close()
    assert abs(stats["faure"][0]) < 0.02 and abs(stats["faure"][1] - stats["none"][1]) < 0.05 * stats["none"][1]
    assert not np.array_equal(stats["faure"][2], stats["none"][2])
    assert stats["faure"][3] < 1.5 * stats["none"][3] + 1.0          # best rollout cost of the same order
