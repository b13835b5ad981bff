"""Error behaviour of the C-ABI (include/m3p2i_hip.h): a caller that binds `libm3p2i_hip.so` from another language sees
return codes, not Python exceptions -- bad configurations are refused by m3_create with the documented code and a
message, null handles never crash, calls that do not fit the handle's sharding are M3_ERR_STATE, and a refused call
leaves the handle usable."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

BAD_ARG, HIP, SHAPE, STATE, UNSUPPORTED = -1, -2, -3, -4, -5
BASE = dict(K=256, T=30, nu=2, u_min=[-3, -3], u_max=[3, 3], noise_sigma_diag=[3, 3])


def try_create(**kw):
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd.engine import make_config
    tweak = kw.pop("tweak", None)
    cfg = make_config(**{**BASE, **kw})
    if tweak:
        tweak(cfg)
    lib = L.load()
    torch.zeros(1, device="cuda:0")
    h = C.c_void_p()
    rc = lib.m3_create(C.byref(cfg), C.byref(h))
    msg = lib.m3_last_error(None)
    if rc == 0:
        lib.m3_destroy(h)
    return rc, (msg.decode() if msg else ""), bool(h)


@pytest.mark.parametrize("kw,code,word", [
    (dict(K=0), SHAPE, "K_global"),
    (dict(K=256, K_local=300), SHAPE, "K_global"),
    (dict(K=256, K_local=128, k_offset=192), SHAPE, "k_offset"),
    (dict(K=10), SHAPE, ">= 20"),                                     # torch.topk(weights, 20), mppi.py:248
    (dict(T=0), SHAPE, "horizon"),
    (dict(T=5), SHAPE, "savgol"),                                     # filter_u needs the 9-row window, mppi.py:190
    (dict(nu=3, u_min=[-3] * 3, u_max=[3] * 3, noise_sigma_diag=[3] * 3), SHAPE, "nu == 2"),
    (dict(env_type="panda_env", nu=2), SHAPE, "nu == 9"),
    (dict(noise_sigma_diag=[3, 0]), BAD_ARG, "noise_sigma"),
    (dict(u_min=[3, 3], u_max=[-3, -3]), BAD_ARG, "bounds"),
    (dict(gamma=0.0), BAD_ARG, "gamma"),
    (dict(substeps=0), BAD_ARG, "substeps"),
    (dict(noise_sigma=[[3.0, 4.0], [4.0, 3.0]]), BAD_ARG, "positive definite"),
    (dict(tweak=lambda c: setattr(c, "abi_version", 999)), BAD_ARG, "abi_version"),
    (dict(tweak=lambda c: setattr(c, "env_type", 7)), BAD_ARG, "env_type"),
    # sharding protocols
    (dict(K=512, K_local=256, shard_mix=2), BAD_ARG, "multi-modal"),                          # ladder tables: M3P2I only
    (dict(K=512, K_local=256, shard_mix=1, multi_modal=True, sampling_random=True), UNSUPPORTED, "noise TABLE"),
    (dict(K=90, K_local=30, k_offset=30, shard_mix=1), SHAPE, None),                           # fine: 3 equal shards >= 20
    (dict(K=100, K_local=30, k_offset=30, shard_mix=1), SHAPE, "equal shards"),
    (dict(K=60, K_local=10, k_offset=10, shard_mix=1), SHAPE, "K_local >= 20"),
    (dict(K=66, K_local=33, k_offset=33, shard_mix=1, multi_modal=True), SHAPE, "even K_local"),
    (dict(K=512, K_local=256, update_cov=True), UNSUPPORTED, "update_cov"),
])
def test_m3_create_refuses_bad_configurations(kw, code, word):
    rc, msg, handle = try_create(**kw)
    if word is None:                     # (the control case: a legal sharding is accepted)
        assert rc == 0, msg
        return
    assert rc == code, (rc, msg)
    assert word in msg, msg
    assert not handle                    # no half-built handle is handed out


def test_null_handles_and_null_arguments_never_crash():
    from m3p2i_aip_amd import _lib as L
    lib = L.load()
    null = C.c_void_p()
    for name in ("m3_rollout", "m3_update", "m3_finalize", "m3_update_finalize", "m3_reset", "m3_p2p_put", "m3_p2p_wait",
                 "m3_p2p_exchange", "m3_update_b", "m3_relabel_samples"):
        assert getattr(lib, name)(null) == BAD_ARG, name
    assert lib.m3_command(null, None) == BAD_ARG
    assert lib.m3_create(None, None) == BAD_ARG
    assert lib.m3_p2p_set_timeout_ms(null, 1, 1) == BAD_ARG
    assert lib.m3_get_info(null, None) == BAD_ARG
    lib.m3_destroy(null)                 # a no-op, as free(NULL)


def test_calls_that_do_not_fit_the_handle_are_refused_and_leave_it_usable(oracle):
    from m3p2i_aip_amd import _lib as L
    from m3p2i_aip_amd.engine import HipEngine, make_config
    rng = np.random.default_rng(0)
    delta = rng.normal(0, 1, (256, 30, 2)).astype(np.float32)
    # a sharded handle has no one-call command and no fused update (the exchange goes in between)
    sh = HipEngine(make_config(**{**BASE, "K": 512, "K_local": 256, "shard_mix": 1}))
    sh.set_noise(delta)
    for call in (sh.command, sh.update_finalize, sh.p2p_put, lambda: sh.p2p_wait(0), lambda: sh.p2p_put(1)):
        with pytest.raises(L.M3Error):
            call()
    with pytest.raises(L.M3Error, match="shard_mix = 3"):
        sh.update_b()                    # (the middle phase of the two-exchange protocol only)
    sh.rollout()
    sh.update()                          # still works after the refusals
    torch.cuda.synchronize()
    assert np.isfinite(sh.buffer(L.BUF_TRAJ_COST).cpu().numpy()).all()
    sh.close()
    sh.close()                           # idempotent
    # an unsharded handle: the p2p entry points need a sharding; the command still runs afterwards
    e = HipEngine(make_config(**BASE))
    e.set_noise(delta)
    with pytest.raises(L.M3Error):
        e.p2p_exchange()
    with pytest.raises(L.M3Error):
        e.set_noise_knots(np.zeros((256, 2, 2), np.float32))      # two knots for a degree-2 spline (splrep: m > k)
    a = e.command(sync_host=True)
    assert a.shape == (30, 2) and np.isfinite(a).all()
    e.close()
