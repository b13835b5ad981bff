"""GPU experiment (needs a -DM3_ABL_COUNT build loaded through M3P2I_HIP_LIB; the per-instance timings
additionally -DM3_ABL_COUNT_CYCLES, which since the pass versions of point_substep no longer gives correct
rollouts -- see planar_dyn.hpp -- so only the histogram part is trustworthy today): histogram of the
wave-uniform pair-group masks the substep dispatcher sees, over commands 5..60 of a bench config."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

NAMES = ["RB", "RD", "RO", "RW", "BW", "DW", "BD", "BO", "DO"]
for name in sys.argv[1:] or ["push"]:
    env, task, goal, mm, K, T = bench.CONFIGS[name]
    pl, sim, obj, _cfg = bench.build_tamp(env, task, goal, mm, K, 0, 1, T, "cuda:0")
    state = sim._dof_state[0]
    lib = ctypes.CDLL(os.environ["M3P2I_HIP_LIB"])
    buf = (ctypes.c_uint * 512)()
    for it in range(5):
        pl.command(state)
    torch.cuda.synchronize()
    lib.m3_dbg_levels(buf, 1)
    cyc = (ctypes.c_uint * 1024)()
    lib.m3_dbg_cycles(cyc, 1)
    for it in range(200):
        pl.command(state)
    torch.cuda.synchronize()
    lib.m3_dbg_levels(buf, 0)
    a = np.frombuffer(buf, dtype=np.uint32).astype(np.float64)
    a /= a.sum()
    print(name)
    for m in np.argsort(-a)[:16]:
        if a[m] > 0:
            print("  %5.1f %%  %s" % (100 * a[m], "|".join(n for i, n in enumerate(NAMES) if m >> i & 1) or "-"))
    lib.m3_dbg_cycles(cyc, 0)
    c = np.frombuffer(cyc, dtype=np.uint32).astype(np.float64).reshape(64, 16) / 200.0
    c = c[c[:, 8:].sum(1) > 0]
    tot = c[:, :6].sum(1)
    wv = int(np.argmax(tot))
    print("  per command, instance classes [none, RB, RB|RD, +BD, no-box-statics, all]; 100 MHz ticks")
    print("  mean wave : n", c[:, 8:14].mean(0).round(1), "us", (c[:, :6].mean(0) / 100).round(1), "sum %.1f us" % (tot.mean() / 100))
    print("  worst wave: n", c[wv, 8:14].round(1), "us", (c[wv, :6] / 100).round(1), "sum %.1f us" % (tot[wv] / 100))
    print("  us per substep (mean over waves):", (c[:, :6].sum(0) / np.maximum(c[:, 8:14].sum(0), 1e-9) / 100).round(2))
