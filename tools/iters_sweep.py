"""GPU experiment: rollout kernel time vs solver passes / substeps (where does the time go)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p2i_aip_amd.engine import HipEngine, make_config
from m3p2i_aip_amd import sampling
K, T = 2000, 30
delta = sampling.halton_knots(K, T, 2)
for iters, sub in ((1, 2), (2, 2), (4, 2), (6, 2), (6, 1)):
    eng = HipEngine(make_config(K=K, T=T, nu=2, u_min=[-3, -3], u_max=[3, 3], noise_sigma_diag=[3, 3],
                                solver_iters=iters, substeps=sub))
    eng.set_objective("push", (-1.0, -1.0))
    eng.set_noise_knots(delta)
    eng.enable_timing(True)
    for _ in range(10):
        eng.command()
    ts = []
    for _ in range(30):
        eng.command()
        ts.append(eng.timing().rollout_ms)
    print(f"iters={iters} substeps={sub}: rollout {np.mean(ts)*1e3:.1f} us (min {np.min(ts)*1e3:.1f})", flush=True)
    eng.close()
