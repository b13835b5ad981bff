"""GPU experiment: rollout kernel time vs lanes-per-wavefront and K (results in DESIGN.md)."""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p2i_aip_amd.engine import HipEngine, make_config
from m3p2i_aip_amd import sampling

T = 30
base = sampling.halton_knots(2000, T, 2)
out = []
for K in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2000,10000").split(",")]:
    delta = sampling.halton_knots(K, T, 2)
    eng = HipEngine(make_config(K=K, T=T, nu=2, u_min=[-3, -3], u_max=[3, 3], noise_sigma_diag=[3, 3]))
    eng.set_objective("push", (-1.0, -1.0))
    eng.set_noise_knots(delta)
    eng.enable_timing(True)
    for lanes in ((0,) if os.environ.get("M3P2I_HIP_LIB") else (0, 1, 2, 4, 8, 16, 32, 64)):
        eng.set_rollout_lanes(lanes)
        eng.reset()
        for _ in range(10):
            eng.command()
        ts = []
        for _ in range(30):
            eng.command()
            t = eng.timing()
            ts.append((t.rollout_ms, t.update_ms, t.finalize_ms, t.total_ms))
        m = np.mean(ts, axis=0)
        out.append(dict(K=K, lanes=lanes, rollout_ms=float(m[0]), update_ms=float(m[1]), finalize_ms=float(m[2]), total_ms=float(m[3])))
        print(out[-1], flush=True)
    eng.close()
json.dump(out, open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "lanes_sweep.json"), "w"))
