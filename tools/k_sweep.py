"""GPU experiment: where does the rollout saturate?  Push task, T = 30, in-kernel noise
(sampling_random: no host sampler, 36 B algorithmic traffic per state-step), K = 2k .. 1M.
Prints one JSON line per K and writes gpurun_out/k_sweep.json.
    python tools/k_sweep.py [K,K,...] [push|panda]      panda: reach, T = 20, nu = 9, 92 B per state-step -> k_sweep_panda.json"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p2i_aip_amd.engine import HipEngine, make_config

PANDA = len(sys.argv) > 2 and sys.argv[2] == "panda"
T = 20 if PANDA else 30
BYTES = 92.0 if PANDA else 36.0
Ks = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else "2000,8000,32000,65536,131072,262144,524288,1048576").split(",")]
out = []
for K in Ks:
    if PANDA:
        eng = HipEngine(make_config(K=K, T=T, nu=9, env_type="panda_env", sampling_random=True, u_min=[-2.0] * 7 + [-1.5] * 2,
                                    u_max=[2.0] * 7 + [1.5] * 2, noise_sigma_diag=[10.0] * 7 + [0.8] * 2, lambda_=0.05, dt=0.01, seed=1))
        eng.set_objective("reach", (0.2, 0.2, 1.115, 0.0, 0.0, 0.0, 1.0), gripper_cmd=1)
    else:
        eng = HipEngine(make_config(K=K, T=T, nu=2, sampling_random=True, u_min=[-3, -3], u_max=[3, 3],
                                    noise_sigma_diag=[3, 3], seed=1))
        eng.set_objective("push", (-1.0, -1.0))
    eng.enable_timing(True)
    for _ in range(5):
        eng.command()
    ts = []
    for _ in range(20):
        eng.command()
        t = eng.timing()
        ts.append((t.rollout_ms, t.update_ms, t.finalize_ms, t.total_ms))
    m = np.mean(ts, axis=0)
    rec = dict(K=K, T=T, rollout_ms=float(m[0]), update_ms=float(m[1]), finalize_ms=float(m[2]), total_ms=float(m[3]),
               state_steps_per_s=K * T / (m[3] * 1e-3), rollout_alg_GBps=BYTES * K * T / (m[0] * 1e-3) / 1e9,
               waves=(K + 63) // 64)
    out.append(rec)
    print(json.dumps(rec), flush=True)
    eng.close()
json.dump(out, open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "k_sweep_panda.json" if PANDA else "k_sweep.json"), "w"), indent=1)
