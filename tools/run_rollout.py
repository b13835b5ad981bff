"""Driver for profiling: N commands at (K, lanes).  usage: run_rollout.py K lanes [task] [n]"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p2i_aip_amd.engine import HipEngine, make_config
from m3p2i_aip_amd import sampling
K, lanes = int(sys.argv[1]), int(sys.argv[2])
task = sys.argv[3] if len(sys.argv) > 3 else "push"
n = int(sys.argv[4]) if len(sys.argv) > 4 else 20
panda = task in ("reach", "pick", "place")
T = 20 if panda else 30
mm = task == "push_pull"
if panda:   # config_panda: K x T=20, nu=9 (usage: run_rollout.py 4000 0 reach)
    delta = sampling.halton_knots(K, T, 9)
    eng = HipEngine(make_config(K=K, T=T, nu=9, env_type="panda_env", u_min=[-2.0] * 7 + [-1.5] * 2,
                                u_max=[2.0] * 7 + [1.5] * 2, noise_sigma_diag=[10.0] * 7 + [0.8] * 2,
                                lambda_=0.05, dt=0.01))
    eng.set_objective(task, (0.2, 0.2, 1.115, 0.0, 0.0, 0.0, 1.0))
else:
    delta = sampling.halton_knots(K, T, 2)
    eng = HipEngine(make_config(K=K, T=T, nu=2, multi_modal=mm, u_min=[-3, -3], u_max=[3, 3], noise_sigma_diag=[3, 3]))
    eng.set_objective(task, (-1.0, -1.0))
eng.set_noise_knots(delta)
eng.set_rollout_lanes(lanes)
for _ in range(n):
    eng.command()
torch.cuda.synchronize()
