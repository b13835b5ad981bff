"""GPU: quirk Q8 two ways on random worlds -- the reach command with the cost kernel behind a rollout without shadow slots (sixteen /
eight lanes per sample) against the shadow-slot rollout (one / eight lanes), single and multi-modal, two warm-started commands each:
every buffer of the command must be equal bit for bit.
    python tools/fuzz_reach_cost_paths.py [n_worlds=120]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle, oracle.panda as P
from tests import test_hip_parity_panda as tq
from m3p2i_aip_amd import _lib as L
from m3p2i_aip_amd.engine import HipEngine, make_config
oracle.build()
sc = P.default_scene()
bad = 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
for seed in range(n):
    rng = np.random.default_rng(7000 + seed)
    K = int(rng.choice([64, 90, 200, 257])); T = 20
    mm = bool(seed % 2)
    w0 = tq._fuzz_batch(seed // 42)[seed % 42].copy().astype(np.float32)
    goal = np.array([0.2, 0.2, 1.115, 0, 0, 0, 1], np.float32)
    delta = rng.standard_normal((K, T, 9)).astype(np.float32)
    outs = []
    for deferred, lps in ((True, 16), (True, 8), (False, 1), (False, 8)):
        eng = HipEngine(make_config(K=K, T=T, nu=9, env_type="panda_env", multi_modal=mm, u_min=tq.UMIN, u_max=tq.UMAX,
                                    noise_sigma_diag=tq.SIG, lambda_=0.05, pre_height_diff=0.05, dt=0.01))
        eng.set_objective("reach", goal, gripper_cmd=1 + seed % 2)
        eng.set_panda_lanes_per_sample(lps); eng.set_panda_reach_cost_kernel(deferred)
        eng.set_noise(delta); eng.set_world_panda_raw(tq.raw31(P, w0))
        for _ in range(2):
            eng.command(sync_host=True)
        outs.append([eng.buffer(b).clone() for b in (L.BUF_TRAJ_COST, L.BUF_COST_HORIZON, L.BUF_STATES, L.BUF_MEAN, L.BUF_ACTION_OUT)])
        eng.close()
    for o in outs[1:]:
        if not all(torch.equal(a, b) for a, b in zip(outs[0], o)):
            bad += 1; print("MISMATCH seed", seed, "K", K, "mm", mm, flush=True); break
print(f"reach, cost kernel vs shadow slots, single / multi-modal, {n} random worlds x 4 forms x 2 commands: {bad} mismatches")
