"""Where the Panda rollout's wavefronts spend their clocks, per sample slot (profiling build only).

    tools/flag_variants.sh prof "-DM3_PABL_PROF"            (build container)
    M3P2I_HIP_LIB=$PWD/gpurun_variants/prof.so python tools/panda_wave_profile.py [panda_pick|panda] [--lps 16]   (GPU box)

The -DM3_PABL_PROF build overwrites rows 0-5 of cost_horizon with each sample's shader-clock totals (whole rollout, velocity
passes, the near path's detection + row build) and substep counts (with gripper rows, with body rows, near); this script runs
the bench scene of the config and prints the distribution over wavefronts -- a launch lasts as long as its SLOWEST wavefront."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "panda_pick"
    lps = int(sys.argv[sys.argv.index("--lps") + 1]) if "--lps" in sys.argv else 0
    device = "cuda:0"
    env, task, goal, mm, K, T = bench.CONFIGS[name]
    pl, sim, obj, cfg = bench.build_tamp(env, task, goal, mm, K, 0, 1, T, device)
    if name == "panda_pick":
        bench.make_pick_scene(bench.panda_pick_scene(device))(pl, sim, obj, cfg)
    if "--settled" in sys.argv:
        bench.settled_panda_scene(pl, sim, obj, cfg)
    state = sim._dof_state[0]
    for _ in range(5):
        pl.command(state)
    eng = pl._engine
    if lps:
        eng.set_panda_lanes_per_sample(lps)
    pl.command(state)
    torch.cuda.synchronize()
    ch = eng.cost_horizon.cpu().numpy()          # [K, T]
    tot, solve, near, n_rob, n_body, n_near = (ch[:, j] for j in range(6))
    detect, post, n_act, n_fk = (ch[:, j] for j in range(6, 10))
    pre, mid, wake, in_step = (ch[:, j] for j in range(10, 14))
    per_wave = (64 // lps if lps else 64) - (1 if task == "reach" else 0)
    nw = (K + per_wave - 1) // per_wave
    w_tot = np.array([tot[i * per_wave:(i + 1) * per_wave].max() for i in range(nw)])
    order = np.argsort(-w_tot)
    out = {"config": name, "K": K, "T": T, "samples_per_wave": per_wave, "waves": nw,
           "wave_clocks": {"min": float(w_tot.min()), "p50": float(np.percentile(w_tot, 50)), "p90": float(np.percentile(w_tot, 90)),
                           "p99": float(np.percentile(w_tot, 99)), "max": float(w_tot.max())},
           "mean_over_samples": {"total": float(tot.mean()), "solver": float(solve.mean()), "near_path": float(near.mean()),
                                 "substeps_with_gripper_rows": float(n_rob.mean()), "substeps_with_body_rows": float(n_body.mean()),
                                 "substeps_near": float(n_near.mean()), "manifold_detect_prepare": float(detect.mean()),
                                 "post_integration_kinematics_grasp": float(post.mean()), "substeps_with_an_awake_cube_in_the_wave": float(n_act.mean()),
                                 "substeps_with_post_kinematics": float(n_fk.mean()),
                                 "servo_and_lazy_tests": float(pre.mean()), "wake_gravity_manifold_init": float(wake.mean()),
                                 "forces_warm_sleep_integration": float(mid.mean()), "inside_panda_step": float(in_step.mean()),
                                 "outside_panda_step_assembly_cost_stores": float((tot - in_step).mean())},
           "slowest_waves": []}
    for wv in order[:3]:
        sl = slice(wv * per_wave, (wv + 1) * per_wave)
        out["slowest_waves"].append({"wave": int(wv), "clocks": float(w_tot[wv]), "solver": float(solve[sl].max()), "near_path": float(near[sl].max()),
                                     "gripper_row_substeps": n_rob[sl].tolist(), "body_row_substeps": n_body[sl].tolist()})
    fast = order[-1]
    sl = slice(fast * per_wave, (fast + 1) * per_wave)
    out["fastest_wave"] = {"wave": int(fast), "clocks": float(w_tot[fast]), "solver": float(solve[sl].max()), "near_path": float(near[sl].max()),
                           "gripper_row_substeps": n_rob[sl].tolist(), "body_row_substeps": n_body[sl].tolist()}
    # how the slowest waves' clocks correlate with what their samples do
    w_body = np.array([n_body[i * per_wave:(i + 1) * per_wave].max() for i in range(nw)])
    w_rob = np.array([n_rob[i * per_wave:(i + 1) * per_wave].max() for i in range(nw)])
    out["corr_clocks_vs_body_row_substeps"] = float(np.corrcoef(w_tot, w_body)[0, 1]) if w_body.std() > 0 else None
    out["corr_clocks_vs_gripper_row_substeps"] = float(np.corrcoef(w_tot, w_rob)[0, 1]) if w_rob.std() > 0 else None
    out["hist_body_row_substeps_max_per_wave"] = np.bincount(w_body.astype(int), minlength=41).tolist()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
