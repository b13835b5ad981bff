"""GPU experiment (needs a -DM3_ABL_COUNT build loaded through M3P2I_HIP_LIB): histogram of the wave-uniform
pair-group masks the substep dispatcher sees while the planner runs in CLOSED loop (1-env world stepped with
the plan's first action, as bench.py's closed_loop) -- the scene evolves towards the goal, where other
contact groups dominate than in the initial scene (box against the walls of the corner goal).

    tools/flag_variants.sh count "-DM3_ABL_COUNT"
    M3P2I_HIP_LIB=gpurun_variants/count.so python tools/mask_count_closed_loop.py push hybrid [ticks]
"""
import ctypes, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from m3p2i_aip_amd import isaacgym_wrapper as wrapper
from m3p2i_aip_amd.compat import check_and_apply_suction

NAMES = ["RB", "RD", "RO", "RW", "BW", "DW", "BD", "BO", "DO"]
args = [a for a in sys.argv[1:] if not a.isdigit()] or ["push"]
ticks = next((int(a) for a in sys.argv[1:] if a.isdigit()), 400)
GOALS = {"pushcorner": ("push", (-3.75, -3.75))}
out = {}
for name in args:
    base = "push" if name in GOALS else name
    env, task, goal, mm, K, T = bench.CONFIGS[base]
    if name in GOALS:
        task, goal = GOALS[name]
    pl, sim, obj, cfg = bench.build_tamp(env, task, goal, mm, K, 0, 1, T, "cuda:0")
    real = wrapper.IsaacGymWrapper(cfg.isaacgym, cfg.env_type, num_envs=1, device="cuda:0")
    pl.attach(sim=real)
    lib = ctypes.CDLL(os.environ["M3P2I_HIP_LIB"])
    buf = (ctypes.c_uint * 512)()
    pull = task in ("pull", "push_pull")
    goal_t = torch.tensor(goal, device="cuda:0")
    hist_by_phase = []
    for seg in range(4):
        lib.m3_dbg_levels(buf, 1)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(seg * ticks // 4, (seg + 1) * ticks // 4):
            real.update_dyn_obs(i)
            a = pl.command(real._dof_state[0])[0]
            real.set_dof_velocity_target_tensor(a.view(1, 2))
            if pull:
                cfg.suction_active = pl.pull_preference_tensor()
                check_and_apply_suction(cfg, real, a.view(1, 2))
            real.step()
        e1.record()
        torch.cuda.synchronize()
        lib.m3_dbg_levels(buf, 0)
        h = np.frombuffer(buf, dtype=np.uint32).astype(np.float64)
        h /= max(h.sum(), 1)
        err = float(torch.norm(real.get_actor_position_by_name("box")[0, :2] - goal_t))
        top = {("|".join(n for b, n in enumerate(NAMES) if m >> b & 1) or "-"): round(100 * h[m], 1) for m in np.argsort(-h)[:8] if h[m] > 0.004}
        hist_by_phase.append({"ticks": f"{seg * ticks // 4}..{(seg + 1) * ticks // 4}", "ms_per_tick": e0.elapsed_time(e1) / (ticks // 4),
                              "box_to_goal_m": round(err, 3), "mask_pct": top})
        print(name, hist_by_phase[-1])
    out[name] = hist_by_phase
    pl._engine.close()
p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
if os.path.isdir(p):
    json.dump(out, open(os.path.join(p, "mask_count_closed_loop.json"), "w"), indent=1)
