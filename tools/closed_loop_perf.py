"""GPU: ms per tick of the closed loop (bench.py's closed_loop, continued past success) for push / push to the
corner goal / hybrid, in four segments of the episode -- the scene at the goal (box against the walls for
the corner goal of config_point.yaml) exercises other substep instances than the initial scene.
    python tools/closed_loop_perf.py [ticks=400]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from m3p2i_aip_amd import isaacgym_wrapper as wrapper
from m3p2i_aip_amd.compat import check_and_apply_suction

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 400
CASES = {"push": ("push", "push", (-1.0, -1.0)), "pushcorner": ("push", "push", (-3.75, -3.75)), "hybrid": ("hybrid", None, None)}
out = {}
for name, (base, task_o, goal_o) in CASES.items():
    env, task, goal, mm, K, T = bench.CONFIGS[base]
    task, goal = task_o or task, goal_o or goal
    pl, sim, obj, cfg = bench.build_tamp(env, task, goal, mm, K, 0, 1, T, "cuda:0")
    real = wrapper.IsaacGymWrapper(cfg.isaacgym, cfg.env_type, num_envs=1, device="cuda:0")
    real.zero_copy_targets = True
    pl.attach(sim=real)
    pull = task in ("pull", "push_pull")
    goal_t = torch.tensor(goal, device="cuda:0")
    segs = []
    for seg in range(4):
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
        segs.append({"ticks": f"{seg * ticks // 4}..{(seg + 1) * ticks // 4}", "ms_per_tick": round(e0.elapsed_time(e1) / (ticks // 4), 4),
                     "box_to_goal_m": round(float(torch.norm(real.get_actor_position_by_name("box")[0, :2] - goal_t)), 3)})
    out[name] = segs
    print(name, segs)
    pl._engine.close()
p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
if os.path.isdir(p):
    json.dump(out, open(os.path.join(p, "closed_loop_perf.json"), "w"), indent=1)
