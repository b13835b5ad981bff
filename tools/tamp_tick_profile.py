"""GPU: where a closed-loop tick of the Panda pick-and-place goes on the HOST side -- the planner side of scripts/reactive_tamp.py:43-61
(Tamp.run_tamp of tools/closed_loop.py), piece by piece, with a device synchronisation after each piece so that every piece owns
its own GPU time as well.  The command() call is the product's hot path; the rest is what its callers do around it each tick.

    python tools/tamp_tick_profile.py [--ticks 60] [--json gpurun_out/tamp_tick_profile.json]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p2i_aip_amd import compat  # noqa: E402
import closed_loop as CL         # noqa: E402  (tools/ is on sys.path when run as a script)


def main(argv):
    ticks, out = 60, None
    it = iter(argv)
    for a in it:
        if a == "--ticks":
            ticks = int(next(it))
        elif a == "--json":
            out = next(it)
    compat.install(force_standins=True)
    import m3p2i_aip.utils.isaacgym_utils.isaacgym_wrapper as wrapper
    cfg = compat.make_config("config_panda", ["mppi.num_samples=4000", "mppi.horizon=20"])
    tamp = CL.Tamp(cfg)
    real = wrapper.IsaacGymWrapper(cfg.isaacgym, cfg.env_type, num_envs=1, viewer=False, device=cfg.mppi.device,
                                   cube_on_shelf=cfg.cube_on_shelf)
    rows = []
    names = ["set_state", "update_plan", "update_gripper_command", "update_objective", "pull_preference", "check_task_success",
             "command", "action_to_host", "real_step"]

    def timed(row, name, f):
        t0 = time.perf_counter()
        r = f()
        torch.cuda.synchronize()
        row[name] = (time.perf_counter() - t0) * 1e6
        return r

    for i in range(ticks):
        torch.cuda.synchronize()
        row = {}
        s = tamp.sim

        def set_state():
            s._dof_state[:] = real._dof_state
            s._root_state[:] = real._root_state
            s.set_dof_state_tensor(s._dof_state)
            s.set_actor_root_state_tensor(s._root_state)
        timed(row, "set_state", set_state)
        timed(row, "update_plan", lambda: tamp.task_planner.update_plan(s))
        timed(row, "update_gripper_command", lambda: tamp.motion_planner.update_gripper_command(tamp.task_planner.task))
        timed(row, "update_objective", lambda: tamp.objective.update_objective(tamp.task_planner.task, tamp.task_planner.curr_goal))
        timed(row, "pull_preference", lambda: tamp.motion_planner.get_pull_preference())
        done = timed(row, "check_task_success", lambda: bool(tamp.task_planner.check_task_success(s)))
        if done:
            break
        a = timed(row, "command", lambda: tamp.motion_planner.command(s._dof_state[0])[0])
        timed(row, "action_to_host", lambda: a.cpu())

        def real_step():
            real.set_dof_velocity_target_tensor(a.view(1, -1))
            real.step()
        timed(row, "real_step", real_step)
        row["task"] = tamp.task_planner.task
        rows.append(row)
    rep = {}
    for task in sorted({r["task"] for r in rows}):
        sel = [r for r in rows if r["task"] == task]
        rep[task] = dict(ticks=len(sel), us_p50={n: round(float(np.median([r[n] for r in sel])), 1) for n in names})
        rep[task]["us_p50"]["total"] = round(sum(rep[task]["us_p50"].values()), 1)
        print(task, rep[task], flush=True)
    if out:
        json.dump(rep, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
