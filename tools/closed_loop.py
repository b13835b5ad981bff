"""Closed-loop reactive TAMP in one process (GPU): the flow of the reference's scripts/sim.py:19-58
(1-env "real world") and scripts/reactive_tamp.py:22-88 (planner side) without the RPC hop.

    python tools/closed_loop.py [-cn config_point|config_panda] [key=value ...] [--ticks N] [--json out]
    python tools/closed_loop.py --serve tcp://127.0.0.1:4242 [...]      planner process (reactive_tamp.py's role)
    python tools/closed_loop.py --connect tcp://127.0.0.1:4242 [...]    world process (sim.py's role), same overrides
    (--torch-blobs on either side: its tensors through torch.save / torch.load instead of m3p2i_aip_amd/blobs.py's patched archives)

e.g.  python tools/closed_loop.py task=push goal=[-1,-1] mppi.num_samples=2000 mppi.horizon=30
      python tools/closed_loop.py task=push_pull multi_modal=True mppi.num_samples=4000 mppi.horizon=30
      python tools/closed_loop.py -cn config_panda mppi.num_samples=4000 mppi.horizon=20

Reports what the reference logs per run: task timeline, ticks and simulated
time to success, final position error, and the per-tick command() wall time (p50 / p99).
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from m3p2i_aip_amd import compat  # noqa: E402


class Tamp:
    """Planner side: scripts/reactive_tamp.py:22-88 (REACTIVE_TAMP) written against the reference's module
    names.  `run_tamp` takes / returns tensors; the *_bytes methods are the RPC surface of the script
    (torch.save blobs, utils/data_transfer.py)."""

    def __init__(self, cfg):
        from m3p2i_aip.planners.motion_planner import m3p2i
        from m3p2i_aip.planners.task_planner import task_planner
        import m3p2i_aip.utils.isaacgym_utils.isaacgym_wrapper as wrapper
        from m3p2i_aip.planners.motion_planner.cost_functions import Objective
        self.cfg = cfg
        self.sim = wrapper.IsaacGymWrapper(cfg.isaacgym, cfg.env_type, num_envs=cfg.mppi.num_samples,
                                           viewer=False, device=cfg.mppi.device, cube_on_shelf=cfg.cube_on_shelf)
        self.objective = Objective(cfg)
        self.task_planner = task_planner.set_task_planner(cfg)
        self.task_success = False
        self.suction_active = False
        self.motion_planner = m3p2i.M3P2I(cfg, dynamics=self.dynamics, running_cost=self.running_cost)

    def dynamics(self, _, u, t=None):
        self.sim.set_dof_velocity_target_tensor(u)
        self.sim.step()
        return torch.stack([self.sim.robot_pos[:, 0], self.sim.robot_vel[:, 0],
                            self.sim.robot_pos[:, 1], self.sim.robot_vel[:, 1]], dim=1), u

    def running_cost(self, _):
        return self.objective.compute_cost(self.sim)

    def run_tamp(self, dof_state, root_state):
        self.sim._dof_state[:] = dof_state
        self.sim._root_state[:] = root_state
        self.sim.set_dof_state_tensor(self.sim._dof_state)
        self.sim.set_actor_root_state_tensor(self.sim._root_state)
        self.task_planner.update_plan(self.sim)
        self.motion_planner.update_gripper_command(self.task_planner.task)
        self.objective.update_objective(self.task_planner.task, self.task_planner.curr_goal)
        self.suction_active = self.motion_planner.get_pull_preference()
        self.task_success = bool(self.task_planner.check_task_success(self.sim))
        if self.task_success:
            return torch.zeros(self.sim.dofs_per_robot, device=self.cfg.mppi.device)
        return self.motion_planner.command(self.sim._dof_state[0])[0]

    # ---- what reactive_tamp.py serves over zerorpc (:43-61, 83-87) + a status call for this tool ----
    def run_tamp_bytes(self, dof_bytes, root_bytes):
        from m3p2i_aip.utils.data_transfer import bytes_to_torch, torch_to_bytes
        dev = self.cfg.mppi.device
        return torch_to_bytes(self.run_tamp(bytes_to_torch(dof_bytes).to(dev), bytes_to_torch(root_bytes).to(dev)))

    def get_suction(self):
        from m3p2i_aip.utils.data_transfer import torch_to_bytes
        return torch_to_bytes(self.suction_active)

    def get_trajs(self):
        from m3p2i_aip.utils.data_transfer import torch_to_bytes
        return torch_to_bytes(self.motion_planner.top_trajs)

    def status(self):
        g = self.task_planner.curr_goal
        from m3p2i_aip_amd.cost_functions import attached_host_values
        host = attached_host_values(g)      # (a goal the task planner made from host values carries them: no read-back)
        return {"task": self.task_planner.task, "success": bool(self.task_success),
                "goal": list(host) if host is not None else [float(x) for x in g.float().cpu().reshape(-1).tolist()]}

    def close(self):
        self.sim.stop_sim()
        self.motion_planner._engine.close()


class RemoteTamp:
    """World side's view of a Tamp served in another process (scripts/sim.py:29-49): same three calls."""

    def __init__(self, endpoint, device):
        import zerorpc   # the real package, or compat's stand-in (m3p2i_aip_amd/rpc.py)
        self.c = zerorpc.Client(timeout=120)
        self.c.connect(endpoint)
        self.device = device
        self.task_success, self.suction_active, self.task = False, False, None

    def run_tamp(self, dof_state, root_state):
        from m3p2i_aip.utils.data_transfer import bytes_to_torch, torch_to_bytes
        a = bytes_to_torch(self.c.run_tamp_bytes(torch_to_bytes(dof_state), torch_to_bytes(root_state))).to(self.device)
        self.suction_active = bytes_to_torch(self.c.get_suction())
        st = self.c.status()
        self.task, self.task_success, self.goal = st["task"], st["success"], st["goal"]
        return a

    def close(self):
        self.c.close()


def serve(cn, overrides, endpoint):
    """Planner process: python tools/closed_loop.py --serve tcp://127.0.0.1:4242 [config overrides]."""
    compat.install(force_standins=True)
    import zerorpc
    tamp = Tamp(compat.make_config(cn, list(overrides)))
    server = zerorpc.Server(tamp)
    server.bind(endpoint)
    print("serving", endpoint, flush=True)
    server.run()


def run(cn="config_point", overrides=(), ticks=2000, connect=None, until_task=None, extra_ticks=0, jitter=None, trace=False,
        settle_ticks=0):
    """One closed-loop episode; returns the report dict (what main() prints).  connect = endpoint of a
    planner served by `--serve` in another process (otherwise the planner lives in this process).
    until_task: stop `extra_ticks` ticks after the task planner first hands out that task and return the
    world at that moment (`captured`: dof_state, root_state, task, goal as lists) -- bench.py's panda_pick row
    starts from the scene the product's own reach phase ends in.
    settle_ticks: after the success tick the world is stepped that many more ticks with the zero action the planner
    side returns once the task is done (reactive_tamp.py:52-54) before the final error is taken -- the reference's logs
    were written after the run, not at the success tick.
    jitter (point_env): dict(dyn_phase=int, box=(dx, dy), robot=(dx, dy)[, box_start=(x, y)]) -- the episode starts with the dyn-obs
    `dyn_phase` ticks into its walk and box / robot displaced (tools/band_stats.py: N episodes per scenario)."""
    overrides = list(overrides)
    compat.install(force_standins=True)
    import m3p2i_aip.utils.isaacgym_utils.isaacgym_wrapper as wrapper
    from m3p2i_aip.utils.skill_utils import check_and_apply_suction
    cfg = compat.make_config(cn, overrides)
    tamp = RemoteTamp(connect, cfg.mppi.device) if connect else Tamp(cfg)
    real = wrapper.IsaacGymWrapper(cfg.isaacgym, cfg.env_type, num_envs=1, viewer=False, device=cfg.mppi.device,
                                   cube_on_shelf=cfg.cube_on_shelf)
    nu = real.dofs_per_robot
    phase = 0
    if jitter and cfg.env_type == "point_env":
        phase = int(jitter.get("dyn_phase", 0))
        ib = int(real._get_actor_index_by_name("box"))
        if jitter.get("box_start") is not None:      # the scenario's own start of the box (corner2_*: in the far corner)
            real._root_state[0, ib, 0] = float(jitter["box_start"][0])
            real._root_state[0, ib, 1] = float(jitter["box_start"][1])
        real._root_state[0, ib, 0] += float(jitter.get("box", (0, 0))[0])
        real._root_state[0, ib, 1] += float(jitter.get("box", (0, 0))[1])
        # the dyn-obs where its walk (update_dyn_obs: 1 cm per tick along the diagonal, back for ticks 0-25 and
        # 75-99 of each 100, forth in between) has taken it after `phase` ticks: same track as an unjittered run
        off = sum(0.01 if 25 < (i % 100) < 75 else -0.01 for i in range(phase))
        idn = int(real._get_actor_index_by_name("dyn-obs"))
        real._root_state[0, idn, 0] += off
        real._root_state[0, idn, 1] += off
        real._dof_state[0, 0] += float(jitter.get("robot", (0, 0))[0])
        real._dof_state[0, 2] += float(jitter.get("robot", (0, 0))[1])
        real.set_dof_state_tensor(real._dof_state)
        real.set_actor_root_state_tensor(real._root_state)
    if jitter and cfg.env_type == "panda_env" and "cube" in jitter:
        ia = int(real._get_actor_index_by_name("cubeA"))
        real._root_state[0, ia, 0] += float(jitter["cube"][0])
        real._root_state[0, ia, 1] += float(jitter["cube"][1])
        real.set_actor_root_state_tensor(real._root_state)
    coll_ticks = 0          # ticks with a contact force on the dyn-obs: |Fx| + |Fy| > 0.1, the test of
                            # get_motion_cost (cost_functions.py:158-169) applied to the REAL world (plot_point.py col 17)
    timeline, lat = [], []
    success_tick = None
    stop_at, captured = None, None
    full = []
    path = []               # trace=True: per tick [robot x, y, box x, y, box quat z, w, dyn-obs x, y, action x, y] (point_env)
    for i in range(ticks):
        if cfg.env_type == "point_env":
            real.update_dyn_obs(i + phase)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        action = tamp.run_tamp(real._dof_state, real._root_state)
        action_host = action.cpu()               # command() wall time incl. the action on the host
        lat.append(time.perf_counter() - t0)
        task = tamp.task if connect else tamp.task_planner.task
        if not timeline or timeline[-1][1] != task:
            timeline.append((i, task))
        if tamp.task_success:
            success_tick = i
            break
        if until_task is not None and stop_at is None and task == until_task:
            stop_at = i + extra_ticks
        if stop_at is not None and i >= stop_at:
            goal = tamp.goal if connect else tamp.task_planner.curr_goal.float().cpu().reshape(-1).tolist()
            captured = dict(tick=i, task=task, goal=[float(x) for x in goal],
                            dof_state=real._dof_state[0].cpu().tolist(), root_state=real._root_state[0].cpu().tolist())
            break
        if trace and cfg.env_type == "point_env":
            b = real.get_actor_link_by_name("box", "box")[0].cpu()
            dyn = real.get_actor_position_by_name("dyn-obs")[0].cpu()
            rp = real.robot_pos[0].cpu()
            path.append([float(rp[0]), float(rp[1]), float(b[0]), float(b[1]), float(b[5]), float(b[6]), float(dyn[0]),
                         float(dyn[1]), float(action_host[0]), float(action_host[1])])
        if trace and cfg.env_type == "panda_env":
            hand = real.get_actor_link_by_name("panda", "panda_hand")[0, :7].cpu().tolist()
            cube = real.get_actor_link_by_name("cubeA", "box")[0, :7].cpu().tolist()
            path.append([task] + hand + cube + real._dof_state[0, [14, 16]].cpu().tolist() + action_host[7:9].tolist())
            full.append(dict(tick=i, task=task, dof_state=real._dof_state[0].cpu().tolist(), root_state=real._root_state[0].cpu().tolist(),
                             action=action_host.reshape(-1).tolist()))     # (enough to replay the world on the CPU oracle)
        real.set_dof_velocity_target_tensor(action.view(1, nu))
        if cfg.env_type == "point_env":
            cfg.suction_active = tamp.suction_active
            check_and_apply_suction(cfg, real, action.view(1, nu))
        real.step()
        if cfg.env_type == "point_env":
            f = real.get_actor_contact_forces_by_name("dyn-obs", "box")[0]
            coll_ticks += int(float(f[0].abs() + f[1].abs()) > 0.1)
    for k in range(settle_ticks if success_tick is not None else 0):
        if cfg.env_type == "point_env":
            real.update_dyn_obs(i + 1 + k + phase)
        real.set_dof_velocity_target_tensor(torch.zeros(1, nu, device=cfg.mppi.device))
        real.step()
    res = dict(config=cn, overrides=overrides, K=cfg.mppi.num_samples, T=cfg.mppi.horizon,
               ticks=i + 1, success=success_tick is not None, transport="rpc " + connect if connect else "in-process",
               sim_time_s=(i + 1) * cfg.isaacgym.dt, timeline=timeline,
               command_ms_p50=float(np.percentile(lat[5:], 50) * 1e3), command_ms_p99=float(np.percentile(lat[5:], 99) * 1e3),
               command_hz_mean=float(1.0 / np.mean(lat[5:])))
    if cfg.env_type == "point_env":
        goal = torch.tensor(tamp.goal) if connect else tamp.task_planner.curr_goal.float().cpu()
        who = real.robot_pos[0].cpu() if cfg.task == "navigation" else real.get_actor_position_by_name("box")[0, :2].cpu()
        res["final_pos_error"] = float(torch.norm(who - goal))
        res["dyn_obs_collision_ticks"] = coll_ticks
    else:
        cube = real.get_actor_link_by_name("cubeA", "box")[0, :3].cpu()
        goal = real.get_actor_link_by_name("cubeB", "box")[0, :3].cpu()
        res["cube_to_goal_xy"] = float(torch.norm(cube[:2] - goal[:2]))
        res["cube_height_above_goal"] = float(cube[2] - goal[2])
    if captured is not None:
        res["captured"] = captured
    if trace:
        res["trace"] = path
        if full:
            res["full"] = full
    real.stop_sim()
    tamp.close()
    return res


def main(argv):
    cn, ticks, out, overrides, serve_ep, connect_ep, trace = "config_point", 2000, None, [], None, None, False
    it = iter(argv)
    for a in it:
        if a in ("-cn", "--config-name"):
            cn = next(it)
        elif a == "--ticks":
            ticks = int(next(it))
        elif a == "--json":
            out = next(it)
        elif a == "--serve":
            serve_ep = next(it)
        elif a == "--connect":
            connect_ep = next(it)
        elif a == "--trace":
            trace = True
        elif a == "--torch-blobs":      # A/B: every tensor through torch.save / torch.load (m3p2i_aip_amd/blobs.py switched off)
            from m3p2i_aip_amd import blobs
            blobs.CODEC.enabled = False
        else:
            overrides.append(a)
    if serve_ep:
        return serve(cn, overrides, serve_ep)
    res = run(cn, overrides, ticks, connect=connect_ep, trace=trace)
    print(json.dumps({k: v for k, v in res.items() if k not in ('trace', 'full')}))
    if out:
        os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
