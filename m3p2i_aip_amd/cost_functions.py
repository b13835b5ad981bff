"""Objective -- the reference's task-cost plugin, evaluated by the HIP cost kernel.

Mirror of src/m3p2i_aip/planners/motion_planner/cost_functions.py:
  Objective.__init__           :6-13
  Objective.update_objective   :15-17
  Objective.compute_cost       :19-36   (dispatch on task)

In FUSED mode (the planner's default when it recognises this plugin) compute_cost is never
called: the same device function runs inside the rollout kernel after every physics step.
In STEP mode (user-supplied dynamics/running_cost callables) compute_cost(sim) launches the
stand-alone cost kernel (m3_cost) on the wrapper's environments; like the reference's
get_pull_cost (:76) it also stages the suction force that acts during the next step.
"""
from __future__ import annotations

import weakref

import numpy as np
import torch

from . import _lib as L

_LIVE = weakref.WeakSet()

POINT_TASKS = ("navigation", "push", "pull", "push_pull")
PANDA_TASKS = ("reach", "pick", "place")


def attached_host_values(t):
    """The host values a tensor carries (`_m3_host`, attached by task_planner._device_tensor together with the tensor's
    version counter at that moment) -- or None when it carries none or was written to in place since (`curr_goal[2] += dz`):
    the caller then reads the tensor back."""
    host = getattr(t, "_m3_host", None)
    if host is None or getattr(t, "_m3_host_version", None) != t._version:
        return None
    return host


class Objective(object):
    def __init__(self, cfg):
        self.cfg = cfg
        self.multi_modal = cfg.multi_modal
        self.num_samples = cfg.mppi.num_samples
        self.half_samples = int(cfg.mppi.num_samples / 2)
        self.device = cfg.mppi.device
        self.pre_height_diff = getattr(cfg, "pre_height_diff", 0.0)
        self.tilt_cos_theta = 0.5
        self.task = getattr(cfg, "task", None)
        goal = getattr(cfg, "goal", None)
        self.goal = None if goal is None else torch.tensor(list(goal), dtype=torch.float32)
        self.gripper_cmd = 0
        # EXTENSION, off by default: `avoid_dyn_obs: True` in the config makes push / pull / push_pull add
        # get_motion_cost (:158-169) as navigation does; the shipped compute_cost returns before it (:23-29 vs :36)
        self.avoid_dyn_obs = bool(getattr(cfg, "avoid_dyn_obs", False)) and getattr(cfg, "env_type", "point_env") == "point_env"
        _LIVE.add(self)

    def update_objective(self, task, goal):
        self.task = task
        if torch.is_tensor(goal):
            self.goal = goal
            host = attached_host_values(goal)   # (a tensor made from host values that carries them: task_planner.py)
            if host is not None:
                self._goal_host = (host, id(goal), goal._version)
        else:
            self.goal = torch.tensor(goal, device=self.device)
            self._goal_host = ([float(x) for x in np.ravel(goal)], id(self.goal), self.goal._version)

    def goal_list(self):
        """Host copy of the goal for the C-ABI (passed by value).  Reading a device tensor
        synchronises the stream, which would serialise every command() with the previous one, so
        the copy is cached until the tensor object or its contents (torch's version counter) change
        -- reactive_tamp.py:79 hands the same `curr_goal` tensor over at every tick."""
        c = getattr(self, "_goal_host", None)
        if c is None or c[1] != id(self.goal) or c[2] != self.goal._version:
            c = ([float(x) for x in self.goal.detach().reshape(-1).cpu().tolist()], id(self.goal),
                 self.goal._version)
            self._goal_host = c
        return c[0]

    def compute_cost(self, sim):
        eng = getattr(sim, "_engine", None)
        if eng is None:
            raise TypeError("Objective.compute_cost needs the HIP-backed IsaacGymWrapper "
                            "(m3p2i_aip_amd.isaacgym_wrapper); there is no CPU fallback")
        if self.task is None or self.goal is None:
            raise RuntimeError("update_objective(task, goal) has not been called")
        if self.task == "push_pull" and not self.multi_modal:
            raise L.M3Error("task 'push_pull' needs multi_modal=True (cost_functions.py:27-29)")
        eng.set_multi_modal(self.multi_modal)  # the wrapper's handle learns it here (:9)
        eng.set_objective(self.task, self.goal_list(), self.gripper_cmd)
        if self.avoid_dyn_obs or getattr(eng, "_avoid_dyn_obs", False):
            eng.set_avoid_dyn_obs(self.avoid_dyn_obs)
            eng._avoid_dyn_obs = self.avoid_dyn_obs
        return eng.cost()


def live_objectives():
    return list(_LIVE)
