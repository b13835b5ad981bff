"""IsaacGymWrapper -- the reference's simulator facade, re-hosted on the HIP integrator.

Answers to the public surface of src/m3p2i_aip/utils/isaacgym_utils/isaacgym_wrapper.py that
the planner side uses (reactive_tamp.py, cost_functions.py, skill_utils.py, task_planner.py):

  constructor                         isaacgym_wrapper.py:39-66
  _dof_state / _root_state / _rigid_body_state / _net_contact_force   :98-112
  robot_pos / robot_vel               :120-126
  _get_actor_index_by_name ... get_actor_contact_forces_by_name       :128-188
  set_dof_state_tensor / set_actor_root_state_tensor                  :190-194
  set_dof_velocity_target_tensor      :196
  apply_rigid_body_force_tensors      :202-203
  update_dyn_obs                      :205-220
  step                                :354-360

The K environments are K lanes of one HIP kernel (m3_sim_step).  The four state tensors are
ordinary torch CUDA tensors (torch owns the memory); the library keeps its own SoA copy and
the two are synchronised exactly where the reference synchronises with PhysX: `set_*_state
_tensor` uploads, `step()` refreshes.  There is no viewer (reactive_tamp.py passes
viewer=False; sim.py's viewer=True runs headless with a warning); keyboard_control raises.
"""
from __future__ import annotations

import weakref
from dataclasses import dataclass, field
from typing import List

import torch

from . import _lib as L
from . import scenes
from .engine import HipEngine, make_config

_LIVE = weakref.WeakSet()  # wrappers alive in this process (planner auto-discovery)


@dataclass
class IsaacGymConfig:
    dt: float = 0.05
    substeps: int = 2
    use_gpu_pipeline: bool = True
    num_threads: int = 8
    viewer: bool = False
    spacing: float = 10
    camera_pos: List[float] = field(default_factory=lambda: [1.5, 6, 8])
    camera_target: List[float] = field(default_factory=lambda: [1.5, 0, 0])


class IsaacGymWrapper:
    def __init__(self, cfg: IsaacGymConfig, env_type: str = "point_env", num_envs: int = 1,
                 viewer: bool = False, device: str = "cuda:0", cube_on_shelf: bool = False,
                 k_offset: int = 0, num_envs_global: int | None = None):
        if viewer or getattr(cfg, "viewer", False):
            # scripts/sim.py:19-27 asks for the viewer of its 1-env world; this build has none: the world
            # runs headless (visualize_trajs / play_with_cube are no-ops), keyboard_control raises
            import warnings
            warnings.warn("IsaacGymWrapper(viewer=True): this build has no viewer, the world runs headless")
        if env_type not in scenes.ENVS:
            raise ValueError(f"unknown env_type {env_type!r}")
        self.cfg = cfg
        self.env_type = env_type
        self.env_cfg = [scenes.Actor(**vars(a)) for a in scenes.ENVS[env_type]]
        for i, a in enumerate(self.env_cfg):
            a.handle = i
        self.device = device
        self.num_envs = int(num_envs)
        self.cube_on_shelf = cube_on_shelf
        self.viewer = None
        dev = torch.device(device)
        if dev.type != "cuda" or not torch.cuda.is_available():
            raise L.M3Error(f"IsaacGymWrapper(device={device!r}): the rollout simulator runs on a HIP "
                            "device only (no CPU fallback)")
        self.robot_indices = torch.tensor(
            [i for i, a in enumerate(self.env_cfg) if a.type == "robot"], device=dev)
        self.robot_per_env = len(self.robot_indices)
        self.dofs_per_robot = scenes.DOFS[env_type]
        self.num_dofs = self.dofs_per_robot * self.num_envs
        self.bodies_per_env = scenes.num_bodies(env_type)
        self.num_bodies = self.bodies_per_env * self.num_envs
        nA = len(self.env_cfg)

        K = self.num_envs
        self._engine = HipEngine(make_config(
            K=num_envs_global or K, K_local=K, k_offset=k_offset, T=1,
            nu=self.dofs_per_robot, env_type=env_type, dt=cfg.dt, substeps=cfg.substeps,
            device=dev.index or 0, sim_only=True, filter_u=False, cube_on_shelf=cube_on_shelf))

        # initial scene (start_sim / set_initial_joint_pose / acquire_states: :68-118,222-240)
        root = torch.zeros(nA, 13)
        for i, a in enumerate(self.env_cfg):
            pos = list(a.init_pos)
            if a.name == "cubeA" and cube_on_shelf:
                pos = list(scenes.CUBE_A_ON_SHELF)
            if env_type == "point_env" and a.type == "box" and not a.fixed:
                pos[2] = a.size[2] / 2  # rests on the ground plane
            root[i, 0:3] = torch.tensor(pos)
            root[i, 3:7] = torch.tensor(a.init_ori)
        self._root_state = root.unsqueeze(0).repeat(K, 1, 1).to(dev).contiguous()
        rb = torch.zeros(self.bodies_per_env, 13)
        b = 0
        for i, a in enumerate(self.env_cfg):
            for _ in a.links:
                rb[b] = root[i]
                b += 1
        self._rigid_body_state = rb.unsqueeze(0).repeat(K, 1, 1).to(dev).contiguous()
        dof = torch.zeros(2 * self.dofs_per_robot)
        robot = [a for a in self.env_cfg if a.type == "robot"][0]
        if robot.init_joint_pose:
            dof = torch.tensor(robot.init_joint_pose, dtype=torch.float32)
        self._dof_state = dof.unsqueeze(0).repeat(K, 1).to(dev).contiguous()
        self._net_contact_force = torch.zeros(K, self.bodies_per_env, 3, device=dev)
        self._engine.sim_bind_views(self._dof_state, self._root_state, self._rigid_body_state,
                                    self._net_contact_force)
        self._engine.sim_pull_state()
        self._engine.sim_push_state()
        self._body_index_cache, self._actor_index_cache, self._rb_host, self._state_version = {}, {}, None, 0
        self._idx02 = torch.tensor([0, 2], device=dev)
        self._idx13 = torch.tensor([1, 3], device=dev)
        _LIVE.add(self)

    # ---- state access (isaacgym_wrapper.py:120-188) ----
    @property
    def robot_pos(self):
        return torch.index_select(self._dof_state, 1, self._idx02)

    @property
    def robot_vel(self):
        return torch.index_select(self._dof_state, 1, self._idx13)

    def _get_actor_index_by_name(self, name: str):
        # (one index tensor per actor, made once: torch.tensor(..., device=) is a synchronous host-to-device copy, and the task
        # planners look an actor up every tick)
        t = self._actor_index_cache.get(name)
        if t is None:
            t = self._actor_index_cache[name] = torch.tensor([a.name for a in self.env_cfg].index(name), device=self.device)
        return t

    def _get_actor_index_by_robot_index(self, robot_idx: int):
        return self.robot_indices[robot_idx]

    def get_actor_position_by_actor_index(self, actor_idx):
        return torch.index_select(self._root_state, 1, actor_idx.reshape(1))[:, 0, 0:3]

    def get_actor_position_by_name(self, name: str):
        return self.get_actor_position_by_actor_index(self._get_actor_index_by_name(name))

    def get_actor_position_by_robot_index(self, robot_idx: int):
        return self.get_actor_position_by_actor_index(self._get_actor_index_by_robot_index(robot_idx))

    def get_actor_velocity_by_actor_index(self, idx):
        return torch.index_select(self._root_state, 1, idx.reshape(1))[:, 0, 7:10]

    def get_actor_velocity_by_name(self, name: str):
        return self.get_actor_velocity_by_actor_index(self._get_actor_index_by_name(name))

    def get_actor_velocity_by_robot_index(self, robot_idx: int):
        return self.get_actor_velocity_by_actor_index(self._get_actor_index_by_robot_index(robot_idx))

    def get_actor_orientation_by_actor_index(self, idx):
        return torch.index_select(self._root_state, 1, idx.reshape(1))[:, 0, 3:7]

    def get_actor_orientation_by_name(self, name: str):
        return self.get_actor_orientation_by_actor_index(self._get_actor_index_by_name(name))

    def get_actor_orientation_by_robot_index(self, robot_idx: int):
        return self.get_actor_orientation_by_actor_index(self._get_actor_index_by_robot_index(robot_idx))

    def get_rigid_body_by_rigid_body_index(self, rigid_body_idx):
        return torch.index_select(self._rigid_body_state, 1, rigid_body_idx.reshape(1))[:, 0, :]

    def _body_index(self, actor_name: str, link_name: str):
        # (one index tensor per link, made once: torch.tensor(..., device=) is a synchronous host-to-device copy)
        key = (actor_name, link_name)
        t = self._body_index_cache.get(key)
        if t is None:
            t = self._body_index_cache[key] = torch.tensor(scenes.body_index(self.env_type, actor_name, link_name),
                                                           device=self.device)
        return t

    def env0_link_states_host(self):
        """Env 0's rigid-body states [bodies_per_env, 13] on the HOST (numpy f32): one device-to-host copy per world state,
        shared by everything that reads link poses of env 0 until the next step() / state upload.  For host-side consumers
        that branch on poses every tick -- the task planner (task_planner.py:62-107 reads four links and syncs on each)."""
        c = self._rb_host
        if c is None or c[0] != self._state_version:
            c = self._rb_host = (self._state_version, self._rigid_body_state[0].cpu().numpy())
        return c[1]

    def link_row(self, actor_name: str, link_name: str) -> int:
        return scenes.body_index(self.env_type, actor_name, link_name)

    def get_actor_link_by_name(self, actor_name: str, link_name: str):
        return self.get_rigid_body_by_rigid_body_index(self._body_index(actor_name, link_name))

    def get_actor_contact_forces_by_name(self, actor_name: str, link_name: str):
        return self._net_contact_force[:, self._body_index(actor_name, link_name)]

    # ---- uploads (isaacgym_wrapper.py:190-203) ----
    def _adopt(self, mine, given):
        if given is not mine and given.data_ptr() != mine.data_ptr():
            mine.copy_(given.reshape(mine.shape))

    def set_dof_state_tensor(self, u):
        self._adopt(self._dof_state, u)
        self._engine.sim_pull_state()
        self._state_version += 1

    def set_actor_root_state_tensor(self, u):
        self._adopt(self._root_state, u)
        self._engine.sim_pull_state()
        self._state_version += 1

    # True: set_dof_velocity_target_tensor hands the caller's tensor to the next step() instead of copying it (the
    # closed-loop tools, which own their tensors; see HipEngine.sim_set_velocity_target)
    zero_copy_targets = False

    def set_dof_velocity_target_tensor(self, u):
        self._engine.sim_set_velocity_target(u.reshape(self.num_envs, self.dofs_per_robot), zero_copy=self.zero_copy_targets)

    def set_dof_actuation_force_tensor(self, u):
        raise NotImplementedError("effort mode is not used by the planner path "
                                  "(actors are velocity-driven: isaacgym_wrapper.py:341-344)")

    def apply_rigid_body_force_tensors(self, u):
        self._engine.sim_apply_body_forces(u.reshape(self.num_envs, self.bodies_per_env, 3))

    def update_dyn_obs(self, i, period=100):
        """The dyn-obs of the point_env walks 1 cm per tick along the diagonal, forth for half a period and
        back (isaacgym_wrapper.py:205-220); in the panda_env the offset is zero (the state is uploaded unchanged).
        The shift of the root_state row and the state upload are one launch: no index tensors, no host round trip."""
        if getattr(self, "_dyn_obs_row", None) is None:
            self._dyn_obs_row = [a.name for a in self.env_cfg].index("dyn-obs")
        forth = period / 4 < i % period < period / 4 * 3
        if self.env_type == "point_env":
            d = 0.01 if forth else -0.01
            self._engine.sim_shift_actor(self._dyn_obs_row, d, d, 0.0)
            self._state_version += 1
        else:
            self.set_actor_root_state_tensor(self._root_state)

    def step(self):
        self._engine.sim_step()
        self._state_version += 1

    def stop_sim(self):
        self._engine.close()

    # ---- viewer-only entry points of the reference (isaacgym_wrapper.py:374-460) ----
    def visualize_trajs(self, trajs):
        pass

    def play_with_cube(self):
        return 0

    def keyboard_control(self):
        raise NotImplementedError("no viewer")


def live_wrappers():
    return list(_LIVE)
