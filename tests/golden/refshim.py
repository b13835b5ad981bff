"""Helpers used ONLY by tests/golden/make_golden.py (build container, reference mounted at
/root/reference).  Imports the reference's planner with two stub modules and provides an
oracle-backed object that answers to the IsaacGymWrapper API the reference's plugins use
(isaacgym_wrapper.py:120-203, 354-360), so the reference's own M3P2I + Objective code can
be driven end-to-end without Isaac Gym.  Nothing here travels to the GPU box as a
dependency of the tests: the tests read the .npz fixtures only.
"""
import sys
import types

import numpy as np
import torch

REF_SRC = "/root/reference/src"


def import_reference():
    """Import the reference's motion-planner modules with isaacgym / ghalton stubbed."""
    if "isaacgym" not in sys.modules:
        ig = types.ModuleType("isaacgym")
        gymapi = types.ModuleType("isaacgym.gymapi")

        class SimParams:  # only referenced in a type annotation (isaacgym_wrapper.py:18)
            pass

        gymapi.SimParams = SimParams
        gymtorch = types.ModuleType("isaacgym.gymtorch")
        ig.gymapi, ig.gymtorch = gymapi, gymtorch
        sys.modules["isaacgym"] = ig
        sys.modules["isaacgym.gymapi"] = gymapi
        sys.modules["isaacgym.gymtorch"] = gymtorch
    if "ghalton" not in sys.modules:
        gh = types.ModuleType("ghalton")  # third-party, absent: only the in-tree
        gh.EA_PERMS = []                  # use_ghalton=False branch is exercised
        gh.GeneralizedHalton = object
        sys.modules["ghalton"] = gh
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    from m3p2i_aip.planners.motion_planner import mppi, m3p2i, cost_functions
    from m3p2i_aip.utils import mppi_utils, skill_utils

    class _TorchProxy:  # skill_utils.py:69 hard-codes device='cuda:0'
        def __getattr__(self, n):
            return getattr(torch, n)

        def zeros(self, *a, **k):
            k["device"] = "cpu"
            return torch.zeros(*a, **k)

    skill_utils.torch = _TorchProxy()
    return types.SimpleNamespace(mppi=mppi, m3p2i=m3p2i, cost_functions=cost_functions,
                                 mppi_utils=mppi_utils, skill_utils=skill_utils)


# actor order of this build (DESIGN.md "Scene tables"): non-robot actors in numeric file
# order, robot last -- the order skill_utils.py:89-90 assumes (robot link = last body).
POINT_ACTORS = ["wall-1", "wall-2", "wall-3", "wall-4", "obs", "dyn-obs", "box", "goal",
                "yaxis", "xaxis", "point_robot"]


class SynthSim:
    """Synthetic sim state for cost-function goldens (no dynamics)."""

    def __init__(self, robot_pos, robot_vel, box_pos, dyn_force=None):
        K = robot_pos.shape[0]
        self.num_envs = K
        self.bodies_per_env = 13
        self._robot_pos = torch.as_tensor(robot_pos, dtype=torch.float32)
        self._robot_vel = torch.as_tensor(robot_vel, dtype=torch.float32)
        bp = torch.zeros(K, 3)
        bp[:, :2] = torch.as_tensor(box_pos, dtype=torch.float32)
        self._box = bp
        self._dyn_force = torch.zeros(K, 3) if dyn_force is None else torch.as_tensor(
            dyn_force, dtype=torch.float32)
        self.applied = None

    @property
    def robot_pos(self):
        return self._robot_pos

    @property
    def robot_vel(self):
        return self._robot_vel

    def get_actor_position_by_name(self, name):
        assert name == "box"
        return self._box

    def _get_actor_index_by_name(self, name):
        return torch.tensor(POINT_ACTORS.index(name))

    def apply_rigid_body_force_tensors(self, f):
        self.applied = f.clone()

    def get_actor_contact_forces_by_name(self, actor, link):
        assert actor == "dyn-obs"
        return self._dyn_force.clone()


class OracleSim:
    """K oracle worlds behind the wrapper API used by reactive_tamp.py:63-73."""

    def __init__(self, K, world0, scene=None):
        import oracle as O
        self.O = O
        self.sc = scene or O.default_scene()
        self.num_envs = K
        self.bodies_per_env = 13
        self.dofs_per_robot = 2
        self.world0 = np.array(world0, np.float32).reshape(-1)[:O.WORLD_FLOATS].copy()
        self.worlds = np.tile(self.world0, (K, 1)).astype(np.float32)
        self.u = np.zeros((K, 2), np.float32)

    def reset(self, world0=None):
        """what run_tamp does (reactive_tamp.py:45-48): states are overwritten, pending
        applied forces are NOT cleared."""
        O = self.O
        if world0 is not None:
            self.world0 = np.array(world0, np.float32).reshape(-1)[:O.WORLD_FLOATS].copy()
        pend = self.worlds[:, O.W_FEXT_R:O.W_FEXT_B + 2].copy()
        self.worlds[:] = self.world0
        self.worlds[:, O.W_FEXT_R:O.W_FEXT_B + 2] = pend

    @property
    def robot_pos(self):
        return torch.from_numpy(self.worlds[:, [0, 1]].copy())

    @property
    def robot_vel(self):
        return torch.from_numpy(self.worlds[:, [4, 5]].copy())

    def get_actor_position_by_name(self, name):
        O = self.O
        base = {"box": O.W_B, "dyn-obs": O.W_D}[name]
        p = np.zeros((self.num_envs, 3), np.float32)
        p[:, :2] = self.worlds[:, base:base + 2]
        p[:, 2] = 0.05
        return torch.from_numpy(p)

    def _get_actor_index_by_name(self, name):
        return torch.tensor(POINT_ACTORS.index(name))

    def apply_rigid_body_force_tensors(self, f):
        O = self.O
        f = f.view(self.num_envs, self.bodies_per_env, 3).numpy()
        self.worlds[:, O.W_FEXT_B:O.W_FEXT_B + 2] = f[:, POINT_ACTORS.index("box"), :2]
        self.worlds[:, O.W_FEXT_R:O.W_FEXT_R + 2] = f[:, -1, :2]

    def get_actor_contact_forces_by_name(self, actor, link):
        O = self.O
        base = {"dyn-obs": O.W_FC_D, "box": O.W_FC_B}[actor]
        p = np.zeros((self.num_envs, 3), np.float32)
        p[:, :2] = self.worlds[:, base:base + 2]
        return torch.from_numpy(p)

    def set_dof_velocity_target_tensor(self, u):
        self.u = u.detach().numpy().astype(np.float32).reshape(self.num_envs, 2).copy()

    def step(self):
        self.O.step_batch(self.sc, self.worlds, self.u)

    @property
    def _dof_state(self):
        return torch.from_numpy(self.worlds[:, [0, 4, 1, 5]].copy())
