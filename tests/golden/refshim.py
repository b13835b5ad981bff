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


PANDA_LINKS = {("panda", "panda_leftfinger"): "left", ("panda", "panda_rightfinger"): "right", ("cubeA", "box"): "cubeA",
               ("cubeB", "box"): "cubeB"}


class OraclePandaSim:
    """K oracle panda worlds (oracle/panda_chain.c) behind the wrapper API the reference's panda plugins
    use: dynamics of reactive_tamp.py:63-70 (set_dof_velocity_target_tensor, step, robot_pos / robot_vel =
    dofs 0 and 1) and the getters of cost_functions.py:91-169 (finger / cube link states, cubeA
    orientation, contact forces of table / shelf_stand / cubeB)."""

    def __init__(self, K, world0, scene=None):
        import oracle.panda as P
        self.P = P
        self.sc = scene or P.default_scene()
        self.num_envs = K
        self.dofs_per_robot = 9
        self.world0 = np.array(world0, np.float32).reshape(-1)[:P.WORLD_FLOATS].copy()
        self.worlds = np.tile(self.world0, (K, 1)).astype(np.float32)
        self.u = np.zeros((K, 9), np.float32)
        self._obs = None
        lib = P.lib()
        import ctypes as C
        lib.m3o_panda_infer_held.argtypes = [C.POINTER(P.PandaScene), C.POINTER(C.c_float)]
        self._lib, self._C = lib, C

    def reset(self, world0=None):
        """run_tamp (reactive_tamp.py:45-48): every environment takes the real world's state; whether the
        cube is clamped between the pads is inferred from the geometry (the tensors carry no such bit)."""
        P = self.P
        if world0 is not None:
            self.world0 = np.array(world0, np.float32).reshape(-1)[:P.WORLD_FLOATS].copy()
        w = self.world0.copy()
        self._lib.m3o_panda_infer_held(self._C.byref(self.sc), w.ctypes.data_as(self._C.POINTER(self._C.c_float)))
        self.worlds[:] = w
        self._obs = None

    def _observe(self):
        if self._obs is None:
            self._obs = np.stack([self.P.observe(self.sc, self.worlds[i]) for i in range(self.num_envs)])
        return self._obs

    @property
    def robot_pos(self):
        return torch.from_numpy(self.worlds[:, [0, 1]].copy())

    @property
    def robot_vel(self):
        return torch.from_numpy(self.worlds[:, [9, 10]].copy())

    @property
    def _dof_state(self):
        d = np.zeros((self.num_envs, 18), np.float32)
        d[:, 0::2] = self.worlds[:, 0:9]
        d[:, 1::2] = self.worlds[:, 9:18]
        return torch.from_numpy(d)

    def get_actor_link_by_name(self, actor, link):
        P, o = self.P, self._observe()
        x = np.zeros((self.num_envs, 13), np.float32)
        which = PANDA_LINKS[(actor, link)]
        if which == "left":
            x[:, 0:3], x[:, 3:7] = o[:, 0:3], o[:, 3:7]
        elif which == "right":
            x[:, 0:3] = o[:, 7:10]
            x[:, 6] = 1.0    # (the costs read only the right finger's position)
        elif which == "cubeA":
            x[:] = self.worlds[:, P.W_CUBEA:P.W_CUBEA + 13]
        else:
            x[:] = self.worlds[:, P.W_CUBEB:P.W_CUBEB + 13]
        return torch.from_numpy(x)

    def get_actor_orientation_by_name(self, actor):
        assert actor == "cubeA"
        P = self.P
        return torch.from_numpy(self.worlds[:, P.W_CUBEA + 3:P.W_CUBEA + 7].copy())

    def get_actor_contact_forces_by_name(self, actor, link):
        P = self.P
        base = {"table": P.W_FT, "shelf_stand": P.W_FS, "cubeB": P.W_FB}[actor]
        f = np.zeros((self.num_envs, 3), np.float32)
        f[:, :3] = self.worlds[:, base:base + 3]
        return torch.from_numpy(f)

    def set_dof_velocity_target_tensor(self, u):
        self.u = u.detach().numpy().astype(np.float32).reshape(self.num_envs, 9).copy()

    def step(self):
        self.P.step_batch(self.sc, self.worlds, self.u)
        self._obs = None
