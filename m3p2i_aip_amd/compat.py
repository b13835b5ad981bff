"""Compatibility layer: lets the reference's `scripts/reactive_tamp.py` (and `scripts/sim.py`)
import and run unchanged on this package.

`install()` registers this package's classes under the module names the scripts import
(`reactive_tamp.py:1-8`, `sim.py:1-6`).  Nothing of the reference is copied and Isaac Gym is
not needed.  Launcher / RPC pieces (hydra, zerorpc) are outside the hot path (SURVEY.md
section 8(f)); when the real packages are absent, small stand-ins cover exactly the calls the
scripts make (`hydra.main` composition of config_point / config_panda + key=value overrides;
`zerorpc.Server / Client` over TCP + msgpack: m3p2i_aip_amd/rpc.py).

Host-side helpers restated here (all O(1) per command, no data parallelism):
  PLANNER_SIMPLE / PLANNER_AIF_PANDA / AiAgent / adapt_act_sel / MDPIsCubeAtReal
                            planners/task_planner/*.py  (-> m3p2i_aip_amd/task_planner.py)
  torch_to_bytes / bytes_to_torch   utils/data_transfer.py:4-12
  calculate_suction / check_suction_condition / check_and_apply_suction
                            utils/skill_utils.py:36-94 -> HIP kernel k_sim_suction (device, no host sync)
  time_tracking             utils/skill_utils.py:25-33
  ExampleConfig defaults    config/config_point.yaml, config_panda.yaml, mppi/*.yaml,
                            isaacgym/*.yaml, config/config_store.py:7-29
"""
from __future__ import annotations

import io
import sys
import time
import types
from dataclasses import dataclass, field
from typing import List

import torch

from . import cost_functions, isaacgym_wrapper, planner, task_planner


# ------------------------------------------------------------------ data_transfer.py:4-12
def torch_to_bytes(t) -> bytes:
    """data_transfer.py:4-7: a torch.save archive.  Small plain tensors -- the states and the action of every tick -- through
    blobs.TensorBlobCodec (the same archive, written by patching a copy: ~10 us instead of ~170)."""
    from .blobs import CODEC
    return CODEC.save(t)


def bytes_to_torch(b: bytes):
    """data_transfer.py:9-12.  The blob comes off a network socket: weights_only=True restricts the unpickler to
    tensors and plain containers (bool / int / float / str / list / dict), which is all the scripts exchange.  An archive that
    equals a known one of a small plain tensor everywhere but in its payload is read by lifting the payload out (CRC checked)."""
    from .blobs import CODEC
    try:
        return CODEC.load(b)
    except TypeError:       # torch < 1.13 has no weights_only: refuse rather than fall back to the full unpickler
        raise RuntimeError("bytes_to_torch needs torch >= 1.13 (torch.load(weights_only=True)); this torch is "
                           + torch.__version__) from None


# ------------------------------------------------------------------ task_planner.py (module)
PLANNER_SIMPLE = task_planner.PLANNER_SIMPLE
PLANNER_AIF_PANDA = task_planner.PLANNER_AIF_PANDA
set_task_planner = task_planner.set_task_planner


# ------------------------------------------------------------------ skill_utils.py:25-94
# The suction skill of the 1-env "real world" (scripts/sim.py:41-49).  The geometry lives in the HIP
# library (m3_sim_suction_forces / m3_sim_check_and_apply_suction, csrc/rollout_point.hip:
# k_sim_suction); what stays on the host is the part that only reads the config.
def _device_gate(cfg):
    """cfg.suction_active handed over as a device tensor (planner.pull_preference_tensor()): the kernel
    reads it, the host never does."""
    g = cfg.suction_active
    return g if (torch.is_tensor(g) and g.is_cuda) else None


def _suction_enabled(cfg):
    if cfg.task not in ("pull", "push_pull"):
        return False
    return True if _device_gate(cfg) is not None else bool(cfg.suction_active)


def calculate_suction(cfg, sim):
    """[num_envs, bodies_per_env, 3] suction forces on the box and the robot's last link."""
    return sim._engine.sim_suction_forces(cfg.kp_suction)


def check_suction_condition(cfg, sim, action):
    """bool for env 0 (one host sync, as the reference's .item())."""
    if not _suction_enabled(cfg):
        return False
    g = _device_gate(cfg)
    return bool(sim._engine.sim_check_and_apply_suction(action, cfg.kp_suction, apply=False, want_flags=True,
                                                        enabled=None if g is None else g.to(torch.int32).reshape(1))[0].item())


def check_and_apply_suction(cfg, sim, action):
    """Stages the suction pair for the next sim.step() wherever the condition holds; enqueued on the
    stream, nothing is read back (the reference's "suction!!!" / "no suction..." prints are dropped)."""
    if _suction_enabled(cfg):
        g = _device_gate(cfg)
        sim._engine.sim_check_and_apply_suction(action, cfg.kp_suction, apply=True,
                                                enabled=None if g is None else g.to(torch.int32).reshape(1))


def time_tracking(t, cfg):
    actual_dt = time.time() - t
    if cfg.isaacgym.dt / actual_dt > 1.0:
        time.sleep(cfg.isaacgym.dt - actual_dt)
    return time.time()


# ------------------------------------------------------------------ config_store.py + yaml values
@dataclass
class ExampleConfig:
    mppi: planner.MPPIConfig
    isaacgym: isaacgym_wrapper.IsaacGymConfig
    env_type: str = "point_env"
    task: str = "push"
    goal: List[float] = field(default_factory=lambda: [-3.75, -3.75])
    kp_suction: int = 0
    suction_active: bool = False
    multi_modal: bool = False
    pre_height_diff: float = 0.0
    cube_on_shelf: bool = False
    render: bool = False
    n_steps: int = 0
    nx: int = 4
    avoid_dyn_obs: bool = False     # EXTENSION (not a key of the reference's config_store.py): cost_functions.Objective


def make_config(config_name="config_point", overrides=()):
    """config/config_{point,panda}.yaml + the mppi/ and isaacgym/ groups they compose, then
    `key=value` / `group.key=value` overrides as on the reference's command line."""
    import yaml
    if config_name == "config_point":
        m = planner.MPPIConfig(mppi_mode="halton-spline", sampling_method="halton", num_samples=200,
                               horizon=15, nx=4, device="cuda:0", lambda_=0.5, u_min=[-3.0, -3.0],
                               u_max=[3.0, 3.0], noise_sigma=[[3.0, 0.0], [0.0, 3.0]], u_per_command=15,
                               sample_null_action=True, filter_u=True, use_priors=False)
        g = isaacgym_wrapper.IsaacGymConfig(dt=0.05, spacing=10)
        cfg = ExampleConfig(mppi=m, isaacgym=g, env_type="point_env", task="push", goal=[-3.75, -3.75],
                            kp_suction=400, suction_active=True, multi_modal=False)
    elif config_name == "config_panda":
        sig = [[0.0] * 9 for _ in range(9)]
        for i in range(7):
            sig[i][i] = 10.0
        sig[7][7] = sig[8][8] = 0.8
        m = planner.MPPIConfig(mppi_mode="halton-spline", sampling_method="halton", num_samples=200,
                               horizon=12, nx=18, device="cuda:0", u_min=[-2.0] * 7 + [-1.5] * 2,
                               u_max=[2.0] * 7 + [1.5] * 2, lambda_=0.05, noise_sigma=sig, u_per_command=12,
                               sample_null_action=True, filter_u=True, use_priors=False)
        g = isaacgym_wrapper.IsaacGymConfig(dt=0.01, spacing=2, camera_pos=[0, 1.5, 2.8], camera_target=[0, 0, 1])
        cfg = ExampleConfig(mppi=m, isaacgym=g, env_type="panda_env", task="reactive_pick", goal=[0.0] * 7,
                            multi_modal=False, pre_height_diff=0.05, cube_on_shelf=False)
    else:
        raise ValueError(f"unknown config {config_name!r}")
    for ov in overrides:
        key, _, val = ov.partition("=")
        obj = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            obj = getattr(obj, p)
        if not hasattr(obj, parts[-1]):
            raise AttributeError(f"unknown config key {key!r}")
        setattr(obj, parts[-1], yaml.safe_load(val))
    return cfg


def _hydra_standin():
    m = types.ModuleType("hydra")

    def main(version_base=None, config_path=None, config_name="config_point"):
        def deco(fn):
            def run(*a, **k):
                if a or k:
                    return fn(*a, **k)
                argv, name, ov = sys.argv[1:], config_name, []
                i = 0
                while i < len(argv):
                    if argv[i] in ("-cn", "--config-name"):
                        name = argv[i + 1]
                        i += 2
                    else:
                        ov.append(argv[i])
                        i += 1
                return fn(make_config(name, ov))
            return run
        return deco

    m.main = main
    return m


def _zerorpc_standin():
    """`zerorpc` is not installed here: the Server / Client pair of m3p2i_aip_amd.rpc (same calls, a plain
    TCP + msgpack transport) answers to the name, so reactive_tamp.py:89-94 and sim.py:29-49 run as two
    processes."""
    from . import rpc
    m = types.ModuleType("zerorpc")
    m.Server, m.Client = rpc.Server, rpc.Client
    m.RemoteError, m.LostRemote = rpc.RemoteError, rpc.LostRemote
    return m


def install(force_standins: bool = False):
    """Make `import m3p2i_aip...`, `import isaacgym`, `hydra`, `zerorpc` resolve to this build."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    ig = mod("isaacgym")
    ig.gymtorch = mod("isaacgym.gymtorch")
    ig.gymapi = mod("isaacgym.gymapi")
    root = mod("m3p2i_aip")
    mod("m3p2i_aip.planners")
    mp = mod("m3p2i_aip.planners.motion_planner")
    mp.m3p2i = mod("m3p2i_aip.planners.motion_planner.m3p2i", M3P2I=planner.M3P2I)
    mp.mppi = mod("m3p2i_aip.planners.motion_planner.mppi", MPPI=planner.MPPI, MPPIConfig=planner.MPPIConfig)
    mp.cost_functions = mod("m3p2i_aip.planners.motion_planner.cost_functions",
                            Objective=cost_functions.Objective)
    tp = mod("m3p2i_aip.planners.task_planner")
    tp.task_planner = mod("m3p2i_aip.planners.task_planner.task_planner", set_task_planner=set_task_planner,
                          PLANNER_SIMPLE=PLANNER_SIMPLE, PLANNER_AIF_PANDA=PLANNER_AIF_PANDA,
                          PLANNER_PATROLLING=task_planner.PLANNER_PATROLLING)
    tp.ai_agent = mod("m3p2i_aip.planners.task_planner.ai_agent", AiAgent=task_planner.AiAgent)
    tp.adaptive_action_selection = mod("m3p2i_aip.planners.task_planner.adaptive_action_selection",
                                       adapt_act_sel=task_planner.adapt_act_sel)
    tp.parallel_action_selection = mod("m3p2i_aip.planners.task_planner.parallel_action_selection",
                                       par_act_sel=task_planner.par_act_sel)
    tp.isaac_state_action_templates = mod(
        "m3p2i_aip.planners.task_planner.isaac_state_action_templates",
        MDPIsCubeAtReal=task_planner.MDPIsCubeAtReal,
        **{n: getattr(task_planner, n) for n in task_planner.TEMPLATE_TABLE})
    cfgm = mod("m3p2i_aip.config")
    cfgm.config_store = mod("m3p2i_aip.config.config_store", ExampleConfig=ExampleConfig)
    ut = mod("m3p2i_aip.utils")
    ut.data_transfer = mod("m3p2i_aip.utils.data_transfer", torch_to_bytes=torch_to_bytes,
                           bytes_to_torch=bytes_to_torch)
    ut.skill_utils = mod("m3p2i_aip.utils.skill_utils", calculate_suction=calculate_suction,
                         check_suction_condition=check_suction_condition,
                         check_and_apply_suction=check_and_apply_suction, time_tracking=time_tracking)
    iu = mod("m3p2i_aip.utils.isaacgym_utils")
    sys.modules["m3p2i_aip.utils.isaacgym_utils.isaacgym_wrapper"] = isaacgym_wrapper
    iu.isaacgym_wrapper = isaacgym_wrapper
    ut.isaacgym_utils = iu
    root.planners, root.config, root.utils = sys.modules["m3p2i_aip.planners"], cfgm, ut
    sys.modules["m3p2i_aip.planners"].motion_planner = mp
    sys.modules["m3p2i_aip.planners"].task_planner = tp
    for name, make in (("hydra", _hydra_standin), ("zerorpc", _zerorpc_standin)):
        have = False
        if not force_standins:
            try:
                __import__(name)
                have = True
            except Exception:
                have = False
        if not have:
            sys.modules[name] = make()
