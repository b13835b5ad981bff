"""GPU tests of the drop-in Python API on the panda_env (config_panda wiring of
scripts/reactive_tamp.py:22-73): FUSED and STEP modes must agree with each other and with the
CPU oracle planner; wrapper getters must expose the poses the reference's costs read."""
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


class Tamp:
    def __init__(self, cfg):
        from m3p2i_aip_amd import isaacgym_wrapper as wrapper
        from m3p2i_aip_amd.cost_functions import Objective
        from m3p2i_aip_amd.planner import M3P2I
        self.sim = wrapper.IsaacGymWrapper(cfg.isaacgym, cfg.env_type, num_envs=cfg.mppi.num_samples,
                                           viewer=False, device=cfg.mppi.device, cube_on_shelf=cfg.cube_on_shelf)
        self.cfg = cfg
        self.objective = Objective(cfg)
        self.motion_planner = M3P2I(cfg, dynamics=self.dynamics, running_cost=self.running_cost)

    def dynamics(self, _, u, t=None):
        self.sim.set_dof_velocity_target_tensor(u)
        self.sim.step()
        states = torch.stack([self.sim.robot_pos[:, 0], self.sim.robot_vel[:, 0],
                              self.sim.robot_pos[:, 1], self.sim.robot_vel[:, 1]], dim=1)
        return states, u

    def running_cost(self, _):
        return self.objective.compute_cost(self.sim)


def make_cfg(K, T, multi_modal=False, fused=None, **mppi_kw):
    from m3p2i_aip_amd.isaacgym_wrapper import IsaacGymConfig
    from m3p2i_aip_amd.planner import MPPIConfig
    sig = [[0.0] * 9 for _ in range(9)]
    for i in range(7):
        sig[i][i] = 10.0
    sig[7][7] = sig[8][8] = 0.8
    kw = dict(num_samples=K, horizon=T, nx=18, device="cuda:0", lambda_=0.05,
              u_min=[-2.0] * 7 + [-1.5] * 2, u_max=[2.0] * 7 + [1.5] * 2, noise_sigma=sig,
              u_per_command=T, sample_null_action=True, filter_u=True, fused=fused)
    kw.update(mppi_kw)
    m = MPPIConfig(**kw)
    return SimpleNamespace(env_type="panda_env", multi_modal=multi_modal, suction_active=False, kp_suction=0,
                           pre_height_diff=0.05, task="reactive_pick", cube_on_shelf=False, mppi=m,
                           isaacgym=IsaacGymConfig(dt=0.01))


def test_wrapper_views_expose_fk_and_scene(oracle):
    import oracle.panda as P
    from m3p2i_aip_amd.isaacgym_wrapper import IsaacGymConfig, IsaacGymWrapper
    sim = IsaacGymWrapper(IsaacGymConfig(dt=0.01), "panda_env", num_envs=64, device="cuda:0")
    assert sim._root_state.shape == (64, 7, 13) and sim._rigid_body_state.shape == (64, 17, 13)
    assert sim._dof_state.shape == (64, 18) and sim.dofs_per_robot == 9 and sim.bodies_per_env == 17
    L = P.fk(P.default_scene(), [0, 0, 0, -2, 0, 1.8675, 0, 0.02, 0.02])
    lf = sim.get_actor_link_by_name("panda", "panda_leftfinger").cpu().numpy()
    rf = sim.get_actor_link_by_name("panda", "panda_rightfinger").cpu().numpy()
    np.testing.assert_array_equal(lf[0, :3], L["pos"][9])
    np.testing.assert_array_equal(rf[0, :3], L["pos"][10])
    np.testing.assert_array_equal(lf[0, 3:7], L["quat"][9])
    cube = sim.get_actor_link_by_name("cubeA", "box").cpu().numpy()
    np.testing.assert_allclose(cube[0, :7], [0.2, -0.2, 1.06, 0, 0, 0, 1])
    np.testing.assert_allclose(sim.get_actor_orientation_by_name("cubeA").cpu().numpy()[0], [0, 0, 0, 1])
    # stepping: cube settles on the table, joints follow velocity targets
    u = torch.zeros(64, 9, device="cuda:0")
    u[:, 0] = 1.0
    for _ in range(30):
        sim.set_dof_velocity_target_tensor(u)
        sim.step()
    assert sim.get_actor_position_by_name("cubeA")[0, 2].item() == pytest.approx(1.05, abs=1e-6)
    assert sim._dof_state[0, 1].item() == pytest.approx(1.0, abs=1e-3)
    assert sim._dof_state[0, 0].item() == pytest.approx(0.3, abs=0.02)


@pytest.mark.parametrize("task,mm", [("reach", False), ("reach", True), ("pick", False)])
def test_fused_step_and_oracle_agree(oracle, task, mm):
    import oracle.panda as P
    K, T = 128, 20
    rng = np.random.default_rng(9)
    delta = rng.standard_normal((K, T, 9)).astype(np.float32)
    goal = torch.tensor([0.2, 0.2, 1.115, 0, 0, 0, 1.0])
    outs = {}
    for mode, fused in (("fused", True), ("step", False), ("auto", None)):
        tamp = Tamp(make_cfg(K, T, mm, fused))
        pl = tamp.motion_planner
        pl.set_noise(delta)
        pl.update_gripper_command(task)
        tamp.objective.update_objective(task, goal)
        acts = []
        for call in range(3):
            # run_tamp resets the K rollout envs to the (unchanged) real state
            tamp.sim._dof_state[:] = tamp.sim._dof_state[0:1].clone() if call == 0 else dof0
            tamp.sim._root_state[:] = tamp.sim._root_state[0:1].clone() if call == 0 else root0
            if call == 0:
                dof0, root0 = tamp.sim._dof_state[0:1].clone(), tamp.sim._root_state[0:1].clone()
            tamp.sim.set_dof_state_tensor(tamp.sim._dof_state)
            tamp.sim.set_actor_root_state_tensor(tamp.sim._root_state)
            acts.append(pl.command(tamp.sim._dof_state[0]).cpu().numpy())
        outs[mode] = (np.stack(acts), pl.weights.cpu().numpy().copy())
        if mode == "auto":
            assert pl.probe_result["fused"] is True and pl.probe_result["max_abs_diff"] == 0.0
    np.testing.assert_allclose(outs["fused"][0], outs["step"][0], atol=1e-5)
    np.testing.assert_allclose(outs["fused"][0], outs["auto"][0], atol=1e-5)
    # oracle planner on the same inputs
    sc = P.default_scene()
    cfg = P.make_cfg(K, T, multi_modal=mm, task=task, goal=goal.numpy(), gripper_cmd=2 if task == "pick" else 1)
    opl = P.OraclePandaPlanner(cfg, delta, sc)
    w0 = P.init_world(1)[0]
    ref = np.stack([opl.command(w0) for _ in range(3)])
    np.testing.assert_allclose(outs["fused"][0], ref, atol=1e-3)
    np.testing.assert_allclose(outs["fused"][1], opl.last["w"], atol=1e-3)


@pytest.mark.parametrize("tag", ["panda_opt_rand", "panda_opt_simple", "panda_opt_cov"])
def test_panda_planner_options_match_reference_traces(golden, tag):
    """The Python mirror on the panda_env with sampling_method='random' (noise mean, non-diagonal noise_sigma),
    mppi_mode='simple' (noise_abs_cost, u_scale != 1) and update_cov, against the reference's own planner
    (make_golden.py g11; in-kernel stream = the recorded noise for the same seed)."""
    from tests.test_oracle_panda import PANDA_OPT
    from tests.test_hip_parity_panda import raw31
    import oracle.panda as P
    from m3p2i_aip_amd import _lib as L
    kw = dict(PANDA_OPT[tag])
    K, T = kw.pop("K"), kw.pop("T")
    simple = kw.pop("mode_simple", False)
    delta = golden[f"g9_{tag}_delta"] if f"g9_{tag}_delta" in golden else None
    opt = dict(kw)
    if delta is None:
        opt["sampling_method"] = "random"
    if simple:
        opt["mppi_mode"] = "simple"
    cfg = make_cfg(K, T, fused=True, **opt)
    cfg.mppi.seed_val = 7
    tamp = Tamp(cfg)
    pl = tamp.motion_planner
    if delta is not None:
        pl.set_noise(delta)
    if simple:
        pl.U = torch.zeros(T, 9, device="cuda:0")
    goal = torch.tensor([0.2, 0.2, 1.115, 0, 0, 0, 1.0])
    pl.update_gripper_command("reach")
    tamp.objective.update_objective("reach", goal)
    sim = tamp.sim
    ia, ib = int(sim._get_actor_index_by_name("cubeA")), int(sim._get_actor_index_by_name("cubeB"))
    for call, w in enumerate(golden[f"g9_{tag}_world"]):
        if simple and call:     # (see test_panda_option_traces_vs_reference_golden: every call from the reference's U)
            pl.U = torch.from_numpy(golden[f"g9_{tag}_mean"][call - 1]).to("cuda:0")
        # run_tamp (reactive_tamp.py:45-48): the real world's state of this call into the wrapper's tensors
        dof = torch.zeros(1, 18)
        dof[0, 0::2] = torch.from_numpy(w[P.W_Q:P.W_Q + 9])
        dof[0, 1::2] = torch.from_numpy(w[P.W_Q + 9:P.W_Q + 18])
        root = sim._root_state[0:1].clone().cpu()
        root[0, ia, :] = torch.from_numpy(w[P.W_CUBEA:P.W_CUBEA + 13])
        root[0, ib, :] = torch.from_numpy(w[P.W_CUBEB:P.W_CUBEB + 13])
        root[0, int(sim._get_actor_index_by_name("dyn-obs")), :] = torch.from_numpy(w[P.W_OBS:P.W_OBS + 13])
        sim._dof_state[:] = dof.to("cuda:0")
        sim._root_state[:] = root.to("cuda:0")
        sim.set_dof_state_tensor(sim._dof_state)
        sim.set_actor_root_state_tensor(sim._root_state)
        a = pl.command(sim._dof_state[0])
        np.testing.assert_allclose(a.cpu().numpy(), golden[f"g9_{tag}_action"][call], atol=1e-3, err_msg=f"{tag} {call}")
        np.testing.assert_allclose(pl.weights.cpu().numpy(), golden[f"g9_{tag}_weights"][call], atol=1e-3)
        np.testing.assert_allclose((pl.U if simple else pl.mean_action).cpu().numpy(), golden[f"g9_{tag}_mean"][call], atol=1e-3)
        if f"g9_{tag}_extra" in golden:
            np.testing.assert_allclose(pl.scale_tril.cpu().numpy(), golden[f"g9_{tag}_extra"][call], rtol=1e-4)
    np.testing.assert_allclose(pl.actions.cpu().numpy(), golden[f"g9_{tag}_actions_last"], atol=1e-3)
