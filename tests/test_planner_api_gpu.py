"""GPU tests of the drop-in Python API (M3P2I / Objective / IsaacGymWrapper) wired the way
scripts/reactive_tamp.py:22-73 wires the reference, against golden traces produced by the
reference's own planner (tests/golden/make_golden.py).  FUSED, STEP and probe ('auto') modes
must all give the same numbers."""
from types import SimpleNamespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

G9 = {
    "push": dict(K=256, T=30, task="push", goal=(-1.0, -1.0)),
    "pull": dict(K=256, T=30, task="pull", goal=(0.0, 0.0)),
    "hybrid": dict(K=256, T=30, task="push_pull", goal=(-3.75, -3.75), multi_modal=True),
    # the MPPIConfig switches no shipped config turns on (reference traces: make_golden.py g11)
    "opt_uscale": dict(K=256, T=30, task="push", goal=(-1.0, 3.0), mppi=dict(u_scale=0.5)),
    "opt_cov": dict(K=256, T=30, task="push", goal=(-1.0, 3.0), mppi=dict(update_cov=True)),
    "opt_dead": dict(K=256, T=30, task="push", goal=(-1.0, 3.0), mppi=dict(U_init=[[1.0, -1.0]] * 30, u_init=0.7)),
}


class Tamp:
    """Same wiring as REACTIVE_TAMP.__init__/dynamics/running_cost (reactive_tamp.py:22-73)."""

    def __init__(self, cfg):
        from m3p2i_aip_amd import isaacgym_wrapper as wrapper
        from m3p2i_aip_amd.cost_functions import Objective
        from m3p2i_aip_amd.planner import M3P2I
        self.sim = wrapper.IsaacGymWrapper(cfg.isaacgym, cfg.env_type, num_envs=cfg.mppi.num_samples,
                                           viewer=False, device=cfg.mppi.device, cube_on_shelf=False)
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


def make_cfg(K, T, task, goal, multi_modal=False, fused=None, **mppi_kw):
    from m3p2i_aip_amd.isaacgym_wrapper import IsaacGymConfig
    from m3p2i_aip_amd.planner import MPPIConfig
    kw = dict(num_samples=K, horizon=T, nx=4, mppi_mode="halton-spline", sampling_method="halton",
              device="cuda:0", lambda_=0.5, u_min=[-3.0, -3.0], u_max=[3.0, 3.0],
              noise_sigma=[[3.0, 0.0], [0.0, 3.0]], u_per_command=T, sample_null_action=True,
              filter_u=True, fused=fused)
    kw.update(mppi_kw)
    m = MPPIConfig(**kw)
    return SimpleNamespace(env_type="point_env", multi_modal=multi_modal, suction_active=True,
                           kp_suction=400, pre_height_diff=0.0, task=task, goal=list(goal),
                           cube_on_shelf=False, mppi=m, isaacgym=IsaacGymConfig(dt=0.05))


def world_to_tensors(sim, w31):
    """oracle world row -> the wrapper's [1, ...] tensors, as sim.py would send them."""
    w = np.asarray(w31, np.float32)
    dof = torch.tensor([[w[0], w[4], w[1], w[5]]], device="cuda:0")
    root = sim._root_state[0:1].clone()
    for name, base in (("box", 7), ("dyn-obs", 14)):
        i = int(sim._get_actor_index_by_name(name))
        x, y, c, s, vx, vy, wz = w[base:base + 7]
        th = np.arctan2(s, c)
        root[0, i, 0:2] = torch.tensor([x, y])
        root[0, i, 3:7] = torch.tensor([0, 0, np.sin(th / 2), np.cos(th / 2)], dtype=torch.float32)
        root[0, i, 7:10] = torch.tensor([vx, vy, 0])
        root[0, i, 10:13] = torch.tensor([0, 0, wz])
    return dof, root


@pytest.mark.parametrize("mode", ["fused", "step", "auto"])
@pytest.mark.parametrize("tag", list(G9))
def test_reactive_tamp_wiring_matches_reference_traces(golden, tag, mode):
    kw = dict(G9[tag])
    fused = {"fused": True, "step": False, "auto": None}[mode]
    tamp = Tamp(make_cfg(kw["K"], kw["T"], kw["task"], kw["goal"], kw.get("multi_modal", False), fused,
                         **kw.get("mppi", {})))
    pl = tamp.motion_planner
    pl.set_noise(golden[f"g9_{tag}_delta"])
    tamp.objective.update_objective(kw["task"], list(kw["goal"]))
    worlds = golden[f"g9_{tag}_world"]
    for call in range(worlds.shape[0]):
        dof, root = world_to_tensors(tamp.sim, worlds[call])
        # run_tamp (reactive_tamp.py:45-48)
        tamp.sim._dof_state[:] = dof
        tamp.sim._root_state[:] = root
        tamp.sim.set_dof_state_tensor(tamp.sim._dof_state)
        tamp.sim.set_actor_root_state_tensor(tamp.sim._root_state)
        pref = pl.get_pull_preference()
        action = pl.command(tamp.sim._dof_state[0])
        assert action.shape == (kw["T"], 2) and action.is_cuda
        np.testing.assert_allclose(action.cpu().numpy(), golden[f"g9_{tag}_action"][call], atol=1e-3,
                                   err_msg=f"{tag}/{mode} call {call}")
        np.testing.assert_allclose(pl.weights.cpu().numpy(), golden[f"g9_{tag}_weights"][call], atol=1e-3)
        np.testing.assert_allclose(pl.mean_action.cpu().numpy(), golden[f"g9_{tag}_mean"][call], atol=1e-3)
        assert pl.top_trajs.shape == (20, kw["T"], 2)
        np.testing.assert_allclose(pl.top_trajs[0].cpu().numpy(), golden[f"g9_{tag}_top_trajs"][call][0],
                                   atol=1e-3)
        if kw.get("multi_modal"):
            assert pl.get_pull_preference() == int(golden[f"g9_{tag}_pref"][call])
        if f"g9_{tag}_extra" in golden:      # update_cov: scale_tril / cov_action after the call (mppi.py:514-516)
            np.testing.assert_allclose(pl.scale_tril.cpu().numpy(), golden[f"g9_{tag}_extra"][call], rtol=1e-4)
            np.testing.assert_allclose(pl.cov_action.cpu().numpy(), golden[f"g9_{tag}_extra"][call] ** 2, rtol=2e-4)
    if mode == "auto":
        assert pl.probe_result["fused"] is True, pl.probe_result
        assert pl.probe_result["max_abs_diff"] == 0.0
    assert pl.states.shape == (kw["K"], kw["T"], 4) and pl.actions.shape == (kw["K"], kw["T"], 2)
    da = np.abs(pl.actions.cpu().numpy() - golden[f"g9_{tag}_actions_last"])
    if kw.get("multi_modal"):
        # (the per-mode means are single softmin sums at beta ~ 0.9^15: the worse-conditioned mode of the last call in
        # the median only -- tests/test_oracle_golden.py::test_g9_command_traces says why)
        half = kw["K"] // 2
        lo, hi = sorted([da[:half].max(), da[half:].max()])
        assert lo < 5e-3 and hi < 0.3 and np.median(da) < 0.05
    else:
        assert da.max() < 5e-4


def test_probe_rejects_a_non_standard_plugin(golden):
    """A user cost that is NOT Objective.compute_cost must keep the planner in STEP mode."""
    tamp = Tamp(make_cfg(64, 12, "push", (-1.0, -1.0)))
    pl = tamp.motion_planner

    def my_cost(_):
        return tamp.objective.compute_cost(tamp.sim) + 5.0 * tamp.sim.robot_pos[:, 0].abs()

    pl.running_cost = my_cost
    tamp.objective.update_objective("push", [-1.0, -1.0])
    a = pl.command(tamp.sim._dof_state[0])
    assert pl.probe_result["fused"] is False and pl._fused is False
    assert a.shape == (12, 2)
    b = pl.command(tamp.sim._dof_state[0])
    assert torch.isfinite(b).all()


def test_wrapper_getters_and_step(oracle):
    """step() / getters of the wrapper against the oracle stepping the same controls."""
    from m3p2i_aip_amd.isaacgym_wrapper import IsaacGymConfig, IsaacGymWrapper
    K = 128
    sim = IsaacGymWrapper(IsaacGymConfig(dt=0.05), "point_env", num_envs=K, device="cuda:0")
    assert sim.bodies_per_env == 13 and sim.dofs_per_robot == 2 and sim._root_state.shape == (K, 11, 13)
    rng = np.random.default_rng(3)
    worlds = oracle.init_world(K)
    worlds[:, 0:2] = rng.uniform(-0.6, 0.6, (K, 2)) + np.array([0.0, 1.4])
    sim._dof_state[:, 0] = torch.from_numpy(worlds[:, 0]).cuda()
    sim._dof_state[:, 2] = torch.from_numpy(worlds[:, 1]).cuda()
    sim.set_dof_state_tensor(sim._dof_state)
    sc = oracle.default_scene()
    for it in range(25):
        u = rng.uniform(-3, 3, (K, 2)).astype(np.float32)
        sim.set_dof_velocity_target_tensor(torch.from_numpy(u).cuda())
        sim.step()
        oracle.step_batch(sc, worlds, u)
    np.testing.assert_array_equal(sim.robot_pos.cpu().numpy(), worlds[:, 0:2])
    np.testing.assert_array_equal(sim.robot_vel.cpu().numpy(), worlds[:, 4:6])
    box = sim.get_actor_position_by_name("box").cpu().numpy()
    np.testing.assert_array_equal(box[:, :2], worlds[:, 7:9])
    np.testing.assert_allclose(box[:, 2], 0.05)
    q = sim.get_actor_orientation_by_name("box").cpu().numpy()
    np.testing.assert_allclose(2 * q[:, 2] * q[:, 3], worlds[:, 10], atol=1e-6)   # sin(yaw)
    f = sim.get_actor_contact_forces_by_name("dyn-obs", "box").cpu().numpy()
    np.testing.assert_array_equal(f[:, :2], worlds[:, oracle.W_FC_D:oracle.W_FC_D + 2])
    link = sim.get_actor_link_by_name("point_robot", "link_y").cpu().numpy()
    np.testing.assert_array_equal(link[:, :2], worlds[:, 0:2])
    assert np.abs(worlds[:, 11:13]).max() > 0.1   # the box really got pushed
    # velocity targets stay in force until they are set again (two steps, one set) ...
    u = rng.uniform(-3, 3, (K, 2)).astype(np.float32)
    sim.set_dof_velocity_target_tensor(torch.from_numpy(u).cuda())
    for _ in range(2):
        sim.step()
        oracle.step_batch(sc, worlds, u)
    np.testing.assert_array_equal(sim.robot_pos.cpu().numpy(), worlds[:, 0:2])
    # ... they are COPIED at set time (Isaac Gym's semantics): what happens to the caller's tensor afterwards does not matter
    t = torch.from_numpy(u).cuda()
    sim.set_dof_velocity_target_tensor(t)
    t.mul_(0.0)
    sim.step()
    oracle.step_batch(sc, worlds, u)
    np.testing.assert_array_equal(sim.robot_pos.cpu().numpy(), worlds[:, 0:2])
    # ... unless the caller opts in to the zero-copy hand-over (the closed-loop tools: one launch less per tick), which
    # refuses a tensor that was changed in place before step()
    sim.zero_copy_targets = True
    t = torch.from_numpy(u).cuda()
    sim.set_dof_velocity_target_tensor(t)
    t.mul_(0.5)
    with pytest.raises(RuntimeError, match="modified in place"):
        sim.step()
    sim.step()       # (the refused target is gone: the last accepted one stays in force, no stale pointer)
    oracle.step_batch(sc, worlds, u)
    np.testing.assert_array_equal(sim.robot_pos.cpu().numpy(), worlds[:, 0:2])


def test_update_dyn_obs_walks_the_obstacle_like_the_reference(oracle):
    """isaacgym_wrapper.py:205-220: the dyn-obs of the point_env moves 1 cm per tick along the diagonal, forth while
    period/4 < i % period < 3 period/4, back otherwise; the shift lands in the root_state view AND in the simulated
    world (the next step() starts from it, and a robot standing in its way is pushed by it)."""
    from m3p2i_aip_amd.isaacgym_wrapper import IsaacGymConfig, IsaacGymWrapper
    sim = IsaacGymWrapper(IsaacGymConfig(dt=0.05), "point_env", num_envs=3, device="cuda:0")
    row = int(sim._get_actor_index_by_name("dyn-obs"))
    pos = sim._root_state[0, row, :3].clone()
    zero = torch.zeros(3, 2, device="cuda:0")
    for i in range(130):
        sim.update_dyn_obs(i)
        off = torch.tensor([0.01, 0.01, 0.0], device="cuda:0")
        pos = pos + off if 25 < i % 100 < 75 else pos - off
        if i % 10 == 0:
            sim.set_dof_velocity_target_tensor(zero)
            sim.step()          # the world carries the shifted obstacle through a step (nothing touches it here)
        torch.testing.assert_close(sim._root_state[:, row, :3], pos.expand(3, 3), atol=2e-6, rtol=0)
    torch.testing.assert_close(sim.get_actor_position_by_name("dyn-obs"), pos.expand(3, 3), atol=2e-6, rtol=0)


@pytest.mark.parametrize("mode", ["fused", "step", "auto"])
@pytest.mark.parametrize("tag", ["nav", "opt_abs", "navr", "opt_navr"])
def test_simple_mode_planner_matches_reference_traces(golden, tag, mode):
    """mppi_mode='simple' through the Python mirror (mppi.py:220-233, :335-372): C1, and the same with
    noise_abs_cost, u_scale != 1, a noise mean and a non-diagonal noise_sigma.  The reference trace was recorded
    with its MultivariateNormal replaced by the build's counter-based stream (make_golden.py g9_trace), which is
    what the kernel draws from for the same seed."""
    fused = {"fused": True, "step": False, "auto": None}[mode]
    simple = tag in ("nav", "opt_abs")
    opt = dict(noise_mu=[0.3, -0.2], noise_sigma=[[3.0, 1.0], [1.0, 2.0]]) if tag.startswith("opt_") else {}
    if tag == "opt_abs":
        opt.update(noise_abs_cost=True, u_scale=0.8)
    if simple:     # C1: K = 100, T = 10, mppi_mode 'simple'
        cfg = make_cfg(100, 10, "navigation", (-3.0, 3.0), fused=fused, mppi_mode="simple", sampling_method="random",
                       u_per_command=10, **opt)
    else:          # halton-spline with sampling_method 'random' (quirk Q4: the sample is scaled twice)
        cfg = make_cfg(128, 12, "navigation", (-3.0, 3.0), fused=fused, sampling_method="random", **opt)
    T = cfg.mppi.horizon
    cfg.mppi.seed_val = 7
    torch.manual_seed(3)
    tamp = Tamp(cfg)
    pl = tamp.motion_planner
    if simple:
        # mppi.py:134: U starts as T draws of N(noise_mu, noise_sigma); the trace was recorded from U = 0
        assert pl.U.shape == (T, 2) and float(pl.U.abs().max()) > 0.0
        pl.U = torch.zeros(T, 2, device="cuda:0")
    tamp.objective.update_objective("navigation", [-3.0, 3.0])
    worlds = golden[f"g9_{tag}_world"]
    for call in range(worlds.shape[0]):
        dof, root = world_to_tensors(tamp.sim, worlds[call])
        tamp.sim._dof_state[:] = dof
        tamp.sim._root_state[:] = root
        tamp.sim.set_dof_state_tensor(tamp.sim._dof_state)              # run_tamp, reactive_tamp.py:45-48
        tamp.sim.set_actor_root_state_tensor(tamp.sim._root_state)
        action = pl.command(tamp.sim._dof_state[0])
        np.testing.assert_allclose(action.cpu().numpy(), golden[f"g9_{tag}_action"][call], atol=1e-3)
        np.testing.assert_allclose(pl.weights.cpu().numpy(), golden[f"g9_{tag}_weights"][call], atol=1e-3)
        np.testing.assert_allclose(pl.U.cpu().numpy(), golden[f"g9_{tag}_mean"][call], atol=1e-3)
        if simple:
            # (the reference's cost_total carries + mean_k(S) through an aliasing quirk; the softmin is
            # shift-invariant and the library leaves it out: compare relative to the minimum)
            ct, ref = pl.cost_total.cpu().numpy(), golden[f"g9_{tag}_J"][call]
            np.testing.assert_allclose(ct - ct.min(), ref - ref.min(), rtol=1e-5, atol=2e-3)
    if mode == "auto":
        assert pl.probe_result["fused"] is True, pl.probe_result
    np.testing.assert_allclose(pl.actions.cpu().numpy(), golden[f"g9_{tag}_actions_last"], atol=5e-4)


@pytest.mark.parametrize("ring", [0, 3])
def test_command_returns_a_fresh_tensor_like_the_reference(ring):
    """mppi.py:238-246 returns a new tensor per command(): plans a caller keeps must not be overwritten by later
    calls.  (`action_ring = n` is the opt-in exception: slot call % n of a planner-owned ring.)"""
    cfg = make_cfg(256, 30, "push", (-1.0, -1.0), fused=True, action_ring=ring)
    tamp = Tamp(cfg)
    tamp.objective.update_objective("push", [-1.0, -1.0])
    pl = tamp.motion_planner.attach(tamp.sim, tamp.objective)
    kept, copies = [], []
    for _ in range(6):
        a = pl.command(tamp.sim._dof_state[0])
        kept.append(a)
        copies.append(a.clone())
    torch.cuda.synchronize()
    ptrs = {a.data_ptr() for a in kept}
    if ring == 0:
        assert len(ptrs) == 6
        for a, c in zip(kept, copies):
            assert torch.equal(a, c)
    else:
        assert len(ptrs) == 3
        for a, c in zip(kept[3:], copies[3:]):
            assert torch.equal(a, c)
    assert not torch.equal(copies[0], copies[5])   # (warm start: the plan moves from call to call)
