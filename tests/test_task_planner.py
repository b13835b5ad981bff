"""Task planners against sequences recorded from the reference's
own modules (tests/golden/make_aif_golden.py -> aif_golden.json): selected actions and outcomes
must be identical, the agents' D / C / E must agree to 1e-9 after every tick."""
import json
import os

import numpy as np
import pytest

from m3p2i_aip_amd import task_planner as tp
from tests.aif_templates import template

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "aif_golden.json")))


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_action_selection_matches_reference_sequences(case):
    agents = [tp.AiAgent(template(t)) for t in case["templates"]]
    for tick, (prefs, obs) in zip(case["ticks"], case["schedule"]):
        for a, p in zip(agents, prefs):
            if p is not None:
                a.set_preferences(np.array(p, dtype=float).reshape(-1, 1))
        if case.get("parallel"):            # parallel_action_selection.par_act_sel: a set of parallel plans per tick
            outcome, plans = tp.par_act_sel(agents, list(obs))
            if tick["outcome"] == "nonterminating":
                assert (outcome, plans) == ("failure", [])
                break
            assert outcome == tick["outcome"] and sorted(sorted(p) for p in plans) == tick["plans"], (outcome, plans, tick)
        else:
            outcome, action = tp.adapt_act_sel(agents, list(obs))
            if tick["outcome"] == "nonterminating":   # the reference hangs here; this build gives up
                assert (outcome, action) == ("failure", "idle_fail")
                break
            assert (outcome, action) == (tick["outcome"], tick["action"])
        for a, g in zip(agents, tick["agents"]):
            np.testing.assert_allclose(a._mdp.D.reshape(-1), g["D"], atol=1e-9)
            np.testing.assert_allclose(np.asarray(a._mdp.C, float).reshape(-1), g["C"], atol=1e-9)
            np.testing.assert_allclose(a._mdp.E.reshape(-1), g["E"], atol=1e-9)


def test_reference_example_walks_reach_pick_place_success():
    """examples/example_aip_panda.py: reach -> pick -> place -> idle_success -> reach."""
    case = next(c for c in CASES if c["name"] == "example_aip_panda")
    agent = [tp.AiAgent(tp.MDPIsCubeAtReal())]
    acts = []
    for prefs, obs in case["schedule"]:
        agent[0].set_preferences(np.array(prefs[0], dtype=float).reshape(-1, 1))
        acts.append(tp.adapt_act_sel(agent, list(obs))[1])
    assert acts[4] == "reach" and acts[9] == "pick" and acts[14] == "place"
    assert acts[19] == "idle_success" and acts[24] == "reach"


def test_single_agent_call_form_and_null_observation():
    a = tp.AiAgent(template("MDPIsCloseTo"))
    a.set_preferences(np.array([[1.0], [0.0]]))
    assert tp.adapt_act_sel(a, 1) == ("running", "approach_obj")      # non-list form
    assert tp.adapt_act_sel(a, 0) == ("success", "idle_success")
    b = [tp.AiAgent(template("MDPIsCloseTo")), tp.AiAgent(template("MDPIsAt"))]
    b[0].set_preferences(np.array([[1.0], [0.0]]))
    assert tp.adapt_act_sel(b, [1, "null"]) == ("running", "approach_obj")


def test_orientation_distance_is_flip_invariant():
    ident = np.array([0.0, 0.0, 0.0, 1.0])
    s = np.sqrt(0.5)
    for q in ([0, 0, s, s], [0, 0, 1, 0], [s, 0, 0, s], [0, s, 0, s], [1, 0, 0, 0]):   # 90/180 deg flips
        assert tp.general_ori_cube2goal(np.array(q, float), ident) < 1e-12
    q30 = np.array([0, 0, np.sin(np.pi / 12), np.cos(np.pi / 12)])
    assert abs(tp.general_ori_cube2goal(q30, ident) - 2 * (1 - np.cos(np.pi / 6))) < 1e-12


class _FakeSim:
    """get_actor_link_by_name / step of the wrapper, with poses set by the test."""

    def __init__(self):
        import torch
        self.torch = torch
        self.poses = {}
        self.steps = 0

    def set(self, actor, link, pos, quat=(0, 0, 0, 1)):
        self.poses[(actor, link)] = self.torch.tensor([list(pos) + list(quat) + [0.0] * 6])

    def get_actor_link_by_name(self, actor, link):
        return self.poses[(actor, link)]

    def step(self):
        self.steps += 1


def test_planner_aif_panda_reach_pick_place_and_latches():
    """task_planner.py:41-107: observation thresholds, latches and the (task, goal) it hands to the
    motion planner."""
    import types
    torch = pytest.importorskip("torch")
    cfg = types.SimpleNamespace(mppi=types.SimpleNamespace(device="cpu"), pre_height_diff=0.05, env_type="panda_env")
    pl = tp.set_task_planner(cfg)
    assert isinstance(pl, tp.PLANNER_AIF_PANDA) and pl.task == "idle"
    sim = _FakeSim()
    sim.set("cubeA", "box", (0.2, -0.2, 1.06))
    sim.set("cubeB", "box", (0.2, 0.2, 1.06))
    sim.set("panda", "panda_leftfinger", (0.0, 0.04, 1.5))
    sim.set("panda", "panda_rightfinger", (0.0, -0.04, 1.5))
    pl.update_plan(sim)
    assert sim.steps == 1 and pl.obs == 0 and pl.task == "reach"
    # gripper within pre_height_diff + 0.005 of the cube -> pick, goal = pre-place pose above cubeB
    sim.set("panda", "panda_leftfinger", (0.2, -0.16, 1.10))
    sim.set("panda", "panda_rightfinger", (0.2, -0.24, 1.10))
    pl.update_plan(sim)
    assert pl.obs == 1 and pl.task == "pick" and pl.pick_always
    np.testing.assert_allclose(pl.curr_goal[:3].numpy(), [0.2, 0.2, 1.06 + 0.055], atol=1e-6)
    # moving the gripper away again does not un-latch the pick
    sim.set("panda", "panda_leftfinger", (0.0, 0.04, 1.5))
    sim.set("panda", "panda_rightfinger", (0.0, -0.04, 1.5))
    pl.update_plan(sim)
    assert pl.obs == 1 and pl.task == "pick"
    assert not pl.check_task_success(sim)
    # cube above the goal (xy) and aligned -> place; success once within 4 cm in xy
    sim.set("cubeA", "box", (0.2, 0.19, 1.115))
    pl.update_plan(sim)
    assert pl.obs == 2 and pl.task == "place" and pl.place_always
    assert pl.check_task_success(sim)


def test_planner_simple_success_thresholds():
    import types
    torch = pytest.importorskip("torch")
    cfg = types.SimpleNamespace(mppi=types.SimpleNamespace(device="cpu"), task="push", goal=[-1.0, -1.0],
                                env_type="point_env")
    pl = tp.set_task_planner(cfg)
    sim = types.SimpleNamespace(robot_pos=torch.tensor([[0.0, 0.0]]),
                                get_actor_position_by_name=lambda n: torch.tensor([[-1.0, -0.92, 0.0]]))
    assert bool(pl.check_task_success(sim))
    sim.get_actor_position_by_name = lambda n: torch.tensor([[-1.0, -0.85, 0.0]])
    assert not bool(pl.check_task_success(sim))


def test_patrolling_planner_and_template_exports():
    """The drop-in surface of m3p2i_aip.planners.task_planner.* (ADVICE r2): the five table-driven templates and
    PLANNER_PATROLLING (task_planner.py:109-125: next waypoint within 0.1 m, wrapping) are exported by compat."""
    import torch
    from m3p2i_aip_amd import compat
    compat.install(force_standins=True)
    from m3p2i_aip.planners.task_planner import isaac_state_action_templates as T
    from m3p2i_aip.planners.task_planner.task_planner import PLANNER_PATROLLING
    for name in ("MDPIsAt", "MDPIsCloseTo", "MDPIsLocFree", "MDPIsBlockAt", "MDPIsCubeAt", "MDPIsCubeAtReal"):
        m = getattr(T, name)()
        assert m.B.shape == (len(m.state_names), len(m.state_names), len(m.action_names)) and m.action_names[0] == "idle"
    p = PLANNER_PATROLLING([[1.0, 0.0], [1.0, 1.0], [0.0, 1.0]], device="cpu")
    assert p.task == "navigation" and p.curr_goal.tolist() == [1.0, 0.0]
    p.update_plan(torch.tensor([0.5, 0.0]), False)
    assert p.goal_id == 0
    for want, pos in ((1, [0.95, 0.0]), (2, [1.0, 0.92]), (0, [0.05, 1.0])):
        p.update_plan(torch.tensor(pos), False)
        assert p.goal_id == want and p.curr_goal.tolist() == p.goals[want].tolist()
    p.goal_id = 2
    p.reset_plan()
    assert p.goal_id == 0 and p.check_task_success(None) is False


def test_planner_aif_panda_memoised_ticks_are_the_agents_own():
    """PLANNER_AIF_PANDA looks a tick up once (stage, belief D) has been seen before -- D reaches an exact fixed point within a
    stage.  Same decisions and the same agent state (C, D, E, F, G, posteriors, u), tick by tick, as an agent that runs every
    tick; and the look-ups do happen."""
    import types
    pytest.importorskip("torch")
    cfg = types.SimpleNamespace(mppi=types.SimpleNamespace(device="cpu"), pre_height_diff=0.05, env_type="panda_env")
    memo, plain = tp.set_task_planner(cfg), tp.set_task_planner(cfg)
    hits = 0
    for stage, ticks in ((0, 40), (1, 30), (2, 12)):
        for _ in range(ticks):
            memo.stage = plain.stage = stage
            known = (stage, memo.ai_agent_task[0]._mdp.D.tobytes()) in memo._aif_memo
            hits += known
            got = memo._select_action()
            agent = plain.ai_agent_task[0]
            agent.set_preferences(np.array(plain.STAGE_PREFERENCE[stage], dtype=float).reshape(-1, 1))
            want = tp.adapt_act_sel(plain.ai_agent_task, [stage])
            assert got == want
            a, b = memo.ai_agent_task[0], agent
            assert a.u == b.u
            for k in ("C", "D", "E"):
                np.testing.assert_array_equal(getattr(a._mdp, k), getattr(b._mdp, k))
            for k in ("F", "G", "post_x", "post_x_bma"):
                np.testing.assert_array_equal(getattr(a, k), getattr(b, k))
    assert hits > 40


@pytest.mark.gpu
def test_planner_aif_panda_host_path_equals_the_per_link_path_gpu():
    """On the HIP wrapper PLANNER_AIF_PANDA reads env 0's link states with ONE device-to-host copy per tick
    (IsaacGymWrapper.env0_link_states_host) and hands its goal over with the host values attached; on any other sim it reads four
    link tensors one by one (task_planner.py:62-107 does).  Same f32 operations either way: a closed loop of 90 ticks run twice --
    once with the wrapper's host copy hidden from the planner -- hands out the same tasks, the same goals bit for bit and ends in
    the same world; the host copy is made once per world state and follows step() / the state uploads."""
    import torch
    from m3p2i_aip_amd import compat
    compat.install(force_standins=True)
    from m3p2i_aip.planners.motion_planner import m3p2i
    from m3p2i_aip.planners.task_planner import task_planner
    import m3p2i_aip.utils.isaacgym_utils.isaacgym_wrapper as wrapper
    from m3p2i_aip.planners.motion_planner.cost_functions import Objective

    class Hidden:
        """the wrapper without its host-copy entry points: what a sim that is not ours looks like to the planner"""
        def __init__(self, sim):
            self._sim = sim

        def __getattr__(self, k):
            if k in ("env0_link_states_host", "link_row"):
                raise AttributeError(k)
            return getattr(self._sim, k)

    def episode(hide):
        cfg = compat.make_config("config_panda", ["mppi.num_samples=512"])
        sim = wrapper.IsaacGymWrapper(cfg.isaacgym, cfg.env_type, num_envs=cfg.mppi.num_samples, viewer=False,
                                      device=cfg.mppi.device, cube_on_shelf=cfg.cube_on_shelf)
        real = wrapper.IsaacGymWrapper(cfg.isaacgym, cfg.env_type, num_envs=1, device=cfg.mppi.device, cube_on_shelf=cfg.cube_on_shelf)
        obj, tpl = Objective(cfg), task_planner.set_task_planner(cfg)

        def dynamics(_, u, t=None):
            sim.set_dof_velocity_target_tensor(u)
            sim.step()
            return torch.stack([sim.robot_pos[:, 0], sim.robot_vel[:, 0], sim.robot_pos[:, 1], sim.robot_vel[:, 1]], dim=1), u
        mp = m3p2i.M3P2I(cfg, dynamics=dynamics, running_cost=lambda _: obj.compute_cost(sim))
        mp.attach(sim, obj)
        seen = Hidden(sim) if hide else sim
        log = []
        for _ in range(90):
            sim._dof_state[:] = real._dof_state
            sim._root_state[:] = real._root_state
            sim.set_dof_state_tensor(sim._dof_state)
            sim.set_actor_root_state_tensor(sim._root_state)
            tpl.update_plan(seen)
            mp.update_gripper_command(tpl.task)
            obj.update_objective(tpl.task, tpl.curr_goal)
            done = bool(tpl.check_task_success(seen))
            log.append((tpl.task, tpl.stage, tuple(tpl.curr_goal.float().cpu().tolist()), tuple(tpl.ee_state.float().cpu().tolist()),
                        tuple(obj.goal_list()), done))
            if done:
                break
            a = mp.command(sim._dof_state[0])[0]
            real.set_dof_velocity_target_tensor(a.view(1, 9))
            real.step()
        return log, real._dof_state.cpu().clone(), real._root_state.cpu().clone(), sim

    fast, dof_f, root_f, sim = episode(hide=False)
    slow, dof_s, root_s, _ = episode(hide=True)
    assert fast == slow
    assert torch.equal(dof_f, dof_s) and torch.equal(root_f, root_s)
    assert {"reach", "pick"} <= {t[0] for t in fast}
    # one copy per world state
    a = sim.env0_link_states_host()
    assert sim.env0_link_states_host() is a
    np.testing.assert_array_equal(a[sim.link_row("cubeA", "box"), :7], sim.get_actor_link_by_name("cubeA", "box")[0, :7].cpu().numpy())
    sim.step()
    b = sim.env0_link_states_host()
    assert b is not a
    sim.set_dof_state_tensor(sim._dof_state)
    assert sim.env0_link_states_host() is not b
