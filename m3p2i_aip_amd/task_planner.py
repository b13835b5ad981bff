"""Task planners of the reactive TAMP loop (host side, numpy).

Mirrors the reference's interface for the caller of the hot path (`reactive_tamp.py:34,76-81`):

* `PLANNER_SIMPLE`            -- task_planner.py:13-39 (fixed task/goal, success test)
* `PLANNER_AIF_PANDA`         -- task_planner.py:41-107 (reach -> pick -> place by active inference)
* `AiAgent`                   -- ai_agent.py:13-193 (discrete active-inference agent, horizon 2)
* `adapt_act_sel`             -- adaptive_action_selection.py:11-84 (action selection with
                                 precondition push-back)
* `MDP`, `MDPIsCubeAtReal`    -- isaac_state_action_templates.py:192-232 (the one template the
                                 planner path uses)
* `MDPIsAt` ... `MDPIsCubeAt` -- isaac_state_action_templates.py:6-190, table-driven (`TEMPLATE_TABLE`): not on
                                 the planner path, built by examples/example_aip_parallel.py
* `PLANNER_PATROLLING`        -- task_planner.py:109-125 (waypoint tour for the navigation task)

This is a restatement, not a transcription: the agent's per-policy loops are evaluated as array
operations over the policy axis and the templates are table-driven.  What is kept exactly is the
observable behaviour the callers and the reference example rely on -- the selected actions and
outcomes, the evolution of the state prior D, and the log-space conventions of C and E (a
preference of 1 is stored as log(1 + 1e-16) == 0.0, "pushed" preconditions as log(2) > 0, an
inhibited habit as log(1e-16)); tests/test_task_planner.py pins all of that to sequences recorded
from the reference's own modules (tests/golden/make_aif_golden.py).
"""
from __future__ import annotations

import numpy as np

_TINY = 1e-16


def _ln(x):
    """log with the reference's floor (ai_agent.py:153-155)."""
    return np.log(np.asarray(x, dtype=np.float64) + _TINY)


def _colnorm(m):
    """Columns scaled to sum 1; an all-zero column becomes uniform (ai_agent.py:157-165)."""
    m = np.array(m, dtype=np.float64)
    s = m.sum(axis=0, keepdims=True)
    uniform = np.full_like(m, 1.0 / m.shape[0])
    return np.where(s > 0, m / np.where(s > 0, s, 1.0), uniform)


def _softmax0(x):
    """exp(x) / sum(exp(x)) along axis 0, no max shift (ai_agent.py:167-172)."""
    e = np.exp(x)
    return e / e.sum(axis=0, keepdims=True)


# ---------------------------------------------------------------------------------------------
# MDP templates (isaac_state_action_templates.py).  Every non-idle action drives the system to
# the FIRST state of its factor (column-stochastic B with a full first row); idle is identity.
# ---------------------------------------------------------------------------------------------
class MDP:
    def __init__(self, state_name, state_names, action_names, preconditions, habits, kappa_d=1.0,
                 prior=0.5):
        n, m = len(state_names), len(action_names)
        self.state_name = state_name
        self.state_names = list(state_names)
        self.action_names = list(action_names)
        self.V = np.arange(m)                       # one-step policies = actions
        self.B = np.zeros((n, n, m))
        self.B[:, :, 0] = np.eye(n)
        self.B[0, :, 1:] = 1.0
        self.preconditions = [list(p) for p in preconditions]
        self.A = np.eye(n)                          # outcomes == states
        self.C = np.zeros((n, 1))                   # preferences over states
        self.D = np.full((n, 1), prior)             # belief about the current state
        self.E = np.asarray(habits, dtype=np.float64).reshape(m, 1)
        self.kappa_d = kappa_d                      # learning rate of D


def MDPIsCubeAtReal():    # :192-232
    return MDP("isCubeAt", ["cube_at_table", "cube_close_to_gripper", "cube_at_pre_place", "cube_at_goal"],
               ["idle", "reach", "pick", "place"],
               [["cube_at_goal"], ["cube_at_table"], ["cube_close_to_gripper"], ["cube_at_pre_place"]],
               [1.0, 1.01, 1.0, 1.0], kappa_d=0.8)


# The templates of the multi-factor examples (isaac_state_action_templates.py:6-190; the planner path itself
# only uses MDPIsCubeAtReal; examples/example_aip_parallel.py builds the first four) as a table:
#   name: (factor, states, actions, preconditions per action, habits E, kappa_d)
TEMPLATE_TABLE = {
    "MDPIsAt": ("isAt", ("at_goal", "not_at_goal"), ("idle", "move_to"), (("none",), ("battery_ok",)), (1.01, 1), 1.0),
    "MDPIsCloseTo": ("isCloseTo", ("close_to", "not_close_to"), ("idle", "approach_obj"), (("none",), ("none",)),
                     (1.01, 1), 1.0),
    "MDPIsLocFree": ("isLocFree", ("loc_free", "not_loc_free"), ("idle", "push_to_non_goal", "pull_to_non_goal"),
                     (("none",), ("close_to",), ("close_to",)), (1.01, 1, 1), 1.0),
    "MDPIsBlockAt": ("isBlockAt", ("block_at_loc", "not_block_at_loc"), ("idle", "push_to_goal", "pull_to_goal"),
                     (("none",), ("loc_free", "close_to"), ("loc_free", "close_to")), (1.01, 1, 1), 1.0),
    "MDPIsCubeAt": ("isCubeAt", ("cube_at_table", "cube_at_hand", "cube_at_goal"), ("idle", "pick", "place"),
                    (("cube_at_goal",), ("cube_at_table",), ("cube_at_hand",)), (1.0, 1.01, 1.0), 0.8),
}


def _template(name):
    def make():
        factor, states, actions, pre, habits, kappa = TEMPLATE_TABLE[name]
        return MDP(factor, states, actions, pre, habits, kappa_d=kappa)
    make.__name__ = make.__qualname__ = name
    make.__doc__ = f"{name}() -> MDP (TEMPLATE_TABLE[{name!r}])"
    return make


MDPIsAt, MDPIsCloseTo, MDPIsLocFree = _template("MDPIsAt"), _template("MDPIsCloseTo"), _template("MDPIsLocFree")
MDPIsBlockAt, MDPIsCubeAt = _template("MDPIsBlockAt"), _template("MDPIsCubeAt")


# ---------------------------------------------------------------------------------------------
class AiAgent(object):
    """Discrete active-inference agent over one state factor, look-ahead of one step
    (ai_agent.py:13-193).  Attribute names the reference's callers touch are kept: `_mdp`
    (with C, D, E in the same log/probability spaces), `u`, `F`, `G`, `post_x`."""

    t_horizon = 2

    def __init__(self, mdp):
        import copy
        self._mdp = copy.deepcopy(mdp)
        m = self._mdp
        self.n_policies = int(np.shape(m.V)[0])
        self.n_states = int(np.shape(m.B)[0])
        self.n_actions = int(np.shape(m.B)[2])
        self.n_outcomes = self.n_states
        self.policy_indexes_v = np.asarray(m.V)
        m.D = _colnorm(m.D) if hasattr(m, "D") else _colnorm(np.ones((self.n_states, 1)))
        m.C = _ln(m.C)
        m.E = _ln(_colnorm(m.E))
        self.default_E = m.E.copy()
        self.likelihood_A = _colnorm(m.A)
        fwd = np.stack([_colnorm(m.B[:, :, a]) for a in range(self.n_actions)], axis=2)
        self.fwd_trans_B = fwd
        self.bwd_trans_B = np.transpose(fwd, (1, 0, 2))
        self.F = np.zeros((self.n_policies, 1))
        self.G = np.zeros((self.n_policies, 1))
        self.post_x = np.full((self.n_states, self.t_horizon, self.n_policies), 1.0 / self.n_states)
        self.u = 0
        # ambiguity term diag(A' ln A): zero for the identity likelihood of every template
        self._ambiguity = np.diagonal(self.likelihood_A.T @ _ln(self.likelihood_A))

    # -- perception: posterior over states for every policy + variational free energy ----------
    def infer_states(self, obs):
        n, P = self.n_states, self.n_policies
        A = self.likelihood_A
        Bf = self.fwd_trans_B[:, :, self.policy_indexes_v]      # [n, n, P]
        Bb = self.bwd_trans_B[:, :, self.policy_indexes_v]
        D = self._mdp.D.reshape(n)
        post = np.full((n, 2, P), 1.0 / n)
        post[:, 0, :] = D[:, None]
        # tau = 0: the observation is given; the future message comes from the still uniform
        # belief about tau = 1
        lnA0 = _ln(A[:, int(obs)])[:, None]
        past0 = _ln(D)[:, None]
        fut0 = _ln(np.einsum("ijp,jp->ip", Bb, post[:, 1, :]))
        s0 = _softmax0(past0 + fut0 + lnA0)
        F = np.einsum("ip,ip->p", s0, _ln(s0) - past0 - lnA0)
        post[:, 0, :] = s0
        # tau = 1: the outcome is the most likely one under the tau = 0 posterior; no future message
        o1 = np.argmax(A @ s0, axis=0)                          # [P]
        lnA1 = _ln(A[:, o1])
        past1 = _ln(np.einsum("ijp,jp->ip", Bf, s0))
        s1 = _softmax0(past1 + lnA1)
        F = F + np.einsum("ip,ip->p", s1, _ln(s1) - past1 - lnA1)
        post[:, 1, :] = s1
        self.post_x = post
        self.F = F.reshape(P, 1)
        return self.F, self.post_x

    # -- action: expected free energy, policy posterior, prior update ---------------------------
    def infer_policies(self):
        n, P = self.n_states, self.n_policies
        Bf = self.fwd_trans_B[:, :, self.policy_indexes_v]
        C = self._mdp.C.reshape(n)
        # predicted outcome of each policy from the CURRENT-state posterior (ai_agent.py:120)
        o = np.argmax(np.einsum("ijp,jp->ip", Bf, self.post_x[:, 0, :]), axis=0)
        onehot = np.zeros((n, P))
        onehot[o, np.arange(P)] = 1.0
        risk = np.einsum("ip,ip->p", _ln(onehot) - C[:, None], onehot)
        G = risk + self._ambiguity @ self.post_x[:, 1, :]
        self.G = G.reshape(P, 1)
        post_pi = _softmax0(self._mdp.E - self.F - self.G)
        self.u = int(np.argmax(_softmax0(_ln(post_pi))))
        # Bayesian model average over policies, then the prior for the next call
        self.post_x_bma = np.einsum("itp,p->it", self.post_x, post_pi.reshape(P))
        D = _colnorm(self._mdp.D + self._mdp.kappa_d * self.post_x_bma[:, 0].reshape(n, 1))
        D[D < 0.00001] = 0.0
        self._mdp.D = _colnorm(D)
        return self.G, self.u

    # -- accessors used by the action selection and the task planner ----------------------------
    def set_observation(self, obs):
        self._mdp.o = obs

    def set_preferences(self, pref, index="none"):
        if isinstance(index, str) and index == "none":
            self._mdp.C = _ln(pref)
        else:
            self._mdp.C[index] = _ln(pref)

    def get_action(self):
        return self.u

    def get_current_state(self):
        return self._mdp.D

    def reset_habits(self, index="none"):
        if isinstance(index, str) and index == "none":
            self._mdp.E = self.default_E.copy()
        else:
            self._mdp.E[index] = _ln(0)

    def reset_current_state(self):
        self._mdp.D = _colnorm(np.ones((self.n_states, 1)))


MAX_SELECTION_ROUNDS = 200


def adapt_act_sel(agent, obs, verbose=False):
    """One tick of the adaptive action selection (adaptive_action_selection.py:11-84): returns
    (outcome, action) with outcome in {'success', 'running', 'failure'}.

    Order of events kept from the reference: habits restored; pushed preconditions that are now
    observed are dropped; an observed state whose preference is exactly 1 (log == 0) is immediate
    success; otherwise infer -> pick -> check the action's preconditions against the believed
    states of ALL factors, pushing a high-priority preference (2) on every missing one and
    inhibiting the action until one is executable or only idle remains.

    One deliberate difference: when a precondition has been pushed and afterwards only idle is
    left, the reference's loop never terminates (adaptive_action_selection.py:52-58 falls through
    with looking_for_alternatives set; recorded as "nonterminating" in the golden file).  Here
    that situation ends after MAX_SELECTION_ROUNDS rounds with ('failure', 'idle_fail')."""
    agents = agent if isinstance(agent, list) else [agent]
    obs = obs if isinstance(agent, list) else [obs]
    n = len(agents)
    for ag, ob in zip(agents, obs):
        ag.reset_habits()
        for idx in range(len(ag._mdp.C)):
            if ag._mdp.C[idx] > 0 and idx == ob:
                if verbose:
                    print("removed preference state", idx)
                ag.set_preferences(0, idx)
    for ag, ob in zip(agents, obs):
        for idx in range(len(ag._mdp.C)):
            if ag._mdp.C[idx] == 0 and idx == ob:
                return "success", "idle_success"

    u = [-1] * n
    believed = ["null"] * n
    searching = False
    for _round in range(MAX_SELECTION_ROUNDS):
        for i, (ag, ob) in enumerate(zip(agents, obs)):
            if isinstance(ob, str) and ob == "null":
                continue
            if not searching:
                ag.infer_states(ob)
            _, u[i] = ag.infer_policies()
            believed[i] = ag._mdp.state_names[int(np.argmax(ag.get_current_state()))]
        if max(u) == 0:
            # only idle left: a failure the first time round, otherwise keep adapting exactly
            # like the reference does (its loop re-enters with the habits still inhibited)
            if not searching:
                if verbose:
                    print("No action found for this situation")
                return "failure", "idle_fail"
            continue
        for i, ag in enumerate(agents):
            if u[i] <= 0:
                continue
            unmet = False
            for need in ag._mdp.preconditions[u[i]]:
                if need != "none" and need not in believed:
                    unmet = True
                    searching = True
                    for other in agents:
                        if need in other._mdp.state_names:
                            other.set_preferences(2, other._mdp.state_names.index(need))
                    ag.reset_habits(u[i])
            if not unmet:
                return "running", ag._mdp.action_names[u[i]]
    return "failure", "idle_fail"


def par_act_sel(agent, obs, verbose=False):
    """One tick of the PARALLEL action selection (parallel_action_selection.py:12-106): instead of stopping at the first
    executable action, every executable action is recorded and inhibited and the selection continues until only idle is
    left; returns (outcome, plans) -- each plan a list of action names of DIFFERENT state factors that can run side by
    side.  The reference assembles the plans through Python sets (their order is undefined there); here every plan is
    sorted and the list of plans is sorted.  Same prologue as adapt_act_sel (habits restored, met pushed preconditions
    dropped, an observed state with preference exactly 1 is success: ('success', [])), same deliberate difference (the
    reference does not terminate when a push leaves only idle and nothing was found: ('failure', []) after
    MAX_SELECTION_ROUNDS rounds)."""
    agents = agent if isinstance(agent, list) else [agent]
    obs = obs if isinstance(agent, list) else [obs]
    n = len(agents)
    for ag, ob in zip(agents, obs):
        ag.reset_habits()
        for idx in range(len(ag._mdp.C)):
            if ag._mdp.C[idx] > 0 and idx == ob:
                if verbose:
                    print("removed preference state", idx)
                ag.set_preferences(0, idx)
    if any(ag._mdp.C[idx] == 0 and idx == ob for ag, ob in zip(agents, obs) for idx in range(len(ag._mdp.C))):
        return "success", []

    found = []                      # (action name, factor) in the order they became executable
    u = [-1] * n
    believed = ["null"] * n
    searching = False
    outcome = "failure"
    for _round in range(MAX_SELECTION_ROUNDS):
        for i, (ag, ob) in enumerate(zip(agents, obs)):
            if isinstance(ob, str) and ob == "null":
                continue
            if not searching:
                ag.infer_states(ob)
            _, u[i] = ag.infer_policies()
            believed[i] = ag._mdp.state_names[int(np.argmax(ag.get_current_state()))]
        if max(u) == 0:             # only idle left
            if found:
                break
            if not searching:
                if verbose:
                    print("No action found for this situation")
                return "failure", []
            continue                # (the reference spins here; MAX_SELECTION_ROUNDS ends it)
        for i, ag in enumerate(agents):
            if u[i] <= 0:
                continue
            missing = [need for need in ag._mdp.preconditions[u[i]] if need != "none" and need not in believed]
            for need in missing:
                searching = True
                for other in agents:
                    if need in other._mdp.state_names:
                        other.set_preferences(2, other._mdp.state_names.index(need))
            ag.reset_habits(u[i])   # inhibited either way: not executable, or taken -- look for alternatives
            if not missing:
                found.append((ag._mdp.action_names[u[i]], i))
                outcome = "running"
    else:
        return "failure", []
    # one plan per found action: that action + every found action of another factor not yet represented in the plan
    plans = set()
    for name, factor in found:
        names, factors = {name}, {factor}
        for other, f in found:
            if f not in factors:
                names.add(other)
                factors.add(f)
        plans.add(tuple(sorted(names)))
    return outcome, sorted(list(p) for p in plans)


# ---------------------------------------------------------------------------------------------
def _rot_cols(q):
    """Columns (x, y, z axes) of the rotation matrix of an (x, y, z, w) quaternion
    (skill_utils.py:140-180)."""
    x, y, z, w = (float(v) for v in q)
    return np.array([[2 * (w * w + x * x) - 1, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 2 * (w * w + y * y) - 1, 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 2 * (w * w + z * z) - 1]])


def general_ori_cube2goal(cube_quat, goal_quat):
    """Flip-invariant orientation distance (skill_utils.py:224-252): for the goal's x and y axes,
    1 - |cos| to the closest cube axis, summed."""
    Rc, Rg = _rot_cols(cube_quat), _rot_cols(goal_quat)
    cos = np.abs(Rg[:, :2].T @ Rc)            # [goal axis x|y, cube axis x|y|z]
    return float(np.sum(1.0 - cos.max(axis=1)))


def _as_goal_tensor(goal, device):
    import torch
    return goal if torch.is_tensor(goal) else torch.tensor(goal, device=device)


class PLANNER_SIMPLE:
    """The point_env "task planner" (behaves like task_planner.py:13-39): task and goal are the configured ones
    for the whole run; all it decides is whether the task is done -- robot (navigation) or box (push / pull /
    push_pull) within SUCCESS_RADIUS of the goal."""

    SUCCESS_RADIUS = 0.1

    def __init__(self, cfg) -> None:
        self.device, self.task = cfg.mppi.device, cfg.task
        self.curr_goal = _as_goal_tensor(cfg.goal, self.device)
        self.dist_threshold = self.SUCCESS_RADIUS

    def update_plan(self, sim):
        """Nothing to re-plan: a single fixed task."""

    def reset_plan(self):
        """(kept for the interface: nothing to reset)"""

    def check_task_success(self, sim):
        import torch
        if self.task == "navigation":
            return torch.norm(sim.robot_pos[0, :] - self.curr_goal) < self.dist_threshold
        if self.task in ("push", "pull", "push_pull"):
            return torch.norm(sim.get_actor_position_by_name("box")[0, :2] - self.curr_goal) <= self.dist_threshold
        return False


class PLANNER_AIF_PANDA(PLANNER_SIMPLE):
    """Pick-and-place task planner of the panda_env (behaves like task_planner.py:41-107).

    Each tick maps the scene to one of three observations of the `isCubeAt` factor and lets the
    active-inference agent pick the skill; what the motion planner receives is `(task, curr_goal)`.

    The progress of the manipulation is kept as ONE monotone stage (0 cube on the table, 1 cube
    within reach of the gripper, 2 cube above its goal): a stage, once reached, is never left --
    the reference expresses the same with two sticky flags, exposed here as the read-only
    `pick_always` / `place_always`."""

    STAGE_PREFERENCE = ((0, 1, 0, 0),   # stage 0: want "cube_close_to_gripper"  -> reach
                        (1, 0, 0, 0),   # stage 1: want "cube_at_table" (template order) -> pick
                        (1, 0, 0, 0))   # stage 2: same preference, observation 2        -> place
    PLACE_TOLERANCE = 0.03              # xy offset to the pre-place pose + orientation distance
    SUCCESS_RADIUS = 0.04               # xy distance cube <-> goal while placing

    def __init__(self, cfg) -> None:
        import torch
        self.device = cfg.mppi.device
        self.task = self.curr_action = "idle"
        self.curr_goal = torch.zeros(7, device=self.device)
        self.ai_agent_task = [AiAgent(MDPIsCubeAtReal())]
        self.stage = 0
        self._ee_state = self._ee_host = self._pre_host = None
        self._aif_memo = {}
        self.pre_pick_place_threshold = cfg.pre_height_diff + 0.005
        self.verbose = False

    obs = property(lambda self: self.stage)
    pick_always = property(lambda self: self.stage >= 1)
    place_always = property(lambda self: self.stage == 2)

    @staticmethod
    def _pose(sim, actor, link):
        return sim.get_actor_link_by_name(actor, link)[0, :7]

    def _scene_stage(self, cube, goal, ee, pre_place):
        """Stage the scene alone would justify (no memory)."""
        cube, goal, ee, pre_place = (v.detach().cpu().numpy().astype(np.float64) for v in (cube, goal, ee, pre_place))
        return self._scene_stage_host(cube, goal, ee, pre_place)

    def _scene_stage_host(self, cube, goal, ee, pre_place):
        misplacement = np.linalg.norm(pre_place[:2] - cube[:2]) + general_ori_cube2goal(goal[3:7], cube[3:7])
        gripper_gap = np.linalg.norm(ee[:3] - cube[:3])
        if self.verbose:
            print("gripper_gap", gripper_gap, "misplacement", misplacement)
        if misplacement < self.PLACE_TOLERANCE:
            return 2
        return 1 if gripper_gap < self.pre_pick_place_threshold else 0

    # the gripper's centre as a device tensor (the reference's attribute): built on demand from the host copy on the fast path
    @property
    def ee_state(self):
        if self._ee_state is None and self._ee_host is not None:
            self._ee_state = self._device_tensor(self._ee_host)
        return self._ee_state

    @ee_state.setter
    def ee_state(self, v):
        self._ee_state, self._ee_host = v, None

    def _device_tensor(self, host):
        """f32 host array -> tensor on the planner's device that carries its host values (`_m3_host`): Objective.goal_list
        hands those to the C-ABI without reading the tensor back (cost_functions.py)."""
        import torch
        t = torch.from_numpy(np.array(host, dtype=np.float32)).to(self.device)
        t._m3_host = [float(x) for x in host]
        t._m3_host_version = t._version      # (an in-place write afterwards invalidates the attached values: cost_functions.py)
        return t

    def _poses(self, sim):
        """(cubeA, cubeB, left finger, right finger) poses of env 0 as f32 host rows -- ONE device-to-host copy per tick on the
        HIP wrapper (env0_link_states_host) -- or None for any other sim (four device tensors, read one by one below)."""
        host = getattr(sim, "env0_link_states_host", None)
        if host is None:
            return None
        rb = host()
        return [rb[sim.link_row(a, l), :7] for a, l in (("cubeA", "box"), ("cubeB", "box"), ("panda", "panda_leftfinger"),
                                                        ("panda", "panda_rightfinger"))]

    def update_plan(self, sim):
        sim.step()   # the reference refreshes the link states with one step of the rollout envs
        rows = self._poses(sim)
        if rows is None:
            cube, goal = self._pose(sim, "cubeA", "box"), self._pose(sim, "cubeB", "box")
            self.ee_state = 0.5 * (self._pose(sim, "panda", "panda_leftfinger") + self._pose(sim, "panda", "panda_rightfinger"))
            self.pre_place_loc = goal.clone()
            self.pre_place_loc[2] += self.pre_pick_place_threshold
            scene = self._scene_stage(cube, goal, self.ee_state, self.pre_place_loc)
        else:
            # the same f32 operations on the host rows (0.5 * (l + r); z + threshold), the same f64 decision
            cube, goal, left, right = rows
            self._ee_state, self._ee_host = None, np.float32(0.5) * (left + right)
            pre = goal.copy()
            pre[2] = pre[2] + np.float32(self.pre_pick_place_threshold)
            if self._pre_host is None or not np.array_equal(pre, self._pre_host):   # (cubeB asleep: the same tensor tick after tick)
                self._pre_host, self.pre_place_loc = pre, self._device_tensor(pre)
            scene = self._scene_stage_host(*(v.astype(np.float64) for v in (cube, goal, self._ee_host, pre)))
        self.stage = max(self.stage, scene)
        _, self.curr_action = self._select_action()
        self.task = self.curr_action
        if self.task == "pick":
            self.curr_goal = self.pre_place_loc

    _AGENT_STATE = ("F", "G", "post_x", "post_x_bma")
    _MDP_STATE = ("C", "D", "E")

    def _select_action(self):
        """One tick of the agent.  A tick is a pure function of (stage, the agent's belief D): the preferences are set from the
        stage, the habits are restored first thing (adapt_act_sel), the observation is the stage.  D is clipped at 1e-5 and
        renormalised every tick (ai_agent.py:140-147), so within a stage it reaches an exact fixed point after ~15 ticks --
        from then on the tick (~200 us of small numpy calls) is a table look-up that puts the same arrays back."""
        agent = self.ai_agent_task[0]
        key = None if self.verbose else (self.stage, agent._mdp.D.tobytes())
        hit = self._aif_memo.get(key) if key is not None else None
        if hit is not None:
            out, mdp_state, agent_state, agent.u = hit
            for k, v in zip(self._MDP_STATE, mdp_state):
                setattr(agent._mdp, k, v.copy())
            for k, v in zip(self._AGENT_STATE, agent_state):
                setattr(agent, k, v.copy())
            return out
        agent.set_preferences(np.array(self.STAGE_PREFERENCE[self.stage], dtype=float).reshape(-1, 1))
        out = adapt_act_sel(self.ai_agent_task, [self.stage], verbose=self.verbose)
        if key is not None and len(self._aif_memo) < 256 and all(hasattr(agent, k) for k in self._AGENT_STATE):
            self._aif_memo[key] = (out, [np.array(getattr(agent._mdp, k)) for k in self._MDP_STATE],
                                   [np.array(getattr(agent, k)) for k in self._AGENT_STATE], agent.u)
        return out

    def check_task_success(self, sim):
        if self.task != "place":
            return False
        from .cost_functions import attached_host_values
        rows, goal = self._poses(sim), attached_host_values(self.curr_goal)
        if rows is not None and goal is None:         # (a goal written to in place since it was made: its device values count)
            goal = self.curr_goal.detach().reshape(-1).cpu().tolist()
        if rows is not None:                          # (the tick's host copy of the link states: no further read-back)
            offset = np.asarray(goal[:2], np.float32) - rows[0][:2]
            return bool((offset * offset).sum(dtype=np.float32) < np.float32(self.SUCCESS_RADIUS ** 2))
        offset = self.curr_goal[:2] - self._pose(sim, "cubeA", "box")[:2]
        return bool((offset * offset).sum() < self.SUCCESS_RADIUS ** 2)


class PLANNER_PATROLLING(PLANNER_SIMPLE):
    """Navigation through a closed tour of waypoints (behaves like task_planner.py:109-125): `curr_goal` is the
    waypoint being approached; reaching it (within SUCCESS_RADIUS) selects the next, the last wraps to the first.
    Never "done": check_task_success of a patrol is False."""

    def __init__(self, goals, device="cuda:0") -> None:
        import torch
        self.task = "navigation"
        self.device = device
        self.goals = torch.as_tensor(goals, dtype=torch.float32, device=device).reshape(-1, 2)
        self.dist_threshold = self.SUCCESS_RADIUS
        self.reset_plan()

    def reset_plan(self):
        self.goal_id = 0

    @property
    def curr_goal(self):
        return self.goals[self.goal_id]

    def update_plan(self, robot_pos, stay_still=False):
        import torch
        if bool(torch.norm(robot_pos.reshape(-1)[:2].to(self.goals) - self.curr_goal) < self.dist_threshold):
            self.goal_id = (self.goal_id + 1) % self.goals.shape[0]

    def check_task_success(self, sim):
        return False


def set_task_planner(cfg):
    """task_planner.py:7-11."""
    return PLANNER_SIMPLE(cfg) if cfg.env_type == "point_env" else PLANNER_AIF_PANDA(cfg)
