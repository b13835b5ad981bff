"""Test data: the state / action tables of the reference's MDP templates that the planner path does
NOT use (isaac_state_action_templates.py:6-190; dead code in the reference).
The golden sequences of tests/golden/aif_golden.json were recorded from the reference on all six
templates -- including multi-factor agent lists whose preconditions live in other factors -- so the
tests build these from the product's generic `MDP` class to keep that coverage of `AiAgent` /
`adapt_act_sel`."""
from m3p2i_aip_amd.task_planner import MDP, MDPIsCubeAtReal  # noqa: F401

_TABLES = {
    # name: (factor, states, actions, preconditions per action, habits, kappa_d)
    "MDPIsAt": ("isAt", ["at_goal", "not_at_goal"], ["idle", "move_to"], [["none"], ["battery_ok"]], [1.01, 1], 1.0),
    "MDPIsCloseTo": ("isCloseTo", ["close_to", "not_close_to"], ["idle", "approach_obj"], [["none"], ["none"]],
                     [1.01, 1], 1.0),
    "MDPIsLocFree": ("isLocFree", ["loc_free", "not_loc_free"], ["idle", "push_to_non_goal", "pull_to_non_goal"],
                     [["none"], ["close_to"], ["close_to"]], [1.01, 1, 1], 1.0),
    "MDPIsBlockAt": ("isBlockAt", ["block_at_loc", "not_block_at_loc"], ["idle", "push_to_goal", "pull_to_goal"],
                     [["none"], ["loc_free", "close_to"], ["loc_free", "close_to"]], [1.01, 1, 1], 1.0),
    "MDPIsCubeAt": ("isCubeAt", ["cube_at_table", "cube_at_hand", "cube_at_goal"], ["idle", "pick", "place"],
                    [["cube_at_goal"], ["cube_at_table"], ["cube_at_hand"]], [1.0, 1.01, 1.0], 0.8),
}


def template(name):
    if name == "MDPIsCubeAtReal":
        return MDPIsCubeAtReal()
    factor, states, actions, pre, habits, kappa = _TABLES[name]
    return MDP(factor, states, actions, pre, habits, kappa_d=kappa)
