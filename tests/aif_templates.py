"""Test helper: the MDP templates by name (product: m3p2i_aip_amd.task_planner.TEMPLATE_TABLE; the reference's
isaac_state_action_templates.py:6-232).  The golden sequences of tests/golden/aif_golden.json were recorded from the
reference on all six templates -- including multi-factor agent lists whose preconditions live in other factors."""
from m3p2i_aip_amd import task_planner as tp


def template(name):
    return getattr(tp, name)()
