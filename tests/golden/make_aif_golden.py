"""Records known-answer sequences of the reference's active-inference task planner (pure numpy:
ai_agent.py, adaptive_action_selection.py, isaac_state_action_templates.py) into
tests/golden/aif_golden.json.  Run in the build container only (needs /root/reference):

    python tests/golden/make_aif_golden.py

Cases: (1) the reference's own example (examples/example_aip_panda.py: reach -> pick -> place ->
idle_success -> reach), (2) seeded random observation/preference schedules on every template,
(3) a multi-factor agent list (isBlockAt + isLocFree + isCloseTo) whose actions have
preconditions in other factors.  Each tick stores outcome, action, and the prior D / preferences
C / habits E of every agent after the tick.
"""
import contextlib
import io
import json
import os
import signal
import sys

import numpy as np

sys.path.insert(0, "/root/reference/src")
from m3p2i_aip.planners.task_planner import ai_agent, adaptive_action_selection, parallel_action_selection  # noqa: E402
from m3p2i_aip.planners.task_planner import isaac_state_action_templates as T  # noqa: E402


def _alarm(signum, frame):
    raise TimeoutError


signal.signal(signal.SIGALRM, _alarm)


def snap(agents):
    return [dict(D=a._mdp.D.reshape(-1).tolist(), C=np.asarray(a._mdp.C, float).reshape(-1).tolist(),
                 E=a._mdp.E.reshape(-1).tolist(), u=int(getattr(a, "u", -1))) for a in agents]


def run(templates, schedule):
    agents = [ai_agent.AiAgent(getattr(T, t)()) for t in templates]
    ticks = []
    for prefs, obs in schedule:
        for a, p in zip(agents, prefs):
            if p is not None:
                a.set_preferences(np.array(p, dtype=float).reshape(-1, 1))
        # the reference's selection loop does not terminate when only idle is left after a
        # precondition push (adaptive_action_selection.py:52-58 falls through with
        # looking_for_alternatives set); such a tick ends the recorded sequence
        signal.alarm(2)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                outcome, action = adaptive_action_selection.adapt_act_sel(agents, list(obs))
        except TimeoutError:
            ticks.append(dict(outcome="nonterminating", action=None, agents=[]))
            break
        finally:
            signal.alarm(0)
        ticks.append(dict(outcome=outcome, action=action, agents=snap(agents)))
    return ticks


def run_parallel(templates, schedule):
    """parallel_action_selection.par_act_sel: outcome + the set of parallel plans per tick (the reference builds them
    through Python sets, so their order is not defined: stored sorted)."""
    agents = [ai_agent.AiAgent(getattr(T, t)()) for t in templates]
    ticks = []
    for prefs, obs in schedule:
        for a, p in zip(agents, prefs):
            if p is not None:
                a.set_preferences(np.array(p, dtype=float).reshape(-1, 1))
        signal.alarm(2)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                outcome, plans = parallel_action_selection.par_act_sel(agents, list(obs))
        except TimeoutError:
            ticks.append(dict(outcome="nonterminating", plans=None, agents=[]))
            break
        finally:
            signal.alarm(0)
        ticks.append(dict(outcome=outcome, plans=sorted(sorted(p) for p in plans), agents=snap(agents)))
    return ticks


def example_schedule():
    s = []
    for i in range(25):
        if i < 5: s.append(([[0, 1, 0, 0]], [0]))
        elif i < 10: s.append(([[1, 0, 0, 0]], [1]))
        elif i < 15: s.append(([[1, 0, 0, 0]], [2]))
        elif i < 20: s.append(([[0, 0, 0, 1]], [3]))
        else: s.append(([[0, 1, 0, 0]], [0]))
    return s


def random_schedule(rng, sizes, n):
    s = []
    for _ in range(n):
        prefs, obs = [], []
        for ns in sizes:
            r = rng.random()
            if r < 0.5:
                p = [0.0] * ns
                p[int(rng.integers(ns))] = 1.0
                prefs.append(p)
            elif r < 0.6:
                prefs.append([0.0] * ns)
            else:
                prefs.append(None)       # keep whatever the previous ticks left (incl. pushed prefs)
            obs.append(int(rng.integers(ns)))
        s.append((prefs, obs))
    return s


def main():
    rng = np.random.default_rng(7)
    cases = []
    cases.append(dict(name="example_aip_panda", templates=["MDPIsCubeAtReal"], schedule=example_schedule()))
    for t, ns in (("MDPIsCubeAtReal", 4), ("MDPIsCubeAt", 3), ("MDPIsBlockAt", 2), ("MDPIsLocFree", 2),
                  ("MDPIsCloseTo", 2), ("MDPIsAt", 2)):
        for r in range(4):
            cases.append(dict(name=f"random_{t}_{r}", templates=[t], schedule=random_schedule(rng, [ns], 16)))
    multi = ["MDPIsBlockAt", "MDPIsLocFree", "MDPIsCloseTo"]
    # goal: block at location; observations walk through the precondition chain
    sched = [([[1, 0], [0, 0], [0, 0]], [1, 1, 1]), ([None, None, None], [1, 1, 1]), ([None, None, None], [1, 1, 0]),
             ([None, None, None], [1, 1, 0]), ([None, None, None], [1, 0, 0]), ([None, None, None], [1, 0, 0]),
             ([None, None, None], [0, 0, 0]), ([None, None, None], [0, 0, 0])]
    cases.append(dict(name="multi_chain", templates=multi, schedule=sched))
    for r in range(4):
        cases.append(dict(name=f"multi_random_{r}", templates=multi, schedule=random_schedule(rng, [2, 2, 2], 16)))
    for c in cases:
        c["ticks"] = run(c["templates"], c["schedule"])
    # parallel action selection: the reference's own example (examples/example_aip_parallel.py) + random schedules
    par = ["MDPIsAt", "MDPIsBlockAt", "MDPIsLocFree", "MDPIsCloseTo"]
    ex = [([None, [1.0, 0.0] if i == 0 else None, None, None], ["null", 1, 0, 1] if i < 5 else ["null", 1, 0, 0]) for i in range(15)]
    pcases = [dict(name="parallel_example", templates=par, schedule=ex, parallel=True)]
    for r in range(6):
        sched = random_schedule(rng, [2, 2, 2, 2], 12)
        if r % 2 == 0:      # (as in the example: the first factor unobserved)
            sched = [(p, ["null"] + list(o[1:])) for p, o in sched]
        pcases.append(dict(name=f"parallel_random_{r}", templates=par, schedule=sched, parallel=True))
    for c in pcases:
        c["ticks"] = run_parallel(c["templates"], c["schedule"])
    cases += pcases
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "aif_golden.json")
    json.dump(cases, open(out, "w"))
    print(out, {c["name"]: len(c["ticks"]) for c in cases})


if __name__ == "__main__":
    main()
