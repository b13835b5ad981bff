"""N closed-loop episodes per scenario of the reference's recorded experiments, with the start jittered
(deterministic seeds): statistics of what the reference logged per run (plot/plot_point.py:26-34) -- final
block-to-goal error, task time, dyn-obs collisions -- next to the logged statistics (tests/golden/behaviour_band.json).

    python tools/band_stats.py [--n 20] [--json out.json] [scenario ...]

The reference's runs differ from each other through PhysX's own non-determinism, its unseeded per-shape torsion
friction (isaacgym_wrapper.py:318) and the asynchronous RPC loop; this build is deterministic, so the spread is
produced by what plausibly varied there: the phase of the dyn-obs walk at the moment the task starts (0..99 ticks of
its 100-tick period) and +-5 cm on the start positions of box and robot.
tests/test_behaviour_band_gpu.py asserts on the output of `episodes()`.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

SCENARIOS = {
    # scenario of the log -> overrides of config_point (reactive_tamp.py:11-16 command lines); K, T of BASELINE configs
    "case2_halton_push_coll": ["task=push", "goal=[-3,3]", "mppi.num_samples=2000", "mppi.horizon=30"],
    "case2_halton_pull_coll": ["task=pull", "goal=[-3,3]", "mppi.num_samples=2000", "mppi.horizon=30"],
    "corner1_push": ["task=push", "goal=[-3.75,-3.75]", "mppi.num_samples=2000", "mppi.horizon=30"],
    "corner1_pull": ["task=pull", "goal=[-3.75,-3.75]", "mppi.num_samples=2000", "mppi.horizon=30"],
    "corner1_hybrid": ["task=push_pull", "multi_modal=True", "goal=[-3.75,-3.75]", "mppi.num_samples=4000", "mppi.horizon=30"],
}


def jitter_of(scenario, episode):
    """Deterministic per (scenario, episode); episode 0 is the unjittered reference scene."""
    if episode == 0:
        return dict(dyn_phase=0, box=(0.0, 0.0), robot=(0.0, 0.0))
    rng = np.random.default_rng([sorted(SCENARIOS).index(scenario), episode])
    return dict(dyn_phase=int(rng.integers(0, 100)), box=tuple(rng.uniform(-0.05, 0.05, 2).tolist()),
                robot=tuple(rng.uniform(-0.05, 0.05, 2).tolist()))


def stats(x):
    x = np.asarray(x, np.float64)
    return {"mean": float(x.mean()), "std": float(x.std()), "min": float(x.min()), "max": float(x.max()), "n": int(x.size)}


def episodes(scenario, n=20, max_sim_time_s=40.0):
    import closed_loop
    runs = []
    for e in range(n):
        j = jitter_of(scenario, e)
        r = closed_loop.run("config_point", SCENARIOS[scenario], ticks=int(max_sim_time_s / 0.05), jitter=j)
        runs.append(dict(episode=e, jitter=j, success=r["success"], final_pos_error_m=r["final_pos_error"],
                         task_time_s=r["sim_time_s"], dyn_obs_collision_ticks=r["dyn_obs_collision_ticks"],
                         command_ms_p50=r["command_ms_p50"]))
    ok = [r for r in runs if r["success"]]
    return dict(scenario=scenario, n=n, successes=len(ok),
                final_pos_error_m=stats([r["final_pos_error_m"] for r in ok]) if ok else None,
                task_time_s=stats([r["task_time_s"] for r in ok]) if ok else None,
                dyn_obs_collided_episodes=int(sum(r["dyn_obs_collision_ticks"] > 0 for r in runs)),
                dyn_obs_collision_ticks=stats([r["dyn_obs_collision_ticks"] for r in runs]),
                command_ms_p50=stats([r["command_ms_p50"] for r in runs]), runs=runs)


def main(argv):
    n, out, names = 20, None, []
    it = iter(argv)
    for a in it:
        if a == "--n":
            n = int(next(it))
        elif a == "--json":
            out = next(it)
        else:
            names.append(a)
    band = json.load(open(os.path.join(ROOT, "tests", "golden", "behaviour_band.json")))["point"]
    res = {}
    for sc in names or list(SCENARIOS):
        r = episodes(sc, n)
        r["logged"] = {k: band[sc][k] for k in ("final_pos_error_m", "task_time_s", "dyn_obs_collisions")}
        res[sc] = r
        lg = r["logged"]
        print(f"{sc}: {r['successes']}/{n} ok; err {r['final_pos_error_m']['mean']:.3f}+-{r['final_pos_error_m']['std']:.3f} "
              f"(logged {lg['final_pos_error_m']['mean']:.3f}+-{lg['final_pos_error_m']['std']:.3f}); "
              f"time {r['task_time_s']['mean']:.2f}+-{r['task_time_s']['std']:.2f} s "
              f"(logged {lg['task_time_s']['mean']:.2f}+-{lg['task_time_s']['std']:.2f}); "
              f"dyn-obs collided in {r['dyn_obs_collided_episodes']}/{n} episodes "
              f"(logged {lg['dyn_obs_collisions']['mean'] * lg['dyn_obs_collisions']['n']:.0f}/{lg['dyn_obs_collisions']['n']})", flush=True)
    if out:
        os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
