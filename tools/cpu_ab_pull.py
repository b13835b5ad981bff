"""CPU A/B study of the PULL scenario's dyn-obs collisions: the reference logged 1 collision in 60 pulls to
(-3, 3) and a task time of 9.97 +- 5.81 s (plot/point/case2_halton_pull_coll.npy cols 17, 18); this build grazes the dyn-obs in
8 of 20 and is done after 2.3 s.  Closed loop on the CPU oracle -- the flow of scripts/sim.py:36-52 + reactive_tamp.py:43-60 with
the reference's shipped planner size (K = 200, T = 15) and its real-world suction skill (skill_utils.py:36-94: suction when the
robot is within 0.6 m of the box and its action points away from it; force kp on the box towards the robot where 1 / d > 1.5,
the opposite force on the robot, clamp +-500; acts during the NEXT step) -- with ONE mechanism toggled at a time:

  spec                  the oracle's default scene: planar spec v1.7 since round 6 (the suction pair consumed by the first substep)
  spec_v16              the scene the table of round 5 was made with: the pair acts in both substeps (fext_substeps = 0);
                        every other variant below is applied on top of THIS one, as in round 5 (profiles/r05/ab_pull*.json)
  kp_200 / kp_100       a weaker suction (the force PhysX transmits through a 1-step force tensor is not pinned)
  suction_ramp          the real-world suction force builds up over 10 ticks of uninterrupted suction instead of at once
  mu_box_1              box-ground friction 1.0 instead of the 0.75 average of (box 0.5, ground 1.0): PhysX combines friction
                        by its material's combine mode, which the reference does not set (isaacgym_wrapper.py:311-326)
  torsion_x10           per-shape torsion friction at the top of the reference's random range and beyond
                        (isaacgym_wrapper.py:318: U(0.001, 0.01) per shape, unseeded): lever x10
  effort_300            drive effort limit 300 N instead of the URDF's 1000 (pointRobot.urdf:35,43)
  avoid                 (not a mechanism: the opt-in extension) the pull cost WITH get_motion_cost, as the logged files' names
                        (case2_halton_pull_coll.npy) suggest those runs had

    python tools/cpu_ab_pull.py [--n 20] [--seed-offset 0] [--json profiles/r05/ab_pull.json] [variant ...]

Reported per variant: successes, task time, episodes in which the dyn-obs felt a contact force above 0.1 N (the test of
get_motion_cost, cost_functions.py:158-169, applied to the real world), and the ratio of the task time to the logged 9.97 s."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle as O                                   # noqa: E402
from m3p2i_aip_amd import sampling                   # noqa: E402  (host-side Halton spline: no GPU needed)

GOAL = (-3.0, 3.0)
DT = 0.05
TIME_LIMIT_TICKS = 764          # 38.2 s: where the reference's pull logs pile up (tests/golden/behaviour_band.json)
LOGGED_TIME_S, LOGGED_HITS = 9.97, "1 / 60"


def variants():
    def mod(**kw):
        def f(sc):
            for k, v in kw.items():
                setattr(sc, k, v)
        return f
    base = dict(kp=400.0, ramp=0, avoid=False)
    v16 = dict(fext_substeps=0)
    return {
        "spec": (mod(), dict(base)),
        "spec_v16": (mod(**v16), dict(base)),
        "kp_200": (mod(**v16), dict(base, kp=200.0)),
        "kp_100": (mod(**v16), dict(base, kp=100.0)),
        "suction_ramp": (mod(**v16), dict(base, ramp=10)),
        "mu_box_1": (mod(box_mu_g=1.0, **v16), dict(base)),
        "torsion_x10": (mod(box_req=1.53, dyn_req=1.53, **v16), dict(base)),
        "effort_300": (mod(drive_fmax=300.0, **v16), dict(base)),
        "avoid": (mod(**v16), dict(base, avoid=True)),
    }


SEED_OFFSET = 0          # --seed-offset: episodes seed_offset .. seed_offset + n - 1 (episodes >= 20 are further draws of the same jitter)


def episode(sc, opt, seed, K=200, T=15):
    # the jitter of tools/band_stats.py's episodes of this scenario (jitter_of("case2_halton_pull_coll", seed): scenario index 0),
    # so that episode e here and episode e of the product's GPU statistics start from the same world
    rng = np.random.default_rng([0, seed])
    phase = 0 if seed == 0 else int(rng.integers(0, 100))
    jb = (0.0, 0.0) if seed == 0 else rng.uniform(-0.05, 0.05, 2)
    jr = (0.0, 0.0) if seed == 0 else rng.uniform(-0.05, 0.05, 2)
    w = O.init_world(1)
    w[0, O.W_B:O.W_B + 2] += jb
    w[0, O.W_R:O.W_R + 2] += jr
    # the dyn-obs where its walk has taken it after `phase` ticks: the same track as an unjittered run, entered later
    # (tools/closed_loop.py does the same for the product's episodes)
    off = sum(0.01 if 25 < (i % 100) < 75 else -0.01 for i in range(phase))
    w[0, O.W_D] += off; w[0, O.W_D + 1] += off
    delta = sampling.halton_spline_delta(K, T, 2)
    cfg = O.make_cfg(K, T, 2, task="pull", goal=GOAL, kp_suction=opt["kp"])
    if opt["avoid"]:
        cfg.avoid_dyn_obs = 1
    pl = O.OraclePointPlanner(cfg, delta, sc)
    hit, hit_ticks, streak = False, 0, 0
    for i in range(TIME_LIMIT_TICKS):
        j = i + phase                           # update_dyn_obs (isaacgym_wrapper.py:205-220): the dyn-obs walks
        d = 0.01 if (25 < j % 100 < 75) else -0.01
        w[0, O.W_D] += d; w[0, O.W_D + 1] += d
        a = pl.command(w[0])[0].astype(np.float32)
        # the real-world suction skill (skill_utils.py:36-94, K = 1: threshold 1.5)
        rb = w[0, O.W_R:O.W_R + 2] - w[0, O.W_B:O.W_B + 2]            # robot - box
        dist = float(np.hypot(*rb))
        if dist < 0.6 and float(a @ rb) > 0.0 and 1.0 / dist > 1.5:
            streak += 1
            kp = opt["kp"] * (min(streak, opt["ramp"]) / opt["ramp"] if opt["ramp"] else 1.0)
            unit = -rb / dist                                           # box - robot, normalised
            fb = np.clip(-kp * unit, -500.0, 500.0)                     # on the box: towards the robot
            w[0, O.W_FEXT_B:O.W_FEXT_B + 2] = fb
            w[0, O.W_FEXT_R:O.W_FEXT_R + 2] = -fb
        else:
            streak = 0
        O.step_batch(sc, w, a[None])
        h = bool(np.abs(w[0, O.W_FC_D:O.W_FC_D + 2]).sum() > 0.1)       # net contact force on the dyn-obs
        hit, hit_ticks = hit or h, hit_ticks + int(h)
        err = float(np.hypot(w[0, O.W_B] - GOAL[0], w[0, O.W_B + 1] - GOAL[1]))
        if err < 0.1:                           # PLANNER_SIMPLE.check_task_success (task_planner.py:24-39)
            return dict(success=True, ticks=i + 1, err=err, hit=hit, hit_ticks=hit_ticks, phase=phase)
    return dict(success=False, ticks=TIME_LIMIT_TICKS, err=err, hit=hit, hit_ticks=hit_ticks, phase=phase)


def main(argv):
    n, out = 20, None
    it = iter(argv)
    for a in it:
        if a == "--n":
            n = int(next(it))
        elif a == "--json":
            out = next(it)
        elif a == "--seed-offset":
            global SEED_OFFSET
            SEED_OFFSET = int(next(it))
    only = [a for a in argv if a in variants()]
    rows = {}
    for name, (modify, opt) in variants().items():
        if only and name not in only:
            continue
        sc = O.default_scene()
        modify(sc)
        eps = [episode(sc, opt, SEED_OFFSET + s) for s in range(n)]
        ok = [e for e in eps if e["success"]]
        t = float(np.mean([e["ticks"] for e in ok]) * DT) if ok else None
        rows[name] = dict(success=len(ok), n=n, episodes=[dict(phase=e["phase"], hit_ticks=e["hit_ticks"], ticks=e["ticks"]) for e in eps], time_s_mean=t, time_s_std=float(np.std([e["ticks"] for e in ok]) * DT) if ok else None,
                          time_ratio_to_logged=(t / LOGGED_TIME_S) if t else None, dyn_obs_hit=sum(e["hit"] for e in eps),
                          hit_ticks_mean=float(np.mean([e["hit_ticks"] for e in eps if e["hit"]])) if any(e["hit"] for e in eps) else 0.0)
        print("%-14s success %2d / %d   time %s s (x%s of the logged %.2f)   dyn-obs hit in %d episodes (%.1f ticks each)" % (
            name, len(ok), n, "%.2f" % t if t else "-", "%.2f" % rows[name]["time_ratio_to_logged"] if t else "-", LOGGED_TIME_S,
            rows[name]["dyn_obs_hit"], rows[name]["hit_ticks_mean"]), flush=True)
    if out:
        os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
        json.dump(dict(scenario="pull to (-3, 3), dyn-obs walking, K = 200, T = 15, halton-spline, CPU oracle closed loop", seed_offset=SEED_OFFSET,
                       logged=dict(dyn_obs_collisions=LOGGED_HITS, task_time_s="9.97 +- 5.81", source="plot/point/case2_halton_pull_coll.npy cols 17, 18"),
                       rows=rows), open(out, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
