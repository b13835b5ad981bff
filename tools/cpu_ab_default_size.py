"""A/B table for the default-size push: the reference's logged push to (-3, 3) succeeds 60 / 60, this
build's planar spec at the reference's SHIPPED planner size (K = 200, T = 15, config/mppi/point.yaml) wedges the box
past the goal in most episodes.  This tool runs that scenario in CLOSED LOOP ON THE CPU ORACLE (test infrastructure:
oracle.OraclePointPlanner + one oracle world as the "real world", the flow of scripts/sim.py:36-52) with ONE mechanism
of the contact model toggled at a time, N jittered episodes each, and prints / writes the table.  The mechanisms are
things PhysX has and the written spec lacks or fixes differently -- not a sweep of material constants:

  spec_v14                  the planar spec as shipped up to round 3 (v1.4): independent linear and torsion friction rows
  spec                      the shipped spec (v1.5 = v1.4 + the coupling `contensou` below, adopted because of this table)
  depen_10                  max depenetration velocity 10 m/s instead of 2 (PhysX's default is far above 2)
  no_speculative            contacts only once penetrating (contact_offset 0): no speculative rows
  patch_corners             box-ground friction torque with the lever of a four-corner contact patch (0.283 m: PhysX
                            resolves the box's ground contact at its corners) instead of the disc-equivalent 0.153 m
  patch_none                no torsional ground friction at all
  patch_4pt                 ground friction AT the four corners of the patch, each opposing its own velocity (the
                            experimental rows of oracle/planar_world.c, scene.friction_coupling = 2): turning resistance that fades
                            with the sliding speed, as a real patch's does
  (spec = spec_v14 + the two rows' limits COUPLED by the sliding-spinning law of a contact patch -- Contensou; Zhuravlev's
   Pade form, factors from the substep's initial velocities: the cheap form of patch_4pt.  Every other row toggles its
   mechanism on top of spec_v14.)
  passes_12 / passes_3      12 / 3 solver passes instead of 6 (how hard the drive row wins against the contact rows)
  drive_soft                drive damping 150 instead of 600: a drive that builds its force over several substeps
  horizon_30                (control, not a mechanism) the same planner with T = 30: the known cure

    python tools/cpu_ab_default_size.py [--n 20] [--json profiles/r04/ab_default_size_push.json]
"""
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
TIME_LIMIT_TICKS = 364          # 18.2 s: where the reference's logs pile up (tests/golden/behaviour_band.json)


def variants():
    def mod(**kw):
        def f(sc):
            for k, v in kw.items():
                setattr(sc, k, v)
        return f
    return {
        "spec_v14": (mod(friction_coupling=0), 15),
        "spec": (mod(), 15),
        "depen_10": (mod(friction_coupling=0, max_bias=10.0), 15),
        "no_speculative": (mod(friction_coupling=0, contact_offset=0.0), 15),
        "patch_corners": (mod(friction_coupling=0, box_req=0.2828, dyn_req=0.2828), 15),
        "patch_none": (mod(friction_coupling=0, box_req=0.0, dyn_req=0.0), 15),
        "patch_4pt": (mod(friction_coupling=2), 15),
        "passes_12": (mod(friction_coupling=0, iters=12), 15),
        "passes_3": (mod(friction_coupling=0, iters=3), 15),
        "drive_soft": (mod(friction_coupling=0, drive_damping=150.0), 15),
        "horizon_30": (mod(friction_coupling=0), 30),
    }


def episode(sc, T, seed, K=200):
    rng = np.random.default_rng([7, seed])
    phase = 0 if seed == 0 else int(rng.integers(0, 100))
    jb = (0.0, 0.0) if seed == 0 else rng.uniform(-0.05, 0.05, 2)
    jr = (0.0, 0.0) if seed == 0 else rng.uniform(-0.05, 0.05, 2)
    w = O.init_world(1)
    w[0, O.W_B:O.W_B + 2] += jb
    w[0, O.W_R:O.W_R + 2] += jr
    delta = sampling.halton_spline_delta(K, T, 2)
    cfg = O.make_cfg(K, T, 2, task="push", goal=GOAL)
    pl = O.OraclePointPlanner(cfg, delta, sc)
    hit = False
    for i in range(TIME_LIMIT_TICKS):
        j = i + phase                           # update_dyn_obs (isaacgym_wrapper.py:205-220): the dyn-obs walks
        d = 0.01 if (25 < j % 100 < 75) else -0.01
        w[0, O.W_D] += d; w[0, O.W_D + 1] += d
        a = pl.command(w[0])
        O.step_batch(sc, w, a[0:1].astype(np.float32))
        hit = hit or bool(np.abs(w[0, 29:31]).sum() > 0.1)          # net contact force on the dyn-obs
        err = float(np.hypot(w[0, O.W_B] - GOAL[0], w[0, O.W_B + 1] - GOAL[1]))
        if err < 0.1:                           # PLANNER_SIMPLE.check_task_success (task_planner.py:24-39)
            return dict(success=True, ticks=i + 1, err=err, hit=hit)
    return dict(success=False, ticks=TIME_LIMIT_TICKS, err=err, hit=hit)


def main(argv):
    n, out = 20, None
    it = iter(argv)
    for a in it:
        if a == "--n":
            n = int(next(it))
        elif a == "--json":
            out = next(it)
    rows = {}
    only = [a for a in argv if not a.startswith("--") and not a.isdigit() and not a.endswith(".json")]
    for name, (modify, T) in variants().items():
        if only and name not in only:
            continue
        sc = O.default_scene()
        modify(sc)
        eps = [episode(sc, T, s) for s in range(n)]
        ok = [e for e in eps if e["success"]]
        rows[name] = dict(success=len(ok), n=n, T=T,
                          time_s_mean=float(np.mean([e["ticks"] for e in ok]) * DT) if ok else None,
                          final_err_failed_mean=float(np.mean([e["err"] for e in eps if not e["success"]])) if len(ok) < n else None,
                          dyn_obs_hit=sum(e["hit"] for e in eps))
        print("%-16s T=%2d  success %2d / %d   time %s s   failed-episode error %s m   dyn-obs hit %d" % (
            name, T, len(ok), n, "%.2f" % rows[name]["time_s_mean"] if ok else "-",
            "%.2f" % rows[name]["final_err_failed_mean"] if rows[name]["final_err_failed_mean"] is not None else "-",
            rows[name]["dyn_obs_hit"]), flush=True)
    if out:
        os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
        json.dump(dict(scenario="push to (-3, 3), K = 200, halton-spline, CPU oracle closed loop", logged="60 / 60 (tests/golden/behaviour_band.json)",
                       rows=rows), open(out, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
