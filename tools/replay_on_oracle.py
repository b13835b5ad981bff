"""CPU: replay the 1-env Panda world of a recorded closed-loop episode on the oracle, tick by tick.

    python tools/trace_panda_episode.py 0 gpurun_out/ep0.json mppi.num_samples=200 mppi.horizon=12     # on the GPU box
    python tools/replay_on_oracle.py gpurun_out/ep0.json 90 100 [--rows]                               # anywhere

The record (closed_loop.run(trace=True) -> "full") holds dof_state, root_state and the action of every tick.  The world at
tick t0 is rebuilt from the record (held / sleep state inferred, as the library does when a world is loaded), stepped with
the recorded actions, and compared with the record's next tick: the oracle and the device integrate the same bits, so the
replay IS the episode -- with M3O_DEBUG-style access to every contact row (--rows prints them).  This is how the mechanism
behind the Panda's shipped-size failures was found (docs/NOTEBOOK.md, round 4)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle.panda as P  # noqa: E402


def world_at(sc, rec):
    w = P.init_world(1)[0].copy()
    ds = np.array(rec["dof_state"], np.float32).reshape(9, 2)
    rs = np.array(rec["root_state"], np.float32)
    w[P.W_Q:P.W_Q + 9] = ds[:, 0]
    w[P.W_QD:P.W_QD + 9] = ds[:, 1]
    w[P.W_CUBEA:P.W_CUBEA + 13] = rs[4]          # config_panda's actor order: table, shelf_stand, shelf, obs, cubeA, cubeB, robot base
    w[P.W_CUBEB:P.W_CUBEB + 13] = rs[5]
    return P.infer_state(sc, w[None])[0]


def main(argv):
    rows = "--rows" in argv
    argv = [a for a in argv if a != "--rows"]
    full = json.load(open(argv[0]))["full"]
    t0 = int(argv[1]) if len(argv) > 1 else 0
    t1 = min(int(argv[2]) if len(argv) > 2 else len(full) - 1, len(full) - 1)
    sc = P.default_scene()
    w = world_at(sc, full[t0])[None].copy()
    if rows:
        os.environ["M3O_DEBUG"] = "1"            # oracle/panda_chain.c prints every contact of every substep to stderr
    worst = 0.0
    for i in range(t0, t1):
        u = np.array(full[i]["action"], np.float32)[None]
        c = w[0, P.W_CUBEA:P.W_CUBEA + 13]
        print(f"tick {i} {full[i]['task']:5s} held {int(w[0, P.W_HELD])} fingers {w[0, 7:9].round(4)} cmd {u[0, 7:].round(2)} "
              f"cubeA p {c[:3].round(4)} v {c[7:10].round(2)} w {c[10:13].round(1)}", flush=True)
        P.step_batch(sc, w, u)
        nxt = np.array(full[i + 1]["root_state"], np.float32)[4]
        err = float(np.abs(nxt[:7] - w[0, P.W_CUBEA:P.W_CUBEA + 7]).max())
        worst = max(worst, err)
        print(f"      gripper rows, cube rows {P.last_rows()}   |record - oracle| of cubeA's pose {err:.2e}", flush=True)
    print("largest difference between the record (device) and the replay (oracle):", worst)


if __name__ == "__main__":
    main(sys.argv[1:])
