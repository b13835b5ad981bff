"""GPU: one jittered Panda closed-loop episode of tools/band_stats.py with the replayable per-tick record (closed_loop.run(trace=True)):
    python tools/trace_panda_episode.py <episode> <out.json> [overrides ...]
(the record holds dof_state, root_state and the action of every tick: the 1-env world can be replayed on the CPU oracle)"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import closed_loop

e = int(sys.argv[1])
out = sys.argv[2]
ov = sys.argv[3:] or ["mppi.num_samples=4000", "mppi.horizon=20"]
rng = np.random.default_rng([77, e])
j = dict(cube=(0.0, 0.0) if e == 0 else tuple(rng.uniform(-0.02, 0.02, 2).tolist()))
r = closed_loop.run("config_panda", list(ov), ticks=int(os.environ.get("TICKS", "200")), jitter=j, trace=True)
json.dump(r, open(out, "w"))
print({k: v for k, v in r.items() if k not in ("trace", "full")})
