"""GPU: the panda_env rollout with fewer samples per 64-wide wavefront (m3_set_rollout_lanes).  Contact response (world
spec v2) made the kernel's cost a matter of DIVERGENCE: a lane in contact runs ~10 000 VALU instructions per substep, a
lane without ~1000, and a wavefront pays the union over its lanes.  K = 4000 samples are 63 wavefronts on a chip with
1024 SIMDs: spreading them over more, narrower wavefronts costs nothing but idle lanes.
    python tools/panda_lanes_sweep.py [--json out.json]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402


class A:   # the arguments bench.run_config reads
    steps, warmup, transport, shard_mix, samples_per_gpu, no_cpu_baseline, no_extras = 100, 10, "rccl", None, None, True, True


def main(argv):
    out = argv[argv.index("--json") + 1] if "--json" in argv else None
    res = {}
    for cname in ("panda", "panda_pick"):
        for lanes in (64, 32, 16, 8, 4):
            os.environ["M3P2I_ROLLOUT_LANES"] = str(lanes)
            r = bench.run_config(cname, A, 1, 0, "cuda:0", None, A.steps, A.warmup, latency=False)
            res[f"{cname}_lanes{lanes}"] = dict(ms_per_step=r["ms_per_step"], rollout_ms=r["rollout_ms"])
            print(cname, "lanes", lanes, "ms/command %.4f rollout %.4f" % (r["ms_per_step"], r["rollout_ms"]), flush=True)
            r["pl"]._engine.close()
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
