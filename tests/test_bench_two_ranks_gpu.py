"""bench.py's N > 1 path (torch.distributed.run, sample sharding, collectives, max-over-ranks timing,
rank-0 JSON line) exercised with two ranks on the ONE GPU of the test box: M3_BENCH_SHARE_GPU=1 puts
both ranks on cuda:0 and uses gloo (RCCL refuses two ranks per device).  Only the plumbing is
checked -- the driver produces the real multi-GPU numbers."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("config,mode,extra", [(None, "single-mode", []), ("c5", "multi-modal", []),
                                               ("c5", "multi-modal", ["--transport", "p2p", "--shard-mix", "3"])])
def test_bench_with_two_ranks(config, mode, extra):
    """config None: the default of an N > 1 run = the N = 1 headline's own per-GPU workload (push, single-mode),
    weak-scaled; c5 = BASELINE configs[4] (push_pull, multi-modal) as the headline; the same through the device-side
    exchange (hipIpc between the two processes) with the two-exchange protocol."""
    env = dict(os.environ, M3_BENCH_SHARE_GPU="1")
    port = 29800 + os.getpid() % 150
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20",
           "--warmup", "3", "--samples-per-gpu", "512"] + (["--config", config] if config else []) + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]           # exactly one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 3 and d["scaling"] == "weak"
    assert "K=1024 (512/GPU)" in d["config"]["workload"] and mode in d["config"]["workload"]
    p3 = "--shard-mix" in extra
    assert d["config"]["name"] == (config or "push")
    assert ("TWO small exchanges" if p3 else "ONE collective per command") in d["config"]["parallelism"]
    assert ("p2p device-side exchange" in d["config"]["parallelism"]) == ("p2p" in extra)
    assert d["collective_ms"]["per_command"] == (2.0 if p3 else 1.0) and d["collective_ms"]["total"] > 0
    assert d["value"] > 0 and abs(d["value"] - 1024 * 30 * 20 / (d["ms_per_step"] * 20e-3)) < 1e-6 * d["value"]
    assert "cpu_baseline" not in d and d["roofline"]["bound"] == "hbm"
    # the other sharded workloads ride along, every row with its own `scaling`
    want = {"c5", "push_saturating"} if config is None else {"push_weak", "push_saturating"}
    assert set(d["other_configs"]) == want
    for w in d["other_configs"].values():
        assert w["n_gpus"] == 2 and w["scaling"] == "weak" and w["collective_ms"]["per_command"] == 1.0 and w["value"] > 0   # (single-mode rows)
        assert "K=1024" in w["workload"]          # (test mode: every row at the test's size)


def test_bench_single_gpu_line_contract():
    """`python bench.py` (N = 1): ONE JSON line with the driver's keys, the BASELINE configs[1] workload as the
    headline, `roofline` and `cpu_baseline` objects, the other configs and the closed-loop figure riding along."""
    # (the cpu_baseline object's sample shortened from 21 s to 3: its content is checked, not its precision)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "5", "--cpu-baseline-seconds", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 5 and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "K=2000" in d["config"]["workload"] and "T=30" in d["config"]["workload"] and "task=push" in d["config"]["workload"]
    assert abs(d["value"] - 2000 * 30 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and rf["bytes_per_launch"] == 36 * 2000 * 30
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0 < rf["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb and cb["reference_shaped"]["value"] > 0
    assert set(d["other_configs"]) == {"northstar", "hybrid", "panda", "panda_settled", "panda_reach_mid", "panda_pick", "c5shard",
                                       "c5_unsharded", "worst_case_scene", "c1", "reference_default_size"}
    assert all(v["value"] > 0 and v["scaling"] == "weak" and 0 < v["roofline"]["frac"] < 1
               for v in d["other_configs"].values()), d["other_configs"]
    oc = d["other_configs"]
    assert "K=64000" in oc["c5_unsharded"]["workload"] and "multi-modal" in oc["c5_unsharded"]["workload"]
    assert oc["panda_reach_mid"]["lanes_per_sample_used"] == 16 and oc["panda_reach_mid"]["near_share_permille"] >= 260
    assert "task=pick" in oc["panda_pick"]["workload"] and oc["panda_pick"]["roofline"]["bytes_per_launch"] == 92 * 4000 * 20
    # the corner scene is the slow end of the same kernel: >= the initial scene's time
    assert oc["worst_case_scene"]["kernel_ms"]["rollout"] > d["kernel_ms"]["rollout"]
    # (a contract test, not a timing test: under `pytest -n 6` other tests' kernels share the GPU with this bench run, and a
    # 40-command region of 5 ms has been seen to take 40 -- the closed loop is checked for what it did, not for how long it took)
    assert d["closed_loop"]["ms_per_step"] > 0 and d["closed_loop"]["ticks"] == 200 and d["closed_loop"]["final_pos_error_m"] < 0.5


def test_bench_starts_its_own_ranks():
    """The driver's command line with N > 1 and NO launcher around it -- exactly `python3 bench.py --gpus 2 --steps 3
    --warmup 1` -- : bench.py launches the two ranks itself (torch.distributed.run on a free loopback port), rank 0's
    ONE JSON line comes through on stdout, the exit code is 0 only if every rank finished and the communicator the
    ranks formed really has two members."""
    env = dict(os.environ, M3_BENCH_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert "K=4000 (2000/GPU)" in d["config"]["workload"] and d["value"] > 0
    assert set(d["other_configs"]) == {"c5", "push_saturating"}
