"""bench.py --gpus N started WITHOUT a launcher (the shape of the driver's N = 1 command line) starts its own ranks.
On this CPU-only box the ranks refuse to run (no CPU fallback for the product path); what can be checked here is the
plumbing around them: N processes are launched, their failure becomes bench.py's exit code, nothing that looks like a
result line is printed.  The GPU-side twin (tests/test_bench_two_ranks_gpu.py) checks the line itself."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-side plumbing test")
def test_self_launch_propagates_rank_failure():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "needs an MI355X" in r.stderr and "2-rank launch failed" in r.stderr


def test_launcher_and_flag_must_agree():
    """under a launcher, --gpus must equal WORLD_SIZE (a 1-rank launch of `--gpus 2` must not print an n_gpus = 1 line)"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
