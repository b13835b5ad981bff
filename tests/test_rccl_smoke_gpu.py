"""RCCL on the one GPU of the test box (world_size 1): distributed.py's collectives on
library-owned buffers (tools/rccl_smoke.py).  The multi-GPU runs themselves are the driver's."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_accepts_library_owned_buffers():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_smoke.py")], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and "rccl smoke ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
