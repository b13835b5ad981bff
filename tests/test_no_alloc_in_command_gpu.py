"""SURVEY.md 8(b), Ownership row: "No allocation in m3_command".  The whole process runs under an LD_PRELOAD interposer
(tests/native/alloc_count_shim.c) that counts every hipMalloc / hipHostMalloc / hipExtMallocWithFlags / hipMallocAsync /
hipFree ... call of ANY library in it; after the first m3_set_objective + first command of a handle the counter must not
move over further commands -- for every task family, including the reach command whose record buffer and report word were
allocated lazily inside m3_rollout until round 5."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROG = r"""
import ctypes, json, sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
shim = ctypes.CDLL(%(shim)r)
shim.m3shim_alloc_calls.restype = ctypes.c_long
from m3p2i_aip_amd.engine import HipEngine, make_config
out = {}
g = torch.Generator().manual_seed(5)

def noise(K, T, nu):
    knots = torch.randn(K, nu, max(T // 4, 2), generator=g)
    return torch.nn.functional.interpolate(knots, size=T, mode="linear", align_corners=True).permute(0, 2, 1).contiguous().numpy()

def run(name, eng, objectives, n=6):
    host = np.zeros((eng.cfg.T, eng.cfg.nu), np.float32)
    first = True
    counts = []
    for task, goal, kw in objectives:
        eng.set_objective(task, goal, **kw)
        if first:
            eng.command(sync_host=True)       # (the first command of the handle: everything lazy has happened by now)
            first = False
        torch.cuda.synchronize()
        before = shim.m3shim_alloc_calls()
        for _ in range(n):
            eng.lib.m3_command(eng._h, host.ctypes.data)      # the C-ABI entry itself, synchronous form
        torch.cuda.synchronize()
        counts.append(shim.m3shim_alloc_calls() - before)
    out[name] = counts
    eng.close()

pk = dict(u_min=[-3, -3], u_max=[3, 3], noise_sigma_diag=[3, 3])
e = HipEngine(make_config(K=2000, T=30, nu=2, **pk)); e.set_noise(noise(2000, 30, 2))
run("push", e, [("push", (-1.0, -1.0), {}), ("pull", (1.0, 1.0), {}), ("navigation", (2.0, -2.0), {})])
e = HipEngine(make_config(K=4000, T=30, nu=2, multi_modal=True, **pk)); e.set_noise(noise(4000, 30, 2))
run("push_pull", e, [("push_pull", (-3.75, -3.75), {})])
e = HipEngine(make_config(K=20000, T=30, nu=2, multi_modal=True, **pk)); e.set_noise(noise(20000, 30, 2))
run("push_pull_three_launch", e, [("push_pull", (-3.75, -3.75), {})])
pa = dict(env_type="panda_env", u_min=[-1.2] * 9, u_max=[1.2] * 9, noise_sigma_diag=[10.0] * 7 + [0.8, 0.8], lambda_=0.05,
          pre_height_diff=0.05, dt=0.01)
e = HipEngine(make_config(K=4000, T=20, nu=9, **pa)); e.set_noise(noise(4000, 20, 9))
# pick first, THEN reach: the reach-only buffers must exist before the first reach command, not be made by it
run("panda", e, [("pick", [0.2, 0.2, 1.115, 0, 0, 0, 1], dict(gripper_cmd=1)),
                 ("reach", [0.2, 0.2, 1.115, 0, 0, 0, 1], dict(gripper_cmd=1)),
                 ("place", [0.3, -0.2, 1.115, 0, 0, 0, 1], dict(gripper_cmd=-1))])
assert shim.m3shim_alloc_calls() > 0, "the interposer saw no allocation at all: it is not in front of the HIP runtime"
print("RESULT" + json.dumps(out))
"""


def test_m3_command_allocates_nothing_after_the_first_set_objective(tmp_path):
    shim = str(tmp_path / "liballocshim.so")
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", os.path.join(ROOT, "tests", "native", "alloc_count_shim.c"),
                           "-o", shim, "-ldl"])
    env = dict(os.environ, LD_PRELOAD=shim + (":" + os.environ["LD_PRELOAD"] if os.environ.get("LD_PRELOAD") else ""))
    r = subprocess.run([sys.executable, "-c", PROG % dict(root=ROOT, shim=shim)], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][-1][6:])
    for name, counts in out.items():
        assert all(c == 0 for c in counts), f"{name}: allocation calls during m3_command per objective: {counts}"
