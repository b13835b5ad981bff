"""The "reference-shaped" CPU baseline of bench.py (oracle/refshaped.py: per-t loop of per-op torch
tensors around a batched simulator step) computes the same command() as the fused C port it is
reported next to -- otherwise its time would be the time of something else."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")


@pytest.mark.parametrize("task,goal,mm", [("push", (-1.0, -1.0), False), ("pull", (0.0, 0.0), False),
                                          ("push_pull", (-3.75, -3.75), True)])
def test_reference_shaped_loop_equals_the_port(oracle, golden, task, goal, mm):
    from oracle import refshaped
    K, T = 256, 30
    delta = golden["g9_push_delta"]
    ref = oracle.OraclePointPlanner(oracle.make_cfg(K, T, 2, task=task, goal=goal, multi_modal=mm), delta)
    shaped = refshaped.RefShapedPointPlanner(task, goal, mm, K, T, delta)
    w0 = oracle.init_world(1)[0]
    w0[0:2] = (0.0, 1.5)            # inside the suction range of the box
    for call in range(3):
        a = ref.command(w0)
        b = shaped.command(w0)
        np.testing.assert_allclose(b, a, atol=1e-3, err_msg=f"call {call}")
        # the leading trajectories (far down the list the weights are ~1e-7 apart and torch's exp and
        # the port's expf may order two neighbours differently)
        np.testing.assert_allclose(shaped.top_trajs.numpy()[:6], ref.last["top_trajs"][:6], atol=1e-4)
