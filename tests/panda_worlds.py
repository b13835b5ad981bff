"""Test data helpers: panda_env worlds built with the oracle (shared by the GPU parity tests and by
tests/golden/make_golden.py, so that the golden traces and the tests start from the same worlds)."""
import numpy as np


def grasp_world(P, sc, close_gripper=True, lift=0.0, offset=(0.0, 0.0)):
    """A world in which the gripper holds cubeA (built with the oracle: IK + closing); with
    close_gripper=False the open gripper is left around the cube (`lift` metres above the grasp
    pose, displaced by `offset` = (dx, dy) in world coordinates: the hand's y axis is world y, its x axis
    world -x), so that rollouts grasp -- or just miss -- it on their own."""
    w = P.init_world(1)
    for _ in range(30):
        P.step_batch(sc, w, np.zeros((1, 9), np.float32))
    target = w[0, P.W_CUBEA:P.W_CUBEA + 3] + np.array([offset[0], offset[1], sc.grasp_z + lift])
    q = np.array([0, 0.3, 0, -2.2, 0, 2.5, 0.785, 0.04, 0.04], np.float32)

    def feat(L):
        return np.concatenate([L["pos"][8], 0.3 * L["az"][8], 0.3 * L["ay"][8]])

    want = np.concatenate([target, 0.3 * np.array([0, 0, -1.0]), 0.3 * np.array([0, 1.0, 0])])
    for _ in range(600):
        L = P.fk(sc, q)
        e = want - feat(L)
        if np.linalg.norm(e) < 1e-4:
            break
        Jm = np.zeros((9, 7))
        for j in range(7):
            dq = q.copy(); dq[j] += 1e-3
            Jm[:, j] = (feat(P.fk(sc, dq)) - feat(L)) / 1e-3
        q[:7] += (np.linalg.pinv(Jm, rcond=1e-3) @ e * 0.5).astype(np.float32)
        q[:7] = np.clip(q[:7], np.array(sc.qlo)[:7], np.array(sc.qhi)[:7])
    w[0, P.W_Q:P.W_Q + 9] = q
    w[0, P.W_QD:P.W_QD + 9] = 0
    if not close_gripper:
        return w[0].copy()
    close = np.zeros((1, 9), np.float32); close[0, 7:] = -1.5
    for _ in range(40):
        P.step_batch(sc, w, close)
    assert w[0, P.W_HELD] == 1.0
    return w[0].copy()
