"""Joint-space inertias of the Panda for chain spec v2, derived from the reference's own assets.

The reference's URDF has no <inertial> elements (franka_panda.urdf), so Isaac Gym derives every link's mass, centre of
mass and inertia tensor from its COLLISION mesh at the default density 1000 kg/m^3.  This script
does the same -- signed-tetrahedron sums over meshes/collision/*.obj -- then assembles the joint-space mass matrix
M(q0) = sum_links ( m Jv^T Jv + Jw^T R Ic R^T Jw ) at the configured initial pose q0 (config/panda_env/panda.yaml:10)
and prints its DIAGONAL: M_ii is the inertia joint i drives with the other joints held, which is what the per-joint
velocity servo and the contact rows of the spec use (the spec keeps the matrix diagonal: the drive, D h = 3 kg m^2 per
substep, dominates every entry, see DESIGN).  Runs in the build container only (reads /root/reference); its output is
committed as constants in csrc/panda_dyn.hpp / oracle/panda_chain.c.

    python tools/panda_inertia.py [/root/reference]
"""
import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
MESH = os.path.join(REF, "src/m3p2i_aip/assets/urdf/franka_description/meshes/collision")
RHO = 1000.0


def load_obj(path):
    v, f = [], []
    for line in open(path):
        p = line.split()
        if not p:
            continue
        if p[0] == "v":
            v.append([float(x) for x in p[1:4]])
        elif p[0] == "f":
            idx = [int(t.split("/")[0]) - 1 for t in p[1:]]
            for i in range(1, len(idx) - 1):
                f.append([idx[0], idx[i], idx[i + 1]])
    return np.array(v), np.array(f)


def mass_props(v, f):
    """mass, centre of mass, inertia tensor about the centre of mass (signed tetrahedra against the origin)."""
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    vol6 = np.einsum("ij,ij->i", a, np.cross(b, c))
    V = vol6.sum() / 6.0
    com = ((a + b + c) / 4.0 * (vol6 / 6.0)[:, None]).sum(0) / V
    # second moments: integral of x_i x_j over a tetrahedron (0, a, b, c) = vol/20 * (sum_{p,q} p_i q_j + sum_p p_i p_j)
    S = np.zeros((3, 3))
    for t in range(len(f)):
        P = np.stack([a[t], b[t], c[t]])
        s = P.sum(0)
        S += (vol6[t] / 6.0) / 20.0 * (np.outer(s, s) + P.T @ P)
    if V < 0:
        V, S = -V, -S
    m = RHO * V
    C = RHO * S                                   # second-moment matrix about the origin
    I0 = np.trace(C) * np.eye(3) - C              # inertia tensor about the origin
    d = com
    Ic = I0 - m * ((d @ d) * np.eye(3) - np.outer(d, d))
    return m, com, Ic


def rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr], [-sp, cp * sr, cp * cr]])


def rz(q):
    c, s = np.cos(q), np.sin(q)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


H = np.pi / 2
# franka_panda.urdf joint origins (:29,50,71,92,116,137,160), hand (:177-187), fingers (:226-242)
JOINTS = [((0, 0, 0.333), (0, 0, 0)), ((0, 0, 0), (-H, 0, 0)), ((0, -0.316, 0), (H, 0, 0)), ((0.0825, 0, 0), (H, 0, 0)),
          ((-0.0825, 0.384, 0), (-H, 0, 0)), ((0, 0, 0), (H, 0, 0)), ((0.088, 0, 0), (H, 0, 0))]
Q0 = [0, 0, 0, -2.0, 0, 1.8675, 0]


def main():
    names = ["link1", "link2", "link3", "link4", "link5", "link6", "link7", "hand", "finger", "finger"]
    props = [mass_props(*load_obj(os.path.join(MESH, n + ".obj"))) for n in names]
    R, p = np.eye(3), np.zeros(3)
    axes, origins, frames = [], [], []
    for (xyz, r), q in zip(JOINTS, Q0):
        p = p + R @ np.array(xyz)
        R = R @ rpy(*r) @ rz(q)
        axes.append(R[:, 2].copy()); origins.append(p.copy()); frames.append((R.copy(), p.copy()))
    Rh = R @ rz(-np.pi / 4); ph = p + R @ np.array([0, 0, 0.107])
    frames.append((Rh, ph))
    for sgn, q in ((1, 0.02), (-1, 0.02)):       # fingers: prismatic along +y / -y of the hand, at (0, 0, 0.0584)
        Rf = Rh if sgn > 0 else Rh @ rz(np.pi)   # (the right finger's collision mesh is the left one turned by pi)
        frames.append((Rf, ph + Rh @ np.array([0, sgn * q, 0.0584])))
    M = np.zeros((9, 9))
    for li, ((m, com, Ic), (Rl, pl)) in enumerate(zip(props, frames)):
        c = pl + Rl @ com
        Iw = Rl @ Ic @ Rl.T
        Jv, Jw = np.zeros((3, 9)), np.zeros((3, 9))
        for j in range(7):
            if j <= min(li, 6):
                Jv[:, j] = np.cross(axes[j], c - origins[j]); Jw[:, j] = axes[j]
        if li == 8: Jv[:, 7] = Rh[:, 1]
        if li == 9: Jv[:, 8] = -Rh[:, 1]
        M += m * Jv.T @ Jv + Jw.T @ Iw @ Jw
        print("%-7s m = %.4f kg  com = (%.4f %.4f %.4f)  I_c diag = (%.5f %.5f %.5f)" % ((names[li], m) + tuple(com) + tuple(np.diag(Ic))))
    np.set_printoptions(precision=4, suppress=True, linewidth=160)
    print("M(q0) =\n", M)
    print("diag M(q0) =", np.diag(M))
    off = np.abs(M - np.diag(np.diag(M))).max()
    print("largest off-diagonal entry %.4f; drive term D*h = %.1f at h = 0.005" % (off, 600 * 0.005))


if __name__ == "__main__":
    main()
