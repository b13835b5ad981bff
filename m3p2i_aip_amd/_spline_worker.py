"""Worker process of sampling._spline_rows_parallel: reads (rows [n, nu, n_knots], T, degree)
pickled on stdin, writes the fitted [n, T, nu] float32 array pickled on stdout.  Imports numpy and
scipy only (no torch, no HIP): it must start fast and must not touch the GPU."""
import pickle
import sys

import numpy as np


def spline_rows(rows, T, degree):
    import scipy.interpolate as si
    n_knots = rows.shape[2]
    x = np.linspace(0, n_knots, n_knots)
    xe = np.linspace(0, n_knots, T)
    out = np.zeros((rows.shape[0], T, rows.shape[1]), np.float32)
    for i in range(rows.shape[0]):
        for j in range(rows.shape[1]):
            out[i, :, j] = si.splev(xe, si.splrep(x, rows[i, j], k=degree, s=0.5), ext=3)
    return out


if __name__ == "__main__":
    rows, T, degree = pickle.load(sys.stdin.buffer)
    sys.stdout.buffer.write(pickle.dumps(spline_rows(rows, T, degree), protocol=4))
    sys.stdout.buffer.flush()
