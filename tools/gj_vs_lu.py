"""CPU, numpy only: why the device's cycle reduction is less accurate than the reference's on ill-conditioned draws.
The same iteration (Bini-Latouche-Meini cycle reduction) with its solves X = A1^-1 [A0 A2] done (a) by LAPACK's LU, as the
reference does, (b) by Gauss-Jordan elimination with partial pivoting, as the device kernels do, against (c) an LU in
extended precision -- on the draw the fuzz campaign flagged (tools/fuzz_cr.py seed 11, trial 1727, draw 3: n = 62,
cond(A1) = 1.2e8 in the second iteration).  Gauss-Jordan is forward stable only; the error it leaves is the device's."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from geconpy_amd import workloads as wl

A, B, C = wl.sw_shaped_system(386903548, n=62, n_state=9, n_lead=3, k=1)[:3]
n = A.shape[0]


def gj_solve(M, R):
    W = np.hstack((M, R)).astype(np.float64)
    used = np.zeros(n, bool)
    prow = np.zeros(n, int)
    for c in range(n):
        r = int(np.argmax(np.where(~used, np.abs(W[:, c]), -1.0)))
        used[r] = True
        prow[c] = r
        W[r, :] = W[r, :] / W[r, c]
        f = W[:, c].copy()
        f[r] = 0.0
        W -= np.outer(f, W[r, :])
    return W[prow, n:]


def lu_solve_ext(M, R):
    M, R = M.copy(), R.copy()
    for c in range(n):
        r = c + int(np.argmax(np.abs(M[c:, c])))
        if r != c:
            M[[c, r]] = M[[r, c]]
            R[[c, r]] = R[[r, c]]
        for i in range(c + 1, n):
            fct = M[i, c] / M[c, c]
            M[i, c:] -= fct * M[c, c:]
            R[i] -= fct * R[c]
    X = np.zeros_like(R)
    for i in range(n - 1, -1, -1):
        X[i] = (R[i] - M[i, i + 1:] @ X[i + 1:]) / M[i, i]
    return X


def cycle_reduction(solve, dtype, iters=8):
    A0, A1, A2, A1h = (x.astype(dtype) for x in (A, B, C, B))
    worst = 0.0
    for _ in range(iters):
        worst = max(worst, np.linalg.cond(A1.astype(np.float64)))
        X = solve(A1, np.hstack((A0, A2)))
        X0, X2 = X[:, :n], X[:, n:]
        A0, A1, A2, A1h = -(A0 @ X0), A1 - A0 @ X2 - A2 @ X0, -(A2 @ X2), A1h - A2 @ X0
    return -solve(A1h, A.astype(dtype)), worst


T_lu, worst = cycle_reduction(np.linalg.solve, np.float64)
T_gj, _ = cycle_reduction(gj_solve, np.float64)
T_ref = cycle_reduction(lu_solve_ext, np.longdouble)[0].astype(np.float64)
res = lambda T: np.abs(A + B @ T + C @ T @ T).max()  # noqa: E731
print(f"worst cond(A1) over the iterations: {worst:.2e}")
print(f"LAPACK LU        : |T - T_ext| = {np.abs(T_lu - T_ref).max():.2e}, residual {res(T_lu):.2e}")
print(f"Gauss-Jordan (pp): |T - T_ext| = {np.abs(T_gj - T_ref).max():.2e}, residual {res(T_gj):.2e}")
print("device (profiles/r2/fuzz_campaign_seeds11_14.txt, rerun): default kernels 2.0e-08, one-wavefront kernels 1.5e-07")
