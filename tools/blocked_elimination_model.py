"""CPU, numpy only: a model of the device's blocked elimination (gauss_jordan_blocked in csrc/dsge_device.hpp: Gauss-Jordan
inside a panel of BS columns, the panel's combined transformation applied to the ORIGINAL pivot rows in one shot, i.e. through
the explicit inverse of the BS x BS pivot block) inside cycle reduction, on the draw the fuzz campaign flagged (tools/fuzz_cr.py
seed 11, trial 1727, draw 3: n = 62, cond(A1) = 1.2e8 in the second iteration), against LAPACK and an extended-precision LU.
Result (profiles/r2/blocked_elimination_model.txt): the blocked Gauss-Jordan is at 3e-7 (the one-wavefront kernels, panels of
eight columns: 1.5e-7 measured), not eliminating above the panel and substituting back afterwards only halves that, LAPACK
is at 1e-10 and an UNBLOCKED Gauss-Jordan at 3e-8 (tools/gj_vs_lu.py).  So the loss has two parts: the one-shot application of
the panel through the inverse of its pivot block (x 10) and the elimination order (x 300); the fix needs sequential
multipliers (TRSM-style panel application) AND an LU order with back substitution."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from geconpy_amd import workloads as wl
A, B, C = wl.sw_shaped_system(386903548, n=62, n_state=9, n_lead=3, k=1)[:3]
n = A.shape[0]

def blocked_solve(M, R, BS=8, lu=True):
    """The device's blocked elimination on W = [M | R]: Gauss-Jordan inside a panel of BS columns; between panels either
    Gauss-Jordan (rows pivoted earlier are eliminated too) or LU (they are not; back substitution afterwards)."""
    W = np.hstack((M, R)).astype(np.float64); n = M.shape[0]; ncol = W.shape[1]
    used = np.zeros(n, bool); prow = np.zeros(n, int); nsteps = (n + BS - 1) // BS; used_before = []
    for kb in range(nsteps):
        j0 = kb * BS; bw = min(BS, n - j0); used_before.append(used.copy())
        pw = W[:, j0:j0 + bw].copy(); idm = np.zeros((n, bw)); rsel = []; inv_own = np.ones(n)
        earlier = used_before[kb]
        for c in range(bw):
            r = int(np.argmax(np.where(~used, np.abs(pw[:, c]), -1.0))); used[r] = True; rsel.append(r); idm[r, c] = 1.0
            inv = 1.0 / pw[r, c]
            f = pw[:, c].copy(); f[r] = 0.0
            if lu: f[earlier] = 0.0
            inv_own[r] = inv
            for c2 in range(bw):
                if c2 > c: pw[:, c2] -= f * (pw[r, c2] * inv)
                if c2 <= c: idm[:, c2] -= f * (idm[r, c2] * inv)
        idm *= inv_own[:, None]
        lh = -idm
        for a, r in enumerate(rsel): lh[r, a] += 1.0
        Y = W[rsel, :].copy()
        prow[j0:j0 + bw] = rsel
        W[:, j0 + bw:] -= lh @ Y[:, j0 + bw:]
    if lu:
        for kb in range(nsteps - 1, 0, -1):
            j0 = kb * BS; bw = min(BS, n - j0)
            L = W[:, j0:j0 + bw].copy(); L[~used_before[kb]] = 0.0
            Y = W[prow[j0:j0 + bw], n:]
            W[:, n:] -= L @ Y
    return W[prow, n:]

def lu_solve_ext(M, R):
    M, R = M.copy(), R.copy()
    for c in range(n):
        r = c + int(np.argmax(np.abs(M[c:, c])))
        if r != c: M[[c, r]] = M[[r, c]]; R[[c, r]] = R[[r, c]]
        for i in range(c + 1, n):
            fct = M[i, c] / M[c, c]; M[i, c:] -= fct * M[c, c:]; R[i] -= fct * R[c]
    X = np.zeros_like(R)
    for i in range(n - 1, -1, -1): X[i] = (R[i] - M[i, i + 1:] @ X[i + 1:]) / M[i, i]
    return X

def cr(solve_of_iter, dtype=np.float64, iters=8):
    A0, A1, A2, A1h = (x.astype(dtype) for x in (A, B, C, B))
    for it in range(iters):
        X = solve_of_iter(it)(A1, np.hstack((A0, A2))); X0, X2 = X[:, :n], X[:, n:]
        A0, A1, A2, A1h = -(A0 @ X0), A1 - A0 @ X2 - A2 @ X0, -(A2 @ X2), A1h - A2 @ X0
    return -solve_of_iter(99)(A1h, A.astype(dtype))
T_ref = cr(lambda it: lu_solve_ext, np.longdouble).astype(np.float64)
for name, f in (("LAPACK", lambda it: np.linalg.solve),
                ("blocked GJ (device today)", lambda it: (lambda M, R: blocked_solve(M, R, 8, False))),
                ("blocked LU all iterations", lambda it: (lambda M, R: blocked_solve(M, R, 8, True))),
                ("blocked LU in iterations 0-1 only", lambda it: (lambda M, R: blocked_solve(M, R, 8, it < 2))),
                ("blocked LU in iterations 0-2 + final", lambda it: (lambda M, R: blocked_solve(M, R, 8, it < 3 or it == 99)))):
    T = cr(f); print(f"{name:40s} |T - T_ext| = {np.abs(T - T_ref).max():.2e}")
