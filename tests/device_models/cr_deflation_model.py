"""numpy model of the static-variable deflation in front of cycle reduction (geconpy_amd/csrc/dsge_cr_deflate.hpp).

A variable whose columns of A (lag) and C (lead) are both exactly zero only enters through B.  A QR of those columns
of B, applied to the whole system, leaves a quadratic matrix equation in the dynamic variables alone; the static rows of
T and R follow by back-substitution.  The model uses the oracle's cycle reduction on the reduced system, so that the
test below pins the algebra (and not an iteration of its own) against the reference's goldens.
"""

from __future__ import annotations

import numpy as np

import oracle


def static_columns(A: np.ndarray, C: np.ndarray) -> np.ndarray:
    return np.where(~(A != 0).any(axis=0) & ~(C != 0).any(axis=0))[0]


def deflated_cycle_reduction(A, B, C, D, max_iter=1000, tol=1e-9, h=None):
    """Returns T, R, number of static variables used, iterations of the reduced system."""
    n, k = A.shape[0], D.shape[1]
    st = static_columns(A, C)
    if h is not None:
        st = st[:h]  # the device keeps the first h static variables; the others stay in the dynamic block
    dy = np.setdiff1d(np.arange(n), st)
    h, nd = len(st), n - len(st)
    W = np.hstack([B[:, st], B[:, dy], A[:, dy], C[:, dy], D])
    if h:
        Q, _ = np.linalg.qr(B[:, st], mode="complete")
        W = Q.T @ W
    R_st = W[:h, :h]
    Btop, Atop, Ctop = (W[:h, h + i * nd : h + (i + 1) * nd] for i in range(3))
    Dtop = W[:h, h + 3 * nd :]
    Bred, Ared, Cred = (W[h:, h + i * nd : h + (i + 1) * nd] for i in range(3))
    Dred = W[h:, h + 3 * nd :]
    T_dy, converged, it = oracle.cycle_reduction_core(Ared, Bred, Cred, max_iter, tol)[:3]
    R_dy = oracle.compute_selection_matrix(Bred, Cred, Dred, T_dy)
    G1 = Btop + Ctop @ T_dy
    T = np.zeros((n, n))
    R = np.zeros((n, k))
    T[np.ix_(dy, dy)] = T_dy
    R[dy] = R_dy
    if h:
        T[np.ix_(st, dy)] = -np.linalg.solve(R_st, G1 @ T_dy + Atop)
        R[st] = -np.linalg.solve(R_st, G1 @ R_dy + Dtop)
    return T, R, h, it
