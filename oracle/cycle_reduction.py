"""Oracle restatement of cycle reduction (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/gEconpy/solvers/cycle_reduction.py:
  * ``cycle_reduction_numpy`` <- same name (:23-114), numpy stopping rule incl. the
    ``elif`` quirk (:96-109)
  * ``cycle_reduction_core``  <- ``_cycle_reduction_core`` (:127-183), njit stopping rule
  * ``scan_cycle_reduction``  <- ``_scan_cycle_reduction`` (:246-294) + shared.py:6-9
  * ``solve_policy_function_with_cycle_reduction`` <- same name (:328-398)
Solves ``A0 + A1 X + A2 X^2 = 0`` (called with (A0,A1,A2) = (A,B,C)).
"""
from __future__ import annotations

import numpy as np

MSG_OK = "Optimization successful"
MSG_A2 = "Iteration on matrix A0 and A1 converged towards a solution, but A2 did not."
MSG_FAIL = "Iteration on all matrices failed to converged"


def _norm1(M):
    """Induced 1-norm = max absolute column sum (np.linalg.norm(., ord=1))."""
    return np.abs(M).sum(axis=0).max()


def _cr_step(A0, A1, A2, A1_hat, jitter=0.0):
    """One Bini-Latouche-Meini step (cycle_reduction.py:88-93 / :151-169)."""
    n = A0.shape[0]
    lhs = A1 + jitter * np.eye(n) if jitter else A1
    X = np.linalg.solve(lhs, np.hstack((A0, A2)))
    X0, X2 = X[:, :n], X[:, n:]
    m00 = A0 @ X0
    m02 = A0 @ X2
    m20 = A2 @ X0
    m22 = A2 @ X2
    return -m00, A1 - m02 - m20, -m22, A1_hat - m20


def cycle_reduction_numpy(A0, A1, A2, max_iter=1000, tol=1e-7):
    """cycle_reduction.py:23-114 -> (X | None, res | None, message, log_norm)."""
    A0i, A1i, A2i = A0, A1, A2
    A1_hat = A1
    log_norm = 0
    for i in range(int(max_iter)):
        with np.errstate(all="ignore"):
            A0, A1, A2, A1_hat = _cr_step(A0, A1, A2, A1_hat)
        nrm0 = _norm1(A0)
        if nrm0 < tol:
            if _norm1(A2) < tol:
                break
        elif np.isnan(nrm0) or i == max_iter - 1:
            # nrm0 >= tol (or NaN) here, so only the second message is reachable; the
            # first is kept because the reference carries it (:102-104).
            if nrm0 < tol:
                return None, None, MSG_A2, np.log(_norm1(A2))
            return None, None, MSG_FAIL, np.log(_norm1(A1))
    X = -np.linalg.solve(A1_hat, A0i)
    res = A0i + A1i @ X + A2i @ X @ X
    return X, res, MSG_OK, log_norm


def cycle_reduction_core(A0, A1, A2, max_iter, tol):
    """cycle_reduction.py:127-183 -> (T, converged); zeros on failure (:181).

    Also returns the iteration count as a third value (the device kernel reports it).
    """
    A0i = A0
    A1_hat = A1
    converged = False
    n_iter = 0
    for _ in range(int(max_iter)):
        with np.errstate(all="ignore"):
            try:
                A0, A1, A2, A1_hat = _cr_step(A0, A1, A2, A1_hat)
            except np.linalg.LinAlgError:  # LAPACK getrs on an exactly singular LU -> NaN fill
                A0 = np.full_like(A0, np.nan)
        n_iter += 1
        nrm0 = _norm1(A0)
        if nrm0 < tol:
            if _norm1(A2) < tol:
                converged = True
                break
        elif np.isnan(nrm0):
            break
    if converged:
        T = -np.linalg.solve(A1_hat, A0i)
    else:
        T = np.zeros_like(A0i)
    return T, converged, n_iter


def scan_cycle_reduction(A, B, C, max_iter=50, tol=1e-7):
    """cycle_reduction.py:246-294: fixed trip count, no-op once ``||A0||_1 < tol``
    (A0 norm only), 1e-16 diagonal jitter on every solve (shared.py:6-9)."""
    A0, A1, A2, A1_hat = A, B, C, B
    norm = 1e9
    n_steps = 0
    for _ in range(int(max_iter)):
        if norm < tol:
            continue
        A0, A1, A2, A1_hat = _cr_step(A0, A1, A2, A1_hat, jitter=1e-16)
        norm = _norm1(A0)
        n_steps += 1
    n = A.shape[0]
    T = -np.linalg.solve(A1_hat + 1e-16 * np.eye(n), A)
    return T, n_steps


def solve_policy_function_with_cycle_reduction(A, B, C, D, max_iter=100, tol=1e-8, verbose=False):
    """cycle_reduction.py:328-398 -> (T, R, message, log_norm)."""
    del verbose
    T, _res, msg, log_norm = cycle_reduction_numpy(A, B, C, max_iter, tol)
    R = None
    if T is not None:
        T = np.ascontiguousarray(T)
        R = -np.linalg.solve(C @ T + B, D)  # :395-396
    return T, R, msg, log_norm
