"""numpy-level entry points with the reference's names, argument meaning and return shapes,
executing on the HIP engine (batch of one or many).

  solve_policy_function_with_cycle_reduction  <- gEconpy/solvers/cycle_reduction.py:328-398
  solve_policy_function_with_backward_direct  <- gEconpy/solvers/backward_looking.py:102-134
  solve_policy_function_with_gensys / gensys  <- gEconpy/solvers/gensys.py:617-631, :398-521
  cycle_reduction_numpy-like batched driver    <- the per-draw loop of
      gEconpy/model/statistics/perturbation_diagnostics.py:453-490 (one launch instead of a pool)
"""
from __future__ import annotations

import logging

import numpy as np

from . import _lib, batched

_log = logging.getLogger(__name__)

MSG_OK = "Optimization successful"
MSG_FAIL = "Iteration on all matrices failed to converged"


def solve_policy_function_with_cycle_reduction(A, B, C, D, max_iter=100, tol=1e-8, verbose=True):
    """``(T, R, result, log_norm)`` for one system.  On failure ``T`` and ``R`` are ``None`` and
    ``result`` carries the reference's failure message; ``log_norm`` (the reference reports
    log||A1||_1 of the last iterate there, cycle_reduction.py:107) is not produced on device and
    is returned as NaN in that case."""
    A3, B3, C3, D3 = (np.ascontiguousarray(x, dtype=np.float64)[None] for x in (A, B, C, D))
    T, status, n_iter = batched.cycle_reduction_batched(A3, B3, C3, max_iter=max_iter, tol=tol)
    if status[0] != 0:
        if verbose:
            _log.info("Solution not found. Solver returned: %s", MSG_FAIL)
        return None, None, MSG_FAIL, float("nan")
    R, resid = batched.selection_batched(B3, C3, D3, T, A=A3)
    if verbose:
        _log.info("Solution found, sum of squared residuals: %0.9f", resid[0])
    return T[0], R[0], MSG_OK, 0


def solve_policy_function_with_backward_direct(A, B, C, D):
    """``(T, R)`` with ``T = (-B)^-1 A``, ``R = -B^-1 D``; ``C`` is accepted and ignored, as in the
    reference."""
    del C
    A3, B3, D3 = (np.ascontiguousarray(x, dtype=np.float64)[None] for x in (A, B, D))
    T, R = batched.backward_direct_batched(A3, B3, D3)
    return T[0], R[0]


def solve_policy_function_with_gensys(A, B, C, D, tol=1e-8, return_all_matrices=True):
    """Reference signature (gEconpy/solvers/gensys.py:617-631).  The estimation and ``solve_model`` call
    sites consume ``G_1[:n, :n]``, ``impact[:n]`` and ``eu`` only (gensys.py:657-666,
    gEconpy/model/model.py:1696-1708); the device produces exactly those, so

      * ``return_all_matrices=False`` -> ``(G_1, eu)`` with ``G_1`` the n x n policy block ``T``;
      * ``return_all_matrices=True``  -> the 9-tuple ``(G_1, constant, impact, f_mat, f_wt, y_wt, gev, eu, loose)``
        with ``G_1 = T`` (n x n), ``constant = 0`` (c = 0, gensys.py:598), ``impact = R = -(C T + B)^-1 D``
        (what ``gensys_pt`` uses, :679-683; it equals ``impact[:n]`` of the QZ formula to ~1e-12) and
        ``None`` for ``f_mat, f_wt, y_wt, gev, loose`` -- the five outputs no caller of the hot path reads;
        on coincident zeros (``eu = [-2, -2, 0]``) the seven matrices are ``None`` as at :515-516.
    Slicing ``G_1[:n, :n]`` / ``impact[:n, :]`` as the callers do is a no-op on these shapes."""
    A3, B3, C3, D3 = (np.ascontiguousarray(x, dtype=np.float64)[None] for x in (A, B, C, D))
    out = batched.gensys_batched(A3, B3, C3, D3, tol=tol)
    eu = [int(v) for v in out["eu"][0]]
    if eu[0] == -2 and eu[1] == -2:
        return (None, eu) if not return_all_matrices else (None,) * 7 + (eu, None)
    G_1 = np.ascontiguousarray(out["T"][0])
    if not return_all_matrices:
        return G_1, eu
    n = G_1.shape[0]
    return G_1, np.zeros((n, 1)), out["R"][0], None, None, None, None, eu, None


def gensys(g0, g1, c, psi, pi, div=None, tol=1e-8, return_all_matrices=True):
    """The raw-pencil entry point (gEconpy/solvers/gensys.py:398-521) is host-side bookkeeping around
    ``_gensys_core`` for an ARBITRARY pencil; the device kernel takes the structural form A, B, C
    (it never materialises the pencil).  Not provided: use
    ``solve_policy_function_with_gensys(A, B, C, D)``."""
    raise NotImplementedError(gensys.__doc__)


def solve_policy_functions_batched(A, B, C, D, solver="cycle_reduction", max_iter=100, tol=1e-8):
    """Many draws, one launch: dict(T, R, resid, success, n_iter).  ``success[i]`` is what
    ``_solve_perturbation`` decides per draw in the reference's ``solvability_check`` loop
    (perturbation_diagnostics.py:69-99); results come back in input order (the reference's fork pool
    returns them in completion order, :484-489)."""
    A, B, C = (np.ascontiguousarray(x, dtype=np.float64) for x in (A, B, C))
    D = np.ascontiguousarray(D, dtype=np.float64)
    if solver == "cycle_reduction":
        T, status, n_iter = batched.cycle_reduction_batched(A, B, C, max_iter=max_iter, tol=tol)
        R, resid = batched.selection_batched(B, C, D, T, A=A)
        ok = status == _lib.ST_OK
        R[~ok] = 0.0
        resid[~ok] = np.inf
    elif solver == "scan_cycle_reduction":
        T, status, n_iter = batched.scan_cycle_reduction_batched(A, B, C, max_iter=max_iter, tol=tol)
        R, resid = batched.selection_batched(B, C, D, T, A=A)
        ok = status == _lib.ST_OK
        R[~ok] = 0.0
        resid[~ok] = np.inf
    elif solver == "gensys":
        out = batched.gensys_batched(A, B, C, D, tol=tol)
        T, R, ok = out["T"], out["R"], out["success"]
        _R2, resid = batched.selection_batched(B, C, D, T, A=A)
        R[~ok] = 0.0
        resid[~ok] = np.inf
        n_iter = np.zeros(A.shape[0], dtype=np.int32)
        return dict(T=T, R=R, resid=resid, success=ok, n_iter=n_iter, eu=out["eu"])
    elif solver == "backward_direct":
        T, R = batched.backward_direct_batched(A, B, D)
        resid = np.square(A + B @ T).sum(axis=(1, 2))
        ok = np.isfinite(resid)
        n_iter = np.zeros(A.shape[0], dtype=np.int32)
    else:
        raise NotImplementedError(f"solver {solver!r} is not available on the HIP engine yet")
    return dict(T=T, R=R, resid=resid, success=ok, n_iter=n_iter)
