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


def gensys_setup(A, B, C, D, tol=1e-8):
    """The Sims pencil of ``_gensys_setup`` (gEconpy/solvers/gensys.py:568-614), by index arithmetic (host bookkeeping, no
    floating-point work): with ``lead`` = columns of C whose absolute column sum exceeds ``tol`` and ``w_t = [y_t; E_t y_{t+1}[lead]]``

        G0 = [[-B, -C[:, lead]], [E, 0]]   G1 = [[A, 0], [0, I]]   c = 0   Psi = [D; 0]   Pi = [0; I]

    -> ``(g0, g1, c, psi, pi)`` of dimension N = n + #lead."""
    A, B, C, D = (np.ascontiguousarray(x, dtype=np.float64) for x in (A, B, C, D))
    n, k = D.shape
    lead = np.flatnonzero(np.abs(C).sum(axis=0) > tol)
    nl = lead.size
    N = n + nl
    g0 = np.zeros((N, N))
    g0[:n, :n] = -B
    g0[:n, n:] = -C[:, lead]
    g0[n + np.arange(nl), lead] = 1.0
    g1 = np.zeros((N, N))
    g1[:n, :n] = A
    g1[n + np.arange(nl), n + np.arange(nl)] = 1.0
    psi = np.zeros((N, k))
    psi[:n] = D
    pi = np.zeros((N, nl))
    pi[n + np.arange(nl), np.arange(nl)] = 1.0
    return g0, g1, np.zeros((N, 1)), psi, pi


def gensys(g0, g1, c, psi, pi, div=None, tol=1e-8, return_all_matrices=True):
    """Reference signature and return (gEconpy/solvers/gensys.py:398-521) for an arbitrary pencil
    ``g0 y_t = g1 y_{t-1} + c + psi z_t + pi eta_t``, solved on the device by ``dsge_gensys_pencil_full_batched`` (ordered
    complex QZ, existence / uniqueness SVDs, one triangular solve; N <= ~52, everything resident in LDS):

      * ``return_all_matrices=False`` -> ``(G_1, eu)``;
      * ``return_all_matrices=True``  -> ``(G_1, C, impact, f_mat, f_wt, y_wt, gev, eu, loose)`` with ``G_1`` (N, N),
        ``C`` (N, 1), ``impact`` (N, k) from the QZ formulas (:336-365), ``gev`` (N, 2) complex ``[alpha, beta]`` (:253),
        ``f_mat`` (nu, nu), ``f_wt`` (nu, k), ``y_wt`` (N, nu) complex and ``loose`` (N, n_eta) real (:367-393), all formed on
        the device.  ``f_mat, f_wt, y_wt`` are defined up to the unitary basis of the unstable block of the ordered Schur
        form (LAPACK's choice in the reference, this library's here); ``y_wt @ f_mat**s @ f_wt`` and ``eig(f_mat)`` agree;
      * coincident zeros (``eu = [-2, -2, 0]``): every matrix is ``None`` (:515-516).
    ``div`` is accepted and ignored, as in the reference (:502).  A pencil beyond the kernel's LDS raises
    ``_lib.DsgeTooLargeError`` (return code DSGE_ERR_TOO_LARGE)."""
    del div
    tol_eff = tol if tol is not None and tol > 0 else float(np.spacing(1))
    g0, g1, psi, pi = (np.ascontiguousarray(x, dtype=np.float64) for x in (g0, g1, psi, pi))
    c = np.ascontiguousarray(c, dtype=np.float64).reshape(g0.shape[0], -1)
    if c.shape[1] != 1:
        raise ValueError("c must have one column")
    out = batched.gensys_pencil_batched(g0[None], g1[None], psi[None], pi[None], c=c[None, :, 0], tol=tol_eff,
                                        forward=bool(return_all_matrices))
    eu = [int(v) for v in out["eu"][0]]
    if eu[0] == -2 and eu[1] == -2:
        return None, None, None, None, None, None, None, eu, None
    G_1 = out["G1"][0]
    if not return_all_matrices:
        return G_1, eu
    nu = int(out["n_unstable"][0])
    f_mat = np.ascontiguousarray(out["f_mat"][0][:nu, :nu])
    f_wt = np.ascontiguousarray(out["f_wt"][0][:nu])
    y_wt = np.ascontiguousarray(out["y_wt"][0][:, :nu])
    return G_1, out["C"][0][:, None], out["impact"][0], f_mat, f_wt, y_wt, out["gev"][0], eu, out["loose"][0]


def solve_policy_function_with_gensys(A, B, C, D, tol=1e-8, return_all_matrices=True):
    """Reference signature and return (gEconpy/solvers/gensys.py:617-631): the pencil of ``_gensys_setup`` handed to
    ``gensys``.  ``G_1`` is (N, N) with N = n + #lead and ``impact`` (N, k); callers slice ``G_1[:n, :n]``,
    ``impact[:n, :]`` (gensys.py:657-666, gEconpy/model/model.py:1696-1708).  The batched estimation path does not go
    through here: it uses the structure-exploiting kernels behind ``batched.gensys_batched`` (T and eu only).

    The raw-pencil kernel keeps H, T, Z and Q as complex N x N matrices in LDS, which limits it to N <= ~52.  A larger pencil
    (n + #lead up to 64; the library answers DSGE_ERR_TOO_LARGE, a return CODE -- the route is never chosen from message text)
    is solved by the window-path kernels of ``batched.gensys_batched``: ``G_1[:n, :n] = T`` and
    ``impact[:n] = R = -(C T + B)^-1 D`` (gensys.py:679-683) are what every caller reads; the lead rows are completed ON the
    stable manifold (``x_t = E_t y_{t+1}[lead] = T[lead] y_t``, so ``G_1[n:, :n] = T[lead] T``, ``impact[n:] = T[lead] R``,
    zero columns for ``x_{t-1}``) and ``C = 0`` (the model pencil has c = 0, :598).  That route keeps the ordered Schur form
    of the active window only, so ``gev, f_mat, f_wt, y_wt, loose`` are ``None`` there.  A draw WITHOUT a unique stable
    solution (``eu != [1, 1]``) still returns ``G_1`` and ``impact`` as arrays on BOTH routes (the reference's consumer slices
    ``G_1[:n, :n]`` before it reads ``eu``, gensys.py:657-666) -- on this route the zero-filled T of the failed draw, not the QZ
    quantities of the failed solve, which nothing downstream reads (model.py:1696-1702 raises first); coincident zeros
    (``eu = [-2, -2, 0]``) give the 9-tuple of ``None`` whatever ``return_all_matrices`` says (:515-516), as ``gensys`` does."""
    g0, g1, c, psi, pi = gensys_setup(A, B, C, D, tol)
    try:
        return gensys(g0, g1, c, psi, pi, tol=tol, return_all_matrices=return_all_matrices)
    except _lib.DsgeTooLargeError:
        pass
    A3, B3, C3, D3 = (np.ascontiguousarray(x, dtype=np.float64)[None] for x in (A, B, C, D))
    out = batched.gensys_batched(A3, B3, C3, D3, tol=tol)
    eu = [int(v) for v in out["eu"][0]]
    if eu[0] == _lib.EU_NO_VERDICT:  # 65 .. 96 variables, draw not certified regular: no QZ at that size, and no made-up matrices
        raise _lib.DsgeNoVerdictError(
            "gensys on %d variables: the draw is not certified regular (eu = [-3, -3, 0]) and there is no ordered QZ beyond 64 "
            "variables to decide it; use solver='cycle_reduction', or reduce the model" % A3.shape[1])
    if eu[0] == -2 and eu[1] == -2:  # coincident zeros: the 9-tuple of Nones, whatever return_all_matrices says (gensys.py:515-516)
        return None, None, None, None, None, None, None, eu, None
    n, k = D3.shape[1:]
    N = g0.shape[0]
    lead = np.flatnonzero(np.abs(C3[0]).sum(axis=0) > tol)
    # Without a unique stable solution (eu != [1, 1]) the reference still returns MATRICES (its consumer slices G_1[:n, :n] before
    # it looks at eu: GensysWrapper.perform, gensys.py:657-666).  So does this route, for every pencil size: T and R as the
    # window kernels wrote them for the failed draw (T zero-filled, R = -(C T + B)^-1 D of that T), never None.
    T, R = out["T"][0], out["R"][0]
    G_1 = np.zeros((N, N))
    G_1[:n, :n] = T
    G_1[n:, :n] = T[lead] @ T
    if not return_all_matrices:
        return G_1, eu
    impact = np.zeros((N, k))
    impact[:n] = R
    impact[n:] = T[lead] @ R
    return G_1, np.zeros((N, 1)), impact, None, None, None, None, eu, None


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
