"""Batched numpy front-end of the HIP engine (host arrays in, host arrays out).

Every function takes arrays with a leading draw axis, calls ONE C-ABI entry point of
libdsge_hip.so (the ``*_host`` twins, which stage through device memory) and returns fresh
numpy arrays.  Layout/dtype coercion mirrors the reference
(``np.ascontiguousarray(..., float64)``, gEconpy/solvers/gensys.py:625-628).
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from ._lib import DsgeHipError  # noqa: F401

JITTER_DEFAULT = 1e-8  # float64 cov_jitter default (gEconpy/model/statespace.py:22,1144)
MISSING_FILL = -9999.0  # default missing_fill_value (gEconpy/model/statespace.py:1143)
FILTER_TYPES_UPSTREAM = ("standard", "univariate", "steady_state", "single", "cholesky")


def check_filter_type(filter_type):
    """``filter_type`` of ``statespace_from_gcn`` / ``DSGEStateSpace`` (gEconpy/model/build.py:577, 608-609; statespace.py:69, 187:
    handed to PyMCStateSpace).  The device computes the log-likelihood of the "standard" filter -- the default, and the only one
    the oracle restates.  The other variants are different algorithms with their own jitter and missing-data conventions
    (univariate: observation-at-a-time updates; steady_state: a constant gain from the first step; single / cholesky: other
    factorisations of F); silently running the standard recursion under their name would return a number that differs from the
    reference's, so anything but "standard" raises."""
    if filter_type == "standard":
        return
    if filter_type in FILTER_TYPES_UPSTREAM:
        raise NotImplementedError(
            f"filter_type={filter_type!r}: only the 'standard' Kalman filter is built on the device "
            "(DESIGN.md section 7); evaluate this model with the reference's CPU filter")
    raise ValueError(f"unknown filter_type {filter_type!r}; upstream knows {FILTER_TYPES_UPSTREAM}")


def _f64(x, ndim=None):
    a = np.ascontiguousarray(x, dtype=np.float64)
    if ndim is not None and a.ndim != ndim:
        raise ValueError(f"expected a {ndim}-d array, got shape {a.shape}")
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data


def _check_abc(A, B, C):
    A, B, C = _f64(A, 3), _f64(B, 3), _f64(C, 3)
    if not (A.shape == B.shape == C.shape and A.shape[1] == A.shape[2]):
        raise ValueError(f"A, B, C must be (batch, n, n); got {A.shape}, {B.shape}, {C.shape}")
    return A, B, C


def cycle_reduction_batched(A, B, C, max_iter=1000, tol=1e-9, options=None):
    """T, status, n_iter for a batch of systems (``_cycle_reduction_core`` semantics,
    gEconpy/solvers/cycle_reduction.py:127-183; Op defaults :190)."""
    A, B, C = _check_abc(A, B, C)
    nb, n, _ = A.shape
    T = np.empty_like(A)
    status = np.empty(nb, dtype=np.int32)
    n_iter = np.empty(nb, dtype=np.int32)
    with _lib.options_scope(options):
        _lib.check(
            _lib.load().dsge_cycle_reduction_batched_host(
                _ptr(A), _ptr(B), _ptr(C), nb, n, int(max_iter), float(tol), _ptr(T), _ptr(status), _ptr(n_iter)
            )
        )
    return T, status, n_iter


def scan_cycle_reduction_batched(A, B, C, max_iter=50, tol=1e-7):
    """Batched ``scan_cycle_reduction`` (gEconpy/solvers/cycle_reduction.py:246-325) -> (T, status, n_steps):
    A0-norm-only stopping rule, fixed trip count, 1e-16 diagonal jitter in every solve."""
    A, B, C = _check_abc(A, B, C)
    nb, n, _ = A.shape
    T = np.empty_like(A)
    status = np.empty(nb, dtype=np.int32)
    n_steps = np.empty(nb, dtype=np.int32)
    _lib.check(
        _lib.load().dsge_scan_cycle_reduction_batched_host(_ptr(A), _ptr(B), _ptr(C), nb, n, int(max_iter), float(tol),
                                                           _ptr(T), _ptr(status), _ptr(n_steps))
    )
    return T, status, n_steps


def lead_hint(C, tol=0.0):
    """Performance hint: number of columns of ``C`` (any leading batch axes) whose absolute column
    sum exceeds ``tol`` in at least one draw -- the forward-looking variables (gensys.py:580-589)."""
    C = np.asarray(C)
    cs = np.abs(C).sum(axis=-2)
    return int(np.count_nonzero(np.any(cs.reshape(-1, C.shape[-1]) > tol, axis=0)))


def gensys_batched(A, B, C, D=None, tol=1e-8, n_lead_hint=None, options=None):
    """Batched ``GensysWrapper`` / ``gensys_pt`` (gEconpy/solvers/gensys.py:634-683):
    returns dict(T, success, eu, status[, R]).

    ``eu[i]`` is the reference's code triple -- with ONE code the reference does not have: for 65 .. 96 variables (no ordered QZ at
    that size) a draw that the spectral-division certificate cannot prove regular comes back as ``eu = [-3, -3, 0]``
    (``_lib.EU_NO_VERDICT``), ``success = False``, ``T = R = 0``: "no verdict at this size", never a wrong one.  It covers non-regular
    draws (the reference would say ``[1, 0, k]``, ``[0, 1, 0]``, ...) AND regular ones with a root within 2e-4 of the unit circle
    or a scale defect (the reference would solve them); consumers written against ``interpret_gensys_output`` should treat it as a
    failed draw."""
    A, B, C = _check_abc(A, B, C)
    nb, n, _ = A.shape
    T = np.empty_like(A)
    eu = np.empty((nb, 3), dtype=np.int32)
    status = np.empty(nb, dtype=np.int32)
    R = None
    k = 1
    if D is not None:
        D = _f64(D, 3)
        k = D.shape[2]
        R = np.empty((nb, n, k))
    nl = lead_hint(C, tol) if n_lead_hint is None else int(n_lead_hint)
    with _lib.options_scope(options):
        _lib.check(
            _lib.load().dsge_gensys_batched_host(_ptr(A), _ptr(B), _ptr(C), _ptr(D), nb, n, k, float(tol), nl, _ptr(T),
                                                 _ptr(R), _ptr(eu), _ptr(status))
        )
    out = dict(T=T, success=status == 0, eu=eu, status=status)
    if R is not None:
        out["R"] = R
    return out


def gensys_pencil_batched(g0, g1, psi, pi, c=None, tol=1e-8, forward=False):
    """Batched ``gensys(g0, g1, c, psi, pi)`` (gEconpy/solvers/gensys.py:398-521) on caller-supplied pencils:
    returns dict(G1, C, impact, gev (batch, N, 2) complex (alpha, beta), eu, status, success).

    ``forward=True`` adds the forward-solution part of the reference's 9-tuple (:367-393), formed on the device by
    ``dsge_gensys_pencil_full_batched``: ``n_unstable`` (batch,) and, full-size with the valid block leading (slice with
    ``n_unstable[i]``), ``f_mat`` (batch, N, N) complex, ``f_wt`` (batch, N, k) complex, ``y_wt`` (batch, N, N) complex and
    ``loose`` (batch, N, n_eta).  When the columns of some ``pi[i]`` are not orthonormal two launches run (matrices from Pi as
    given, existence / uniqueness codes from an orthonormal basis of its column space; see the comment below)."""
    g0, g1 = _f64(g0, 3), _f64(g1, 3)
    psi, pi = _f64(psi, 3), _f64(pi, 3)
    nb, N, _ = g0.shape
    if g1.shape != (nb, N, N) or psi.shape[:2] != (nb, N) or pi.shape[:2] != (nb, N):
        raise ValueError("g0, g1: (batch, N, N); psi: (batch, N, k); pi: (batch, N, n_eta)")
    k, ne = psi.shape[2], pi.shape[2]
    if c is not None:
        c = _f64(c).reshape(nb, N)
    G1 = np.empty_like(g0)
    Cc = np.empty((nb, N))
    impact = np.empty((nb, N, k))
    gev = np.empty((nb, N, 4))
    eu = np.empty((nb, 3), dtype=np.int32)
    status = np.empty(nb, dtype=np.int32)
    lib = _lib.load()

    def call(fw):
        _lib.check(lib.dsge_gensys_pencil_full_batched_host(_ptr(g0), _ptr(g1), _ptr(c), _ptr(psi), _ptr(pi), nb, N, k, ne,
                                                            float(tol), _ptr(G1), _ptr(Cc), _ptr(impact), _ptr(gev),
                                                            _ptr(eu), _ptr(status),
                                                            None if fw is None else ctypes.addressof(fw)))

    # Pi with orthonormal columns (every gEconpy pencil: [0; I], gensys.py:606-611): one launch, Pi as given.  Any other Pi: the
    # matrices come from a launch on Pi AS GIVEN (for a non-unique solution G1, impact and loose depend on Pi itself, not only on
    # its column space: Phi = (Q1 Pi)(Q2 Pi)^+), the existence / uniqueness codes from a launch on an orthonormal basis of its
    # column space (they are rank decisions, which the device takes with orthonormal columns; include/dsge_hip.h).
    gram = np.einsum("bij,bik->bjk", pi, pi)
    orthonormal = ne == 0 or bool(np.abs(gram - np.eye(ne)).max() <= 1e-14)
    out = {}
    fw = None
    if forward:
        f_mat, y_wt = np.zeros((nb, N, N, 2)), np.zeros((nb, N, N, 2))
        f_wt = np.zeros((nb, N, k, 2))
        loose = np.zeros((nb, N, max(ne, 1)))
        n_unst = np.zeros(nb, dtype=np.int32)
        fw = _lib.GensysForward(_ptr(f_mat), _ptr(f_wt), _ptr(y_wt), _ptr(loose) if ne > 0 else None, _ptr(n_unst), 1)
    elif not orthonormal:
        fw = _lib.GensysForward(None, None, None, None, None, 1)
    call(fw)
    if forward:
        out = dict(f_mat=f_mat[..., 0] + 1j * f_mat[..., 1], f_wt=f_wt[..., 0] + 1j * f_wt[..., 1],
                   y_wt=y_wt[..., 0] + 1j * y_wt[..., 1], loose=loose[:, :, :ne], n_unstable=n_unst)
    if not orthonormal:
        keep = [x.copy() for x in (G1, Cc, impact, gev)]
        call(None)  # eu / status of the orthonormalised launch stay; the matrices of the launch on Pi itself come back
        for dst, src in zip((G1, Cc, impact, gev), keep):
            dst[...] = src
    gev_c = np.stack([gev[..., 0] + 1j * gev[..., 1], gev[..., 2] + 1j * gev[..., 3]], axis=-1)
    out.update(G1=G1, C=Cc, impact=impact, gev=gev_c, eu=eu, status=status, success=status == 0)
    return out


def selection_batched(B, C, D, T, A=None):
    """R = -(C T + B)^-1 D (gEconpy/solvers/shared.py:74-75); with ``A`` also the residual
    ``sum((A + B T + C T T)^2)`` (gEconpy/model/statespace.py:213)."""
    B, C, T = _check_abc(B, C, T)
    D = _f64(D, 3)
    nb, n, _ = B.shape
    k = D.shape[2]
    if D.shape[:2] != (nb, n):
        raise ValueError("D must be (batch, n, k)")
    R = np.empty((nb, n, k))
    resid = None
    if A is not None:
        A = _f64(A, 3)
        resid = np.empty(nb)
    _lib.check(
        _lib.load().dsge_selection_batched_host(_ptr(A), _ptr(B), _ptr(C), _ptr(D), _ptr(T), nb, n, k, _ptr(R), _ptr(resid))
    )
    return (R, resid) if A is not None else R


def selection_adjoints_batched(B, C, T, R, R_bar):
    """Pullback of ``R = -(C T + B)^-1 D`` (gEconpy/solvers/shared.py:74-75) -> ``(B_bar, C_bar, D_bar, T_bar)``:
    ``G = -(C T + B)^-T R_bar``, ``D_bar = G``, ``B_bar = G R'``, ``C_bar = G R' T'``, ``T_bar = C' G R'``."""
    B, C, T = _check_abc(B, C, T)
    R, R_bar = _f64(R, 3), _f64(R_bar, 3)
    nb, n, _ = B.shape
    k = R.shape[2]
    if R.shape != (nb, n, k) or R_bar.shape != (nb, n, k):
        raise ValueError("R and R_bar must be (batch, n, k)")
    Bb, Cb, Tb = np.empty_like(B), np.empty_like(B), np.empty_like(B)
    Db = np.empty_like(R)
    _lib.check(_lib.load().dsge_selection_adjoints_batched_host(_ptr(B), _ptr(C), _ptr(T), _ptr(R), _ptr(R_bar), nb, n, k,
                                                                _ptr(Bb), _ptr(Cb), _ptr(Db), _ptr(Tb)))
    return Bb, Cb, Db, Tb


def policy_adjoints_batched(B, C, T, T_bar):
    """``(A_bar, B_bar, C_bar, status)``: the reverse-mode sensitivities of
    ``o1_policy_function_adjoints`` (gEconpy/solvers/shared.py:12-71) for a batch of draws."""
    B, C, T = _check_abc(B, C, T)
    T_bar = _f64(T_bar, 3)
    nb, n, _ = B.shape
    Ab, Bb, Cb = np.empty_like(B), np.empty_like(B), np.empty_like(B)
    status = np.empty(nb, dtype=np.int32)
    _lib.check(_lib.load().dsge_policy_adjoints_batched_host(_ptr(B), _ptr(C), _ptr(T), _ptr(T_bar), nb, n, _ptr(Ab),
                                                             _ptr(Bb), _ptr(Cb), _ptr(status)))
    return Ab, Bb, Cb, status


def policy_norms_batched(A, B, C, D, T, R, state_mask):
    """``deterministic_norm`` and ``stochastic_norm`` of gEconpy/model/statespace.py:1181-1204 for a
    batch of draws; ``state_mask`` is the boolean vector ``tm1_idx & t_idx`` (:1186-1193)."""
    A, B, C = _check_abc(A, B, C)
    T, R, D = _f64(T, 3), _f64(R, 3), _f64(D, 3)
    nb, n, _ = A.shape
    k = D.shape[2]
    mask = np.ascontiguousarray(np.asarray(state_mask) != 0, dtype=np.int32)
    if mask.shape != (n,):
        raise ValueError("state_mask must have one entry per variable")
    det = np.empty(nb)
    sto = np.empty(nb)
    _lib.check(_lib.load().dsge_policy_norms_batched_host(_ptr(A), _ptr(B), _ptr(C), _ptr(D), _ptr(T), _ptr(R),
                                                          _ptr(mask), nb, n, k, _ptr(det), _ptr(sto)))
    return det, sto


def backward_direct_batched(A, B, D):
    """T = (-B)^-1 A, R = -B^-1 D (gEconpy/solvers/backward_looking.py:102-134)."""
    A, B = _f64(A, 3), _f64(B, 3)
    D = _f64(D, 3)
    nb, n, _ = A.shape
    k = D.shape[2]
    T = np.empty_like(A)
    R = np.empty((nb, n, k))
    _lib.check(_lib.load().dsge_backward_direct_batched_host(_ptr(A), _ptr(B), _ptr(D), nb, n, k, _ptr(T), _ptr(R)))
    return T, R


def _q_mode(Q, nb, k):
    """Infer the covariance layout from the shape (ambiguous only when batch == k)."""
    Q = _f64(Q)
    if Q.shape == (k,):
        return Q, _lib.Q_DIAG_SHARED
    if Q.shape == (nb, k, k):
        return Q, _lib.Q_FULL_BATCHED
    if Q.ndim == 2 and nb != k:
        if Q.shape == (nb, k):
            return Q, _lib.Q_DIAG_BATCHED
        if Q.shape == (k, k):
            return Q, _lib.Q_FULL_SHARED
    raise ValueError(f"cannot infer the layout of Q with shape {Q.shape} (batch={nb}, k={k}); pass q_mode")


def _resolve_q(Q, q_mode, nb, k):
    if q_mode is None:
        return _q_mode(Q, nb, k)
    modes = {"diag": _lib.Q_DIAG_SHARED, "diag_batched": _lib.Q_DIAG_BATCHED, "full": _lib.Q_FULL_SHARED,
             "full_batched": _lib.Q_FULL_BATCHED}
    code = modes[q_mode] if isinstance(q_mode, str) else int(q_mode)
    Q = _f64(Q)
    want = {0: (k,), 1: (nb, k), 2: (k, k), 3: (nb, k, k)}[code]
    if Q.shape != want:
        raise ValueError(f"Q has shape {Q.shape}, q_mode needs {want}")
    return Q, code


def lyapunov_batched(T, R, Q, q_mode=None):
    """P0 = solve_discrete_lyapunov(T, R Q R') (gEconpy/model/statespace.py:814-815) ->
    (P0, RQR, status)."""
    T, R = _f64(T, 3), _f64(R, 3)
    nb, m, _ = T.shape
    k = R.shape[2]
    Q, code = _resolve_q(Q, q_mode, nb, k)
    P0 = np.empty_like(T)
    RQR = np.empty_like(T)
    status = np.empty(nb, dtype=np.int32)
    _lib.check(_lib.load().dsge_lyapunov_batched_host(_ptr(T), _ptr(R), _ptr(Q), code, nb, m, k, _ptr(P0), _ptr(RQR), _ptr(status)))
    return P0, RQR, status


def autocorrelation_matrices_batched(T, R, Q, n_lags=10, lag_step=1, Z=None, Hdiag=None, correlation=True, q_mode=None,
                                     return_sigma=False):
    """Per-draw autocorrelation (or autocovariance) matrices at lags 0..n_lags -- the batch version of
    ``_compute_autocovariance_matrix`` (gEconpy/model/statistics/covariance.py:133-161; note that one returns lags
    0..n_lags-1) and of the graph ``sample_autocorrelation_matrices`` evaluates per posterior draw
    (gEconpy/model/statespace.py:1262-1300; ``observed=True`` <=> ``Z`` given, measurement-error variances
    ``Hdiag`` enter the lag-0 matrix).  Returns ``acf[batch, n_lags+1, dim, dim]`` (+ status [, Sigma])."""
    T, R = _f64(T, 3), _f64(R, 3)
    nb, m, _ = T.shape
    k = R.shape[2]
    Q, code = _resolve_q(Q, q_mode, nb, k)
    p = 0
    if Z is not None:
        Z = _f64(Z, 2)
        p = Z.shape[0]
        if Z.shape[1] != m:
            raise ValueError("Z must be (p, m)")
        Hdiag = None if Hdiag is None else _f64(Hdiag, 1)
    dim = p if Z is not None else m
    out = np.empty((nb, n_lags + 1, dim, dim))
    sigma = np.empty((nb, m, m)) if return_sigma else None
    status = np.empty(nb, dtype=np.int32)
    _lib.check(
        _lib.load().dsge_autocorrelation_batched_host(_ptr(T), _ptr(R), _ptr(Q), code, _ptr(Z), _ptr(Hdiag), nb, m, k, p,
                                                      int(n_lags), int(lag_step), int(bool(correlation)), _ptr(out),
                                                      _ptr(sigma), _ptr(status))
    )
    return (out, status, sigma) if return_sigma else (out, status)


def _obs_args(Z, d, Hdiag, nb, p, m):
    Z = _f64(Z)
    if Z.shape == (p, m):
        zb = 0
    elif Z.shape == (nb, p, m):
        zb = 1
    else:
        raise ValueError(f"Z must be (p, m) or (batch, p, m); got {Z.shape}")

    def vec(x, name):
        if x is None:
            return None, 0
        x = _f64(x)
        if x.shape == (p,):
            return x, 0
        if x.shape == (nb, p):
            return x, 1
        raise ValueError(f"{name} must be (p,) or (batch, p); got {x.shape}")

    d, db = vec(d, "d")
    Hdiag, hb = vec(Hdiag, "Hdiag")
    return Z, zb, d, db, Hdiag, hb


def state_hint(M):
    """Performance hint: number of columns of ``M`` (A or T, any leading batch axes) that are
    non-zero in at least one draw -- the model's state variables."""
    M = np.asarray(M)
    return int(np.count_nonzero(np.any(M != 0, axis=tuple(range(M.ndim - 1)))))


def static_hint(A, C):
    """Performance hint (``dsge_options.n_static_hint``): number of variables whose columns of ``A`` AND ``C`` are
    exactly zero in every draw -- the static variables the cycle-reduction launcher deflates."""
    A, C = np.asarray(A), np.asarray(C)
    n = A.shape[-1]
    return int(np.count_nonzero(~(np.any(A.reshape(-1, n) != 0, axis=0) | np.any(C.reshape(-1, n) != 0, axis=0))))


def selector_hint(Z):
    """Performance hint: 1 if every row of Z has exactly one non-zero entry, in distinct columns."""
    Z2 = np.asarray(Z).reshape(-1, Z.shape[-2], Z.shape[-1])
    nz = Z2 != 0
    one_per_row = np.all(nz.sum(axis=2) == 1)
    distinct = np.all(nz.sum(axis=1) <= 1)
    return int(bool(one_per_row and distinct))


def bk_eigenvalues_batched(A, B, C, tol=1e-8):
    """Batched ``compute_bk_eigenvalues`` (gEconpy/model/perturbation.py:412-445): generalized eigenvalues of the
    Sims pencil sorted by ascending modulus, plus the two counts ``check_bk_condition`` (:448-565) compares.

    Returns dict(real, imag (batch, 2n; the first ``n_eig[i]`` entries of a row are valid), n_eig, n_forward,
    n_unstable, satisfied, status)."""
    A, B, C = _check_abc(A, B, C)
    nb, n, _ = A.shape
    re = np.empty((nb, 2 * n))
    im = np.empty((nb, 2 * n))
    ne, nf, nu, st = (np.empty(nb, dtype=np.int32) for _ in range(4))
    _lib.check(
        _lib.load().dsge_bk_eigenvalues_batched_host(_ptr(A), _ptr(B), _ptr(C), nb, n, float(tol), _ptr(re), _ptr(im),
                                                     _ptr(ne), _ptr(nf), _ptr(nu), _ptr(st))
    )
    return dict(real=re, imag=im, n_eig=ne, n_forward=nf, n_unstable=nu, satisfied=(st == 0) & (nf == nu), status=st)


def kalman_logp_batched(T, R, Q, Z, y, d=None, Hdiag=None, q_mode=None, status=None,
                        jitter=JITTER_DEFAULT, missing_fill_value=MISSING_FILL, n_state_hint=None,
                        z_selector_hint=None, options=None, filter_type="standard"):
    """Per-draw Kalman log-likelihood (the filter DSGEStateSpace hands to PyMC,
    gEconpy/model/statespace.py:1151-1157) -> (logp, status).  ``filter_type``: see ``check_filter_type``."""
    check_filter_type(filter_type)
    T, R = _f64(T, 3), _f64(R, 3)
    y = _f64(y, 2)
    nb, m, _ = T.shape
    if m > _lib.MAX_N:
        # More than 64 variables: the filter of the model restricted to F = {variables with a non-zero column of T in some
        # draw} u {observed variables} is the same filter (x_t[F] depends on x_{t-1}[F] only, y_t on x_t[F] only), and the
        # device kernels take |F| <= 64 -- the gather the fused entry point does on the device (csrc/dsge_big.hpp), here on
        # the host because T and R arrive as host arrays.  Exact zeros only: a T whose non-state columns carry rounding noise
        # keeps them, and the call fails loudly instead of dropping them.
        Zf = _f64(Z)
        # (a draw that arrives with a non-zero status is skipped by the filter anyway, and a failed solve may have left NaN in
        #  its T -- NaN compares non-zero --: neither may widen F for the whole batch)
        live = np.ones(nb, dtype=bool) if status is None else (np.asarray(status).reshape(-1) == 0)
        Tl = T[live]
        keep = np.any(np.isfinite(Tl) & (Tl != 0.0), axis=(0, 1)) | np.any(Zf.reshape(-1, m) != 0.0, axis=0)
        idx = np.flatnonzero(keep)
        if idx.size > _lib.MAX_N:
            raise _lib.DsgeTooLargeError(f"kalman_logp_batched: {idx.size} state and observed variables, the filter kernels "
                                         f"take at most {_lib.MAX_N}")
        T = np.ascontiguousarray(T[:, idx][:, :, idx])
        R = np.ascontiguousarray(R[:, idx])
        Z = np.ascontiguousarray(Zf[..., idx])
        m = idx.size
        n_state_hint = None
    k = R.shape[2]
    T_len, p = y.shape
    Q, code = _resolve_q(Q, q_mode, nb, k)
    Z, zb, d, db, Hdiag, hb = _obs_args(Z, d, Hdiag, nb, p, m)
    st = np.zeros(nb, dtype=np.int32) if status is None else np.ascontiguousarray(status, dtype=np.int32).copy()
    logp = np.empty(nb)
    ns = state_hint(T) if n_state_hint is None else int(n_state_hint)
    zs = selector_hint(Z) if z_selector_hint is None else int(z_selector_hint)
    with _lib.options_scope(options):
        _lib.check(
            _lib.load().dsge_kalman_logp_batched_host(
                _ptr(T), _ptr(R), _ptr(Q), code, _ptr(Z), zb, _ptr(d), db, _ptr(Hdiag), hb, _ptr(y), nb, m, k, p, T_len,
                float(jitter), float(missing_fill_value), ns, zs, _ptr(logp), _ptr(st)
            )
        )
    return logp, st


def kalman_filter_outputs_batched(T, R, Q, Z, y, d=None, Hdiag=None, q_mode=None, status=None, jitter=JITTER_DEFAULT,
                                 missing_fill_value=MISSING_FILL, full_covariances=False, options=None):
    """Per-step filter outputs for a batch of draws -- what ``save_kalman_filter_outputs_in_idata=True`` stores
    (gEconpy/model/statespace.py:1145, 1151-1157): dict(ll (batch, T_len), predicted_states / filtered_states (batch, T_len, m),
    predicted_covs / filtered_covs: the diagonals (batch, T_len, m), or the matrices (batch, T_len, m, m) with
    ``full_covariances``, status).  ``ll.sum(axis=1)`` is the log-likelihood of ``kalman_logp_batched``."""
    T, R = _f64(T, 3), _f64(R, 3)
    y = _f64(y, 2)
    nb, m, _ = T.shape
    k = R.shape[2]
    T_len, p = y.shape
    Q, code = _resolve_q(Q, q_mode, nb, k)
    Z, zb, d, db, Hdiag, hb = _obs_args(Z, d, Hdiag, nb, p, m)
    st = np.zeros(nb, dtype=np.int32) if status is None else np.ascontiguousarray(status, dtype=np.int32).copy()
    cshape = (nb, T_len, m, m) if full_covariances else (nb, T_len, m)
    out = dict(ll=np.empty((nb, T_len)), predicted_states=np.empty((nb, T_len, m)), filtered_states=np.empty((nb, T_len, m)),
               predicted_covs=np.empty(cshape), filtered_covs=np.empty(cshape))
    with _lib.options_scope(options):  # (the filter conventions: _lib.filter_conventions)
        _lib.check(
            _lib.load().dsge_kalman_filter_outputs_batched_host(
                _ptr(T), _ptr(R), _ptr(Q), code, _ptr(Z), zb, _ptr(d), db, _ptr(Hdiag), hb, _ptr(y), nb, m, k, p, T_len, float(jitter),
                float(missing_fill_value), _ptr(out["ll"]), _ptr(out["predicted_states"]), _ptr(out["filtered_states"]),
                _ptr(out["predicted_covs"]), _ptr(out["filtered_covs"]), int(bool(full_covariances)), _ptr(st)
            )
        )
    out["status"] = st
    return out


def solve_kalman_logp_batched(A, B, C, D, Q, Z, y, d=None, Hdiag=None, q_mode=None, solver="cycle_reduction",
                              tol=1e-6, max_iter=50, jitter=JITTER_DEFAULT, missing_fill_value=MISSING_FILL,
                              return_policy=False, n_state_hint=None, z_selector_hint=None, n_lead_hint=None,
                              options=None, add_solver_success_check=True, filter_type="standard"):
    """One fused evaluation per draw: A,B,C,D -> T,R -> P0 -> logp.  ``tol``/``max_iter``
    default to what ``DSGEStateSpace.configure`` passes (statespace.py:835-836).
    ``filter_type`` (build.py:577): only "standard" is built, the others raise (``check_filter_type``).
    ``add_solver_success_check=False`` (the reference's default, statespace.py:1148): a draw whose cycle reduction fails carries
    ``T = 0`` on and gets the finite log-likelihood of that system (``DSGE_SOLVER_FLAG_ZERO_T_ON_FAILURE``); the default here
    is the safe one, ``-inf``.
    ``options``: per-call kernel-variant switches (dict of ``dsge_options`` fields or ``_lib.Options``).
    Returns dict(logp, status[, T, R, resid, n_iter])."""
    check_filter_type(filter_type)
    A, B, C = _check_abc(A, B, C)
    D = _f64(D, 3)
    y = _f64(y, 2)
    nb, n, _ = A.shape
    k = D.shape[2]
    T_len, p = y.shape
    Q, code = _resolve_q(Q, q_mode, nb, k)
    Z, zb, d, db, Hdiag, hb = _obs_args(Z, d, Hdiag, nb, p, n)
    logp = np.empty(nb)
    status = np.empty(nb, dtype=np.int32)
    T = R = resid = n_iter = None
    if return_policy:
        T = np.empty_like(A)
        R = np.empty((nb, n, k))
        resid = np.empty(nb)
        n_iter = np.empty(nb, dtype=np.int32)
    ns = state_hint(A) if n_state_hint is None else int(n_state_hint)
    zs = selector_hint(Z) if z_selector_hint is None else int(z_selector_hint)
    nl = (lead_hint(C, tol) if solver == "gensys" else 0) if n_lead_hint is None else int(n_lead_hint)
    op, _keep = _lib.opt_ptr(options)
    _lib.check(
        _lib.load().dsge_solve_kalman_logp_batched_host_opt(
            op, _ptr(A), _ptr(B), _ptr(C), _ptr(D), _ptr(Q), code, _ptr(Z), zb, _ptr(d), db, _ptr(Hdiag), hb, _ptr(y), nb,
            n, k, p, T_len, _lib.SOLVER_CODES[solver] | (0 if add_solver_success_check else _lib.SOLVER_FLAG_ZERO_T_ON_FAILURE),
            float(tol), int(max_iter), float(jitter),
            float(missing_fill_value), ns, zs, nl, _ptr(logp), _ptr(status), _ptr(T), _ptr(R), _ptr(resid), _ptr(n_iter)
        )
    )
    out = dict(logp=logp, status=status)
    if return_policy:
        out.update(T=T, R=R, resid=resid, n_iter=n_iter)
    return out


def solve_kalman_logp_grad_batched(A, B, C, D, q, Z, y, d=None, Hdiag=None, solver="cycle_reduction", tol=1e-6, max_iter=50,
                                   jitter=JITTER_DEFAULT, missing_fill_value=MISSING_FILL, n_filter_hint=None,
                                   n_lead_hint=None, options=None, Q=None, dense_z=None, return_Z_bar=False):
    """logp and its reverse-mode gradient per draw (include/dsge_hip.h: dsge_solve_kalman_logp_grad_batched): what
    pytensor autodiff computes for the reference's logp graph, on the device.  ``q``: (k,) or (batch, k) diagonal shock
    variances, or ``q=None, Q=`` a full symmetric shock covariance (k, k) / (batch, k, k) (``full_covariance``,
    statespace.py:247-251); ``Z``: selector design matrix (p, n), p <= 8; n <= 56.
    Returns dict(logp, status, A_bar, B_bar, C_bar, D_bar, q_bar[, d_bar][, h_bar]); with ``Q=`` the key is ``Q_bar``
    (batch, k, k): the cotangent of all k x k entries taken as independent (symmetric).
    A design matrix that is not a selector (observation equations, statespace.py:298-332) takes the dense-Z entry point
    (``dsge_solve_kalman_logp_grad_dense_z_batched``: the observed combinations become p extra variables, n + p <= 56) --
    automatically, or forced with ``dense_z=True``; ``return_Z_bar=True`` adds ``Z_bar`` (batch, p, n), the cotangent of Z."""
    A, B, C = _check_abc(A, B, C)
    D = _f64(D, 3)
    y = _f64(y, 2)
    nb, n, _ = A.shape
    k = D.shape[2]
    T_len, p = y.shape
    if Q is not None:
        if q is not None:
            raise ValueError("pass either q (diagonal variances) or Q (full covariance)")
        q = _f64(Q)
        if q.shape not in ((k, k), (nb, k, k)):
            raise ValueError("Q must be (k, k) or (batch, k, k)")
        qb = 2 + int(q.ndim == 3)
    else:
        q = _f64(q)
        if q.shape not in ((k,), (nb, k)):
            raise ValueError("q must be (k,) or (batch, k) (diagonal shock covariance)")
        qb = int(q.ndim == 2)
    Z, zb, d, db, Hdiag, hb = _obs_args(Z, d, Hdiag, nb, p, n)
    if dense_z is None:
        dense_z = bool(return_Z_bar) or not selector_hint(Z)
    if n_filter_hint is not None:
        ns = int(n_filter_hint)
    elif dense_z:  # the dense entry point wants the number of STATE variables (non-zero columns of A in any draw)
        ns = int(np.count_nonzero(np.any(A.reshape(-1, n) != 0, axis=0)))
    else:  # |S u O|: non-zero columns of A (in any draw) or of Z
        ns = int(np.count_nonzero(np.any(A.reshape(-1, n) != 0, axis=0) | np.any(Z.reshape(-1, n) != 0, axis=0)))
    nl = (lead_hint(C, tol) if solver == "gensys" else 0) if n_lead_hint is None else int(n_lead_hint)
    out = dict(logp=np.empty(nb), status=np.empty(nb, dtype=np.int32), A_bar=np.empty_like(A), B_bar=np.empty_like(A),
               C_bar=np.empty_like(A), D_bar=np.empty_like(D), q_bar=np.empty((nb, k, k) if qb >= 2 else (nb, k)))
    if d is not None:
        out["d_bar"] = np.empty((nb, p))
    if Hdiag is not None:
        out["h_bar"] = np.empty((nb, p))
    if dense_z:
        if return_Z_bar:
            out["Z_bar"] = np.empty((nb, p, n))
        with _lib.options_scope(options):
            _lib.check(
                _lib.load().dsge_solve_kalman_logp_grad_dense_z_batched_host(
                    _ptr(A), _ptr(B), _ptr(C), _ptr(D), _ptr(q), qb, _ptr(Z), zb, _ptr(d), db, _ptr(Hdiag), hb, _ptr(y), nb, n, k,
                    p, T_len, _lib.SOLVER_CODES[solver], float(tol), int(max_iter), float(jitter), float(missing_fill_value), ns,
                    nl, _ptr(out["logp"]), _ptr(out["status"]), _ptr(out["A_bar"]), _ptr(out["B_bar"]), _ptr(out["C_bar"]),
                    _ptr(out["D_bar"]), _ptr(out["q_bar"]), _ptr(out.get("d_bar")), _ptr(out.get("h_bar")), _ptr(out.get("Z_bar"))
                )
            )
    else:
        op, _keep = _lib.opt_ptr(options)
        _lib.check(
            _lib.load().dsge_solve_kalman_logp_grad_batched_host_opt(
                op, _ptr(A), _ptr(B), _ptr(C), _ptr(D), _ptr(q), qb, _ptr(Z), zb, _ptr(d), db, _ptr(Hdiag), hb, _ptr(y), nb, n, k,
                p, T_len, _lib.SOLVER_CODES[solver], float(tol), int(max_iter), float(jitter), float(missing_fill_value), ns, nl,
                _ptr(out["logp"]), _ptr(out["status"]), _ptr(out["A_bar"]), _ptr(out["B_bar"]), _ptr(out["C_bar"]),
                _ptr(out["D_bar"]), _ptr(out["q_bar"]), _ptr(out.get("d_bar")), _ptr(out.get("h_bar"))
            )
        )
    if qb >= 2:
        out["Q_bar"] = out.pop("q_bar")
    return out


def second_order_structure(A, C, Z):
    """Model structure the second-order path takes as index lists (int32): the state variables S (non-zero columns of A in
    any draw), the forward-looking variables L (non-zero columns of C) and the variables the pruned filter retains, U = S
    followed by the observed non-states (non-zero columns of Z)."""
    A = np.asarray(A)
    C = np.asarray(C)
    n = A.shape[-1]
    S = np.flatnonzero(np.any(A.reshape(-1, n) != 0, axis=0))
    Lc = np.flatnonzero(np.any(C.reshape(-1, n) != 0, axis=0))
    obs = np.flatnonzero(np.any(np.asarray(Z).reshape(-1, n) != 0, axis=0))
    U = np.concatenate([S, np.setdiff1d(obs, S)])
    return S.astype(np.int32), Lc.astype(np.int32), U.astype(np.int32)


def second_order_logp_batched(A, B, C, D, hess_idx, hess_val, q, Z, y, d=None, Hdiag=None, solver="cycle_reduction", tol=1e-8,
                              max_iter=1000, jitter=JITTER_DEFAULT, missing_fill_value=MISSING_FILL, return_solution=False,
                              structure=None, options=None):
    """Second-order perturbation + pruned-state-space quasi-likelihood per draw (include/dsge_hip.h:
    ``dsge_second_order_logp_batched``; BASELINE configs[4]).  The reference raises ``NotImplementedError`` for ``order != 1``
    (gEconpy/model/perturbation.py:97-98); this is what replaces the raise.
    ``hess_idx``: (nnz, 3) int32 (equation, z_a <= z_b) sorted by equation, z = [y-; y; y+; u]; ``hess_val``: (batch, nnz);
    ``q``: (k,) or (batch, k) shock variances; ``Z``: (p, n), p <= 8.
    Returns dict(logp, status[, T, R, g_yy (batch, n, s, s), g_yu (batch, n, s, k), g_uu (batch, n, k, k), g_ss (batch, n), S])."""
    A, B, C = _check_abc(A, B, C)
    D = _f64(D, 3)
    y = _f64(y, 2)
    nb, n, _ = A.shape
    k = D.shape[2]
    T_len, p = y.shape
    hess_idx = np.ascontiguousarray(hess_idx, dtype=np.int32).reshape(-1, 3)
    nnz = hess_idx.shape[0]
    hess_val = _f64(hess_val, 2)
    if hess_val.shape != (nb, nnz):
        raise ValueError("hess_val must be (batch, nnz)")
    q = _f64(q)
    if q.shape not in ((k,), (nb, k)):
        raise ValueError("q must be (k,) or (batch, k) (diagonal shock covariance)")
    Z = _f64(Z, 2)
    if Z.shape != (p, n):
        raise ValueError("Z must be (p, n)")
    d = None if d is None else _f64(d, 1)
    Hdiag = None if Hdiag is None else _f64(Hdiag, 1)
    S, Lc, U = second_order_structure(A, C, Z) if structure is None else (np.ascontiguousarray(x, dtype=np.int32) for x in structure)
    s = len(S)
    out = dict(logp=np.empty(nb), status=np.empty(nb, dtype=np.int32))
    T = R = gyy = gyu = guu = gss = None
    if return_solution:
        T, R = np.empty_like(A), np.empty((nb, n, k))
        gyy, gyu, guu, gss = np.empty((nb, n, s, s)), np.empty((nb, n, s, k)), np.empty((nb, n, k, k)), np.empty((nb, n))
    with _lib.options_scope(options):
        _lib.check(
            _lib.load().dsge_second_order_logp_batched_host(
                _ptr(A), _ptr(B), _ptr(C), _ptr(D), _ptr(hess_idx), nnz, _ptr(hess_val), _ptr(q), int(q.ndim == 2), _ptr(Z),
                _ptr(d), _ptr(Hdiag), _ptr(y), nb, n, k, p, T_len, _lib.SOLVER_CODES[solver], float(tol), int(max_iter),
                float(jitter), float(missing_fill_value), _ptr(S), s, _ptr(Lc), len(Lc), _ptr(U), len(U), _ptr(out["logp"]),
                _ptr(out["status"]), _ptr(T), _ptr(R), _ptr(gyy), _ptr(gyu), _ptr(guu), _ptr(gss)
            )
        )
    if return_solution:
        out.update(T=T, R=R, g_yy=gyy, g_yu=gyu, g_uu=guu, g_ss=gss, S=S)
    return out
