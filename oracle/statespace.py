"""Oracle restatement of the state-space assembly + Kalman log-likelihood
(TEST INFRASTRUCTURE ONLY).  *** The recursion is pinned against statsmodels (jitter = 0, complete data:
tests/golden/statsmodels_kalman.npz); the pymc_extras conventions below are restated and UNPINNED ("parity unpinned") until
tests/golden/pymc_extras_kalman.npz exists: tests/golden/make_pymc_extras_golden.py produces it on any machine with
pymc_extras, tests/test_oracle_kalman.py::test_pymc_extras_pin consumes it, ``FilterConventions`` holds the switches. ***

Reference call sites (gEconpy):
  * ``P0 = solve_discrete_lyapunov(T_aug, R Q R', method="bilinear")`` statespace.py:814-815
  * ``a0 = 0`` statespace.py:812;  ``Q = diag(sigma^2)`` statespace.py:240-258
  * ``H = diag(error_sigma^2)`` statespace.py:800-810; pure-selector Z statespace.py:282-296
  * filter: ``PyMCStateSpace.build_statespace_graph`` statespace.py:1151-1157 with
    ``missing_fill_value`` / ``cov_jitter=JITTER_DEFAULT`` (:1143-1144), default
    ``filter_type="standard"`` (gEconpy/model/build.py:577).

The recursion itself lives in third-party ``pymc_extras>=0.12.0``
(``pymc_extras/statespace/filters/kalman_filter.py``: ``BaseFilter.kalman_step``,
``handle_missing_values``, ``predict``, ``StandardFilter.update``) and the Lyapunov solve
in ``pytensor>=3.0.4``; neither is vendored under /root/reference nor installed in this
image (pyproject.toml:43-45).  What follows restates their published algorithm:

    per step:  mask rows of Z/H and entries of y that are missing (NaN or == fill value)
               v = y - (d + Z a);  F = Z P Z' + H + jitter I
               K = P Z' F^-1;  a+ = a + K v
               P+ = sym((I-KZ) P (I-KZ)') + sym(K H K') + jitter I      (Joseph form)
               ll_t = 0 if every entry is missing else -1/2 (p ln 2pi + ln det F + v' F^-1 v)
               a = T a+ + c;  P = sym(T P+ T') + sym(R Q R')
    with sym(X) = (X + X')/2 and jitter = 1e-8 in float64.

Note that ``d`` is *not* masked and ``p`` is the full observation dimension even when
some entries are missing (a masked entry contributes ``ln(jitter) + d_i^2/jitter``);
both mirror upstream and are exercised by the tests.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla

JITTER_DEFAULT = 1e-8  # pymc_extras.statespace.utils.constants.JITTER_DEFAULT, float64
MISSING_FILL = -9999.0  # pymc_extras.statespace.utils.constants.MISSING_FILL
_LN2PI = np.log(2.0 * np.pi)


class FilterConventions:
    """The third-party conventions of the "standard" filter that this oracle RESTATES and cannot pin in this image
    (pymc_extras is not installed).  Every field is a switch, so that the day a real install produces
    tests/golden/pymc_extras_kalman.npz (tests/golden/make_pymc_extras_golden.py) a mismatch is a one-line change of
    ``DEFAULT_CONVENTIONS`` here -- tests/test_oracle_kalman.py::test_pymc_extras_pin then names the combination that matches --
    and a CONFIGURATION of the product: since ABI 8 the device kernels take the same switches at run time (``dsge_options.ll_constant
    / jitter_F / jitter_P / mask_d / joseph``; ``geconpy_amd._lib.filter_conventions`` has this constructor's keyword names), and
    tests/test_gpu_conventions.py holds every kernel to every combination of them.

      ll_constant     "p": p ln 2pi with p the FULL observation dimension, also under missing entries (restated default);
                      "observed": (#observed entries) ln 2pi;  "one": a single ln 2pi (older upstream ``StandardFilter.update``)
      jitter_on_F     jitter I added to F = Z P Z' + H
      jitter_on_P     jitter I added to the filtered covariance P+
      mask_d          observation intercept zeroed on missing entries (default: d is NOT masked)
      joseph          Joseph-form covariance update (default) instead of P+ = P - K F K'
    """

    def __init__(self, ll_constant="p", jitter_on_F=True, jitter_on_P=True, mask_d=False, joseph=True):
        if ll_constant not in ("p", "observed", "one"):
            raise ValueError(ll_constant)
        self.ll_constant, self.jitter_on_F, self.jitter_on_P = ll_constant, bool(jitter_on_F), bool(jitter_on_P)
        self.mask_d, self.joseph = bool(mask_d), bool(joseph)

    def __repr__(self):
        return (f"FilterConventions(ll_constant={self.ll_constant!r}, jitter_on_F={self.jitter_on_F}, "
                f"jitter_on_P={self.jitter_on_P}, mask_d={self.mask_d}, joseph={self.joseph})")


DEFAULT_CONVENTIONS = FilterConventions()


def solve_discrete_lyapunov(T, RQR, method="bilinear"):
    """X = T X T' + RQR (statespace.py:814-815; ``"bilinear"`` default, ``"direct"`` if
    ``use_direct_lyapunov``)."""
    return sla.solve_discrete_lyapunov(T, RQR, method=method)


def _sym_quad(A, B):
    out = A @ B @ A.T
    return 0.5 * (out + out.T)


def kalman_filter_logp(
    y,
    T,
    R,
    Q,
    Z,
    H=None,
    d=None,
    c=None,
    a0=None,
    P0=None,
    jitter=JITTER_DEFAULT,
    missing_fill_value=MISSING_FILL,
    return_per_step=False,
    return_states=False,
    conventions=None,
):
    """Standard Kalman filter log-likelihood, ``sum_t ll_t``.  ``conventions``: the third-party
    switches (``FilterConventions``; ``None`` = ``DEFAULT_CONVENTIONS``, what the device kernels implement).

    y : (T_len, p) data (NaN or ``missing_fill_value`` marks a missing entry);
    T : (m, m); R : (m, k); Q : (k, k); Z : (p, m); H : (p, p) or None (= 0);
    d : (p,) or None; c : (m,) or None; a0 : (m,) or None (= 0); P0 : (m, m) or None
    (= stationary covariance).
    """
    y = np.atleast_2d(np.asarray(y, dtype=np.float64))
    m = T.shape[0]
    p = Z.shape[0]
    H = np.zeros((p, p)) if H is None else np.asarray(H, dtype=np.float64)
    d = np.zeros(p) if d is None else np.asarray(d, dtype=np.float64)
    c = np.zeros(m) if c is None else np.asarray(c, dtype=np.float64)
    a = np.zeros(m) if a0 is None else np.asarray(a0, dtype=np.float64).copy()
    RQR = R @ Q @ R.T
    P = solve_discrete_lyapunov(T, RQR) if P0 is None else np.asarray(P0, dtype=np.float64).copy()
    RQR_sym = 0.5 * (RQR + RQR.T)
    eye_m = np.eye(m)
    eye_p = np.eye(p)

    cv = DEFAULT_CONVENTIONS if conventions is None else conventions
    jit_F = jitter if cv.jitter_on_F else 0.0
    jit_P = jitter if cv.jitter_on_P else 0.0
    ll = np.zeros(y.shape[0])
    states = dict(a_pred=[], a_filt=[], P_pred=[], P_filt=[])
    for t in range(y.shape[0]):
        states["a_pred"].append(a.copy())
        states["P_pred"].append(P.copy())
        yt = y[t]
        miss = np.isnan(yt) | (yt == missing_fill_value)
        W = np.diag((~miss).astype(np.float64))
        Zm = W @ Z
        Hm = W @ H
        ym = np.where(miss, 0.0, yt)

        v = ym - ((np.where(miss, 0.0, d) if cv.mask_d else d) + Zm @ a)
        PZt = P @ Zm.T
        F = Zm @ PZt + Hm + jit_F * eye_p
        K = np.linalg.solve(F.T, PZt.T).T
        IKZ = eye_m - K @ Zm
        a_f = a + K @ v
        if cv.joseph:
            P_f = _sym_quad(IKZ, P) + _sym_quad(K, Hm) + jit_P * eye_m
        else:
            P_f = P - K @ F @ K.T
            P_f = 0.5 * (P_f + P_f.T) + jit_P * eye_m
        if miss.all():
            ll[t] = 0.0
        else:
            inner = v @ np.linalg.solve(F, v)
            n_const = {"p": p, "observed": int((~miss).sum()), "one": 1}[cv.ll_constant]
            ll[t] = -0.5 * (n_const * _LN2PI + np.log(np.linalg.det(F)) + inner)
        states["a_filt"].append(a_f.copy())
        states["P_filt"].append(P_f.copy())
        a = T @ a_f + c
        P = _sym_quad(T, P_f) + RQR_sym
    total = float(ll.sum())
    if return_states:  # (per-step outputs: what save_kalman_filter_outputs_in_idata stores, statespace.py:1145)
        return total, ll, {k_: np.array(v_) for k_, v_ in states.items()}
    return (total, ll) if return_per_step else total


def solve_kalman_logp(
    A,
    B,
    C,
    D,
    Q,
    Z,
    y,
    H=None,
    d=None,
    solver="cycle_reduction",
    tol=1e-8,
    max_iter=1000,
    jitter=JITTER_DEFAULT,
    missing_fill_value=MISSING_FILL,
    inv_var_order=None,
    add_solver_success_check=True,
    conventions=None,
):
    """One full evaluation: A,B,C,D -> T,R -> P0 -> logp (SURVEY.md §3 A hot loop).

    Mirrors ``DSGEStateSpace._setup_policy_matrices`` (statespace.py:197-222) followed by
    ``make_symbolic_graph`` (:781-820) without augmentation, then the filter.  A failed
    solve gives ``logp = -inf`` (what the Potentials at :1206-1215 do to the model logp).
    Returns dict(logp, T, R, resid, success, n_iter, P0).
    ``add_solver_success_check=False`` is the reference's DEFAULT graph (statespace.py:1148, 1210-1215: no Potential on the
    policy residual): a failed cycle reduction hands on ``T = 0`` (cycle_reduction.py:181) and the log-likelihood of that
    system -- finite -- is what the model sees.
    """
    from .cycle_reduction import cycle_reduction_core
    from .gensys_qz import gensys_T_success
    from .shared import compute_selection_matrix, policy_residual

    n_iter = 0
    if solver == "gensys":
        Tm, ok, _eu = gensys_T_success(A, B, C, D, tol)
    elif solver == "cycle_reduction":
        Tm, ok, n_iter = cycle_reduction_core(A, B, C, max_iter, tol)
    elif solver == "scan_cycle_reduction":  # statespace.py:205-207: no success flag, T from the last iterate
        from .cycle_reduction import scan_cycle_reduction

        Tm, n_iter = scan_cycle_reduction(A, B, C, max_iter, tol)
        ok = bool(np.all(np.isfinite(Tm)))
    elif solver == "backward_direct":
        Tm, ok = np.linalg.solve(-B, A), True
    else:
        raise ValueError(solver)
    out = {"T": Tm, "success": bool(ok), "n_iter": n_iter}
    if not ok and not (solver == "cycle_reduction" and not add_solver_success_check):
        out.update(logp=-np.inf, R=np.zeros_like(D), resid=np.inf, P0=None)
        return out
    Rm = compute_selection_matrix(B, C, D, Tm)
    out["resid"] = policy_residual(A, B, C, Tm)
    if inv_var_order is not None:  # statespace.py:217-220
        Tm = Tm[inv_var_order][:, inv_var_order]
        Rm = Rm[inv_var_order]
    P0 = solve_discrete_lyapunov(Tm, Rm @ Q @ Rm.T)
    out["logp"] = kalman_filter_logp(
        y, Tm, Rm, Q, Z, H=H, d=d, P0=P0, jitter=jitter, missing_fill_value=missing_fill_value, conventions=conventions
    )
    out.update(T=Tm, R=Rm, P0=P0)
    return out


def autocorrelation_matrices(T, R, Q, n_lags=10, lag_step=1, Z=None, H=None, correlation=True):
    """Lags 0..n_lags of the model-implied autocorrelation, following the graph of
    ``DSGEStateSpace.sample_autocorrelation_matrices`` (statespace.py:1262-1300): ``Sigma = dlyap(T, R Q R')``,
    ``T_step = T^lag_step``, ``G_k = T_step^k Sigma`` (observed: ``Z G_k Z'``, lag 0 ``+ H``), normalised by
    ``std = sqrt(diag(G_0))``.  With ``lag_step=1, Z=None`` the first ``n_lags`` matrices are exactly
    ``_compute_autocovariance_matrix(T, Sigma, n_lags, correlation)`` (covariance.py:133-161)."""
    Sigma = solve_discrete_lyapunov(T, R @ Q @ R.T)
    T_step = np.linalg.matrix_power(T, lag_step)
    powers = [np.eye(T.shape[0])]
    for _ in range(n_lags):
        powers.append(powers[-1] @ T_step)  # statespace.py:1279-1286: prev @ mat
    T_powers = np.stack(powers)
    if Z is not None:
        autocov = (Z @ (T_powers @ Sigma)) @ Z.T
        autocov[0] = Z @ Sigma @ Z.T + (0.0 if H is None else H)
    else:
        autocov = T_powers @ Sigma
        autocov[0] = Sigma
    if correlation:
        std = np.sqrt(np.diag(autocov[0]))
        autocov = autocov / np.outer(std, std)[None]
    return autocov
