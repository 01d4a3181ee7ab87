"""Oracle restatement of Sims' gensys as used by gEconpy (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/gEconpy/solvers/gensys.py:
  * ``gensys_setup``  <- ``_gensys_setup``  (:568-614)
  * ``gensys_core``   <- ``_gensys_core``   (:190-395)
  * ``gensys``        <- ``gensys``         (:398-521)
  * ``solve_policy_function_with_gensys`` <- same name (:617-631)
The reference calls LAPACK zgges+ztgsen / gesdd / getrf / getrs / trtrs through
pytensor's numba helpers; here the same LAPACK routines are reached through scipy.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla

_EPS = np.spacing(1.0)


def gensys_setup(A, B, C, D, tol=1e-8):
    """Build (G0, G1, c, Psi, Pi) of ``G0 w_t = G1 w_{t-1} + c + Psi z_t + Pi eta_t``.

    gensys.py:568-614.  ``w_t = [y_t ; E_t y_{t+1}[lead]]`` where ``lead`` are the
    columns of C whose absolute column sum exceeds ``tol`` (:580-589).  The full
    2n-dim pencil is formed and then cut down to rows/cols ``[0..n) U (n+lead)``
    (:606-609); G0 is the negated Gamma_0 (:611).
    """
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    C = np.asarray(C, dtype=np.float64)
    D = np.asarray(D, dtype=np.float64)
    n = A.shape[0]
    k = D.shape[1]
    lead = np.flatnonzero(np.abs(C).sum(axis=0) > tol)
    nl = lead.size
    N = n + nl

    G0 = np.zeros((N, N))
    G0[:n, :n] = -B
    G0[:n, n:] = -C[:, lead]
    G0[n + np.arange(nl), lead] = 1.0  # -(-I) restricted to the lead rows
    G1 = np.zeros((N, N))
    G1[:n, :n] = A
    G1[n:, n:] = np.eye(nl)
    Psi = np.zeros((N, k))
    Psi[:n] = D
    Pi = np.zeros((N, nl))
    Pi[n:, :] = np.eye(nl)
    c = np.zeros((N, 1))
    return G0, G1, c, Psi, Pi


def _thin_svd_keep(M, realsmall):
    """Thin SVD keeping sigma > realsmall (gensys.py:121-125, 276-280)."""
    u, s, vh = sla.svd(M, full_matrices=False, lapack_driver="gesdd")
    keep = np.flatnonzero(s > realsmall)
    return u[:, keep], s[keep], vh.conj().T[:, keep]


def _rank_svd(M, tol):
    """gensys.py:175-187: rank by singular values; NaN singular values count as 0."""
    if M.shape[0] == 0 or M.shape[1] == 0:
        return 0
    s = sla.svd(M, compute_uv=False, lapack_driver="gesdd")
    return int(np.sum(s > tol))


def gensys_core(g0, g1, c, psi, pi, tol):
    """gensys.py:190-395.  Always returns the 9-tuple; ``eu`` is int64[3]."""
    N = g1.shape[0]
    n_eta = pi.shape[1]
    k = psi.shape[1]
    rs = tol if tol > 0 else _EPS  # :223

    # ordered complex QZ, stable roots (|beta/alpha| < 1) first (:227-235)
    AA, BB, alpha, beta, Qraw, Z = sla.ordqz(
        g0.astype(np.complex128), g1.astype(np.complex128), sort="ouc", output="complex"
    )
    Q = Qraw.conj().T
    ZH = Z.conj().T

    abs_a = np.abs(alpha)
    abs_b = np.abs(beta)
    zxz = bool(np.any((abs_a < rs) & (abs_b < rs)))  # :243-244
    stable = ((abs_b < rs) & (abs_a >= rs)) | ((abs_b >= rs) & (abs_a > abs_b))  # :246
    nu = int(np.sum(~stable))
    ns = N - nu
    eu = np.zeros(3, dtype=np.int64)
    gev = np.column_stack((alpha, beta))

    if zxz:  # :255-265
        eu[:2] = -2
        cz = np.complex128
        return (
            np.zeros((N, N)),
            np.zeros((N, c.shape[1])),
            np.zeros((N, k)),
            np.zeros((nu, nu), cz),
            np.zeros((nu, k), cz),
            np.zeros((N, nu), cz),
            gev,
            eu,
            np.zeros((N, n_eta)),
        )

    Q1, Q2 = Q[:ns], Q[ns:]  # :267
    pic = pi.astype(np.complex128)

    # unstable block: eta_wt = Q2 Pi (:270-280)
    eta2 = Q2 @ pic
    if nu == 0:
        u2 = np.zeros((0, 0), np.complex128)
        d2 = np.zeros(0)
        v2 = np.zeros((n_eta, 0), np.complex128)
    else:
        u2, d2, v2 = _thin_svd_keep(eta2, rs)
    if d2.size >= nu:  # existence (:282-283)
        eu[0] = 1

    # stable block: eta_wt_1 = Q1 Pi (:285-296)
    if nu == N:
        eta1 = np.zeros((0, n_eta), np.complex128)
        u1 = np.zeros((0, 0), np.complex128)
        d1 = np.zeros(0)
        v1 = np.zeros((n_eta, 0), np.complex128)
    else:
        eta1 = Q1 @ pic
        u1, d1, v1 = _thin_svd_keep(eta1, rs)

    # uniqueness (:301-310)
    if v1.shape[0] == 0 or v1.shape[1] == 0:
        unique = True
    else:
        n_loose = _rank_svd(v1 - v2 @ (v2.conj().T @ v1), rs * N)
        eu[2] = n_loose
        unique = n_loose == 0
    if unique:
        eu[1] = 1

    # inner = U2 D2^-1 V2^H V1 D1 U1^H  (:314-320)
    v2h_scaled = v2.conj().T / d2[:, None] if d2.size else v2.conj().T
    u1h_scaled = d1[:, None] * u1.conj().T if d1.size else u1.conj().T
    inner = u2 @ v2h_scaled @ v1 @ u1h_scaled  # nu x ns
    Tmat = np.hstack((np.eye(ns, dtype=np.complex128), -inner.conj().T))  # ns x N (:322)

    G0m = np.zeros((N, N), np.complex128)  # :323-330
    G0m[:ns] = Tmat @ AA
    G0m[ns:, ns:] = np.eye(nu)
    lu_piv = sla.lu_factor(G0m)  # one LU reused for all solves (:333)

    rhs = np.zeros((N, N), np.complex128)
    rhs[:ns] = Tmat @ BB
    G1 = (Z @ sla.lu_solve(lu_piv, rhs) @ ZH).real  # :336-343

    AA22 = AA[ns:, ns:]
    BB22 = BB[ns:, ns:]
    cc = c.astype(np.complex128)
    psic = psi.astype(np.complex128)
    TQ = Tmat @ Q

    if nu == 0:  # :353-357
        c_tail = np.zeros((0, c.shape[1]), np.complex128)
    else:
        c_tail = sla.solve_triangular(AA22 - BB22, Q2 @ cc, lower=False)
    C_out = (Z @ np.vstack((TQ @ cc, c_tail))).real

    rhs_imp = np.zeros((N, k), np.complex128)  # :359-365
    rhs_imp[:ns] = TQ @ psic
    impact = (Z @ sla.lu_solve(lu_piv, rhs_imp)).real

    if nu == 0:  # :367-374
        fmat = np.zeros((0, 0), np.complex128)
        fwt = np.zeros((0, k), np.complex128)
    else:
        fmat = sla.solve_triangular(BB22, AA22, lower=False)
        fwt = -sla.solve_triangular(BB22, Q2 @ psic, lower=False)

    eye_cols = np.zeros((N, nu), np.complex128)  # :376-381
    eye_cols[ns + np.arange(nu), np.arange(nu)] = 1.0
    ywt = Z @ sla.lu_solve(lu_piv, eye_cols)

    rhs_loose = np.zeros((N, n_eta), np.complex128)  # :383-393
    rhs_loose[:ns] = eta1 @ (np.eye(n_eta) - v2 @ v2.conj().T)
    loose = (Z @ sla.lu_solve(lu_piv, rhs_loose)).real

    return G1, C_out, impact, fmat, fwt, ywt, gev, eu, loose


def gensys(g0, g1, c, psi, pi, div=None, tol=1e-8, return_all_matrices=True):
    """gensys.py:398-521 (``div`` accepted and ignored, :502)."""
    del div
    tol_eff = tol if tol is not None and tol > 0 else _EPS
    f = lambda x: np.ascontiguousarray(x, dtype=np.float64)  # noqa: E731
    out = gensys_core(f(g0), f(g1), f(c), f(psi), f(pi), tol_eff)
    eu = [int(x) for x in out[7]]
    if eu[0] == -2 and eu[1] == -2:  # :515-516
        return None, None, None, None, None, None, None, eu, None
    if not return_all_matrices:
        return out[0], eu
    return out[:7] + (eu, out[8])


def solve_policy_function_with_gensys(A, B, C, D, tol=1e-8, return_all_matrices=True):
    """gensys.py:617-631."""
    g0, g1, c, psi, pi = gensys_setup(A, B, C, D, tol)
    return gensys(g0, g1, c, psi, pi, tol=tol, return_all_matrices=return_all_matrices)


def gensys_T_success(A, B, C, D, tol=1e-8):
    """What ``GensysWrapper.perform`` hands to the graph (gensys.py:657-666):
    ``T = G1[:n,:n]`` and ``success = eu[0]==1 and eu[1]==1``.  On coincident zeros the
    njit path (:702-710) yields a zero T; the numpy path would fail on ``None``."""
    n = np.asarray(A).shape[0]
    g0, g1, c, psi, pi = gensys_setup(A, B, C, D, tol)
    out = gensys_core(g0, g1, c, psi, pi, tol if tol > 0 else _EPS)
    eu = out[7]
    return np.ascontiguousarray(out[0][:n, :n]), bool(eu[0] == 1 and eu[1] == 1), eu


def compute_bk_eigenvalues(A, B, C, D, tol=1e-8):
    """``compute_bk_eigenvalues`` (gEconpy/model/perturbation.py:412-445): ordered QZ of ``(-G0, G1)`` from
    ``gensys_setup`` (i.e. of Sims' (Gamma_0, Gamma_1)), ``lambda = diag(BB) / (diag(AA) + tol)``, sorted by ascending
    modulus; returns (real, imag, n_forward)."""
    G0, G1, *_ = gensys_setup(A, B, C, D, tol)
    AA, BB, *_ = sla.ordqz(-G0, G1, sort="ouc", output="complex")
    eig = np.diag(BB) / (np.diag(AA) + tol)
    eig = eig[np.argsort(np.abs(eig))]
    n_forward = int((np.abs(np.asarray(C)).sum(axis=0) > tol).sum())
    return np.real(eig), np.imag(eig), n_forward


def check_bk_condition(A, B, C, D, tol=1e-8):
    """The boolean of ``check_bk_condition`` (perturbation.py:448-565): #{|lambda| > 1} == n_forward.
    Returns (satisfied, n_forward, n_unstable)."""
    re, im, n_forward = compute_bk_eigenvalues(A, B, C, D, tol)
    n_unstable = int((np.sqrt(re**2 + im**2) > 1).sum())
    return n_forward == n_unstable, n_forward, n_unstable
