"""numpy model of gensys by spectral division (csrc/dsge_gensys_doubling.hpp, csrc/dsge_big.hpp::gensys_certify_big_kernel): the
decision rule of the device restated operation by operation -- which draws get eu = [1, 1, 0] from the certificate, which go to
the ordered QZ -- so that the rule itself is validated on the CPU against the oracle's gensys
(gEconpy/solvers/gensys.py:190-395 restated in oracle/gensys_qz.py)."""
import numpy as np

from oracle.cycle_reduction import cycle_reduction_core

MAX_SQUARINGS = 12


def certify_contraction(P):
    """rho(P) < 1 certified: ||P^(2^k)||_F < 1/2 for some k <= 12 (then rho < 2^(-1/2^k) < 1; reach: rho < 0.99983)."""
    if P.size == 0:
        return True
    cur = P.copy()
    for k in range(MAX_SQUARINGS + 1):
        with np.errstate(all="ignore"):
            fro2 = float(np.sum(cur * cur))
        if not np.isfinite(fro2) or not fro2 < 1e300:
            return False
        if fro2 < 0.25:
            return True
        if k == MAX_SQUARINGS:
            break
        with np.errstate(all="ignore"):
            cur = cur @ cur
    return False


GUARD_MARGIN2 = 1.5625  # (1.25)^2: both guards hold with a quarter to spare


def q2pi_singular_values(B, C, T, tol=1e-8):
    """The singular values of gensys's `Q2 @ pi` (gensys.py:270-283) WITHOUT a QZ.  The left deflating subspace of the unstable roots
    of the pencil (G0, G1) of gensys_setup is the row space of X = [N_L, I], N_L = (M^-1)[L, :], M = B + C T:
        X G0 = -G_LL [-T_L, I],   X G1 = [-T_L, I]      (G = M^-1 C)
    so Q2 = W X for the W with W X X' W' = I, and Q2 pi = W (pi selects the last l columns): singular values
    1 / sqrt(1 + sigma_i(N_L)^2), whatever orthonormal basis the QZ returned."""
    csum = np.abs(C).sum(axis=0)
    L = np.flatnonzero(csum > tol)
    M = B + C[:, L] @ T[L, :]
    NL = np.linalg.inv(M)[L]
    return 1.0 / np.sqrt(1.0 + np.linalg.svd(NL, compute_uv=False) ** 2)


def scale_guards(B, C, T, L, tol, power_steps=3):
    """(pass_E, pass_Z) as the device computes them (csrc/dsge_gensys_doubling.hpp, "the scale guards").
    E: ||N_L||_F bounds sigma_max(N_L): existence (all singular values of Q2 pi above tol) and no coincident-zero pair in the
       unstable block.
    Z: no coincident-zero pair in the stable block: |alpha_i| >= 1 / (sqrt(1 + ||T_L||_F^2) ||M^-1 (I + N_L'N_L)^-1/2||_F), and for
       ANY v, with w = N_L'N_L v and tau = |N_L v|^2:  N_L'N_L >= w w'/tau, hence
       ||M^-1 (I + N_L'N_L)^-1/2||_F^2 <= ||M^-1||_F^2 - |M^-1 w|^2 / (tau + |w|^2);  v = the power iteration's vector, started at
       the largest row of N_L."""
    rs = tol if tol > 0 else np.spacing(1.0)
    M = B + C[:, L] @ T[L, :]
    with np.errstate(all="ignore"):
        try:
            Mi = np.linalg.inv(M)
        except np.linalg.LinAlgError:
            return False, False
        NL = Mi[L]
        mi2, nl2, tl2 = float(np.sum(Mi * Mi)), float(np.sum(NL * NL)), float(np.sum(T[L] * T[L]))
        cut = 0.0
        if L.size:
            v = NL[int(np.argmax(np.sum(NL * NL, axis=1)))].copy()
            tau = w2 = 0.0
            w = v
            for _ in range(power_steps):
                nv = float(v @ v)
                v = v / np.sqrt(nv) if nv > 0 else v * 0.0
                t = NL @ v
                tau = float(t @ t)
                w = NL.T @ t
                w2 = float(w @ w)
                v = w
            z = Mi @ w
            den = tau + w2
            cut = float(z @ z) / den if den > 0 else 0.0
        mw2 = max(mi2 - cut, 0.0) + 1e-10 * mi2
        m2 = GUARD_MARGIN2 * rs * rs
        return bool((1.0 + nl2) * m2 < 1.0), bool((1.0 + tl2) * mw2 * m2 < 1.0)


def certificate(B, C, T, tol=1e-8, lcap=None, scap=None, guards=True):
    """True iff the device certifies eu = [1, 1, 0] for the solvent T of A + B T + C T^2 = 0.  guards=False: the round-5 rule
    (no scale guards), kept so that the tests can show what the guards are for."""
    n = B.shape[0]
    csum = np.abs(C).sum(axis=0)
    if np.any((csum > 0.0) & ~(csum > tol)) or not np.all(np.isfinite(csum)):
        return False  # a column gensys drops from the pencil (gensys.py:587) although the iteration used it
    if not np.all(np.isfinite(T)) or not np.max(np.abs(T), initial=0.0) < 1e6:
        return False
    L = np.flatnonzero(csum > tol)
    S = np.flatnonzero(np.any(T != 0.0, axis=0))
    if (lcap is not None and L.size > lcap) or (scap is not None and S.size > scap):
        return False
    M = B + C[:, L] @ T[L, :]
    try:
        with np.errstate(all="ignore"):
            G = np.linalg.solve(M, C[:, L])
    except np.linalg.LinAlgError:
        return False
    if guards and not all(scale_guards(B, C, T, L, tol)):
        return False
    return certify_contraction(G[L, :]) and certify_contraction(T[np.ix_(S, S)])


def gensys_by_spectral_division(A, B, C, tol=1e-8, guards=True):
    """(T, certified): the doubling iteration with the launcher's settings (50 iterations, 1e-9), then the certificate.  Not
    certified: the device hands the draw to the ordered QZ (n <= 64) or reports eu = [-3, -3, 0] (65 .. 96 variables)."""
    T, conv, _ = cycle_reduction_core(A, B, C, 50, 1e-9)
    if not conv:
        return T, False
    return T, certificate(B, C, T, tol, guards=guards)
