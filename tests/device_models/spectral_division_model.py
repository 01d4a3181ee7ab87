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


def certificate(B, C, T, tol=1e-8, lcap=None, scap=None):
    """True iff the device certifies eu = [1, 1, 0] for the solvent T of A + B T + C T^2 = 0."""
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
    return certify_contraction(G[L, :]) and certify_contraction(T[np.ix_(S, S)])


def gensys_by_spectral_division(A, B, C, tol=1e-8):
    """(T, certified): the doubling iteration with the launcher's settings (50 iterations, 1e-9), then the certificate.  Not
    certified: the device hands the draw to the ordered QZ (n <= 64) or reports eu = [-3, -3, 0] (65 .. 96 variables)."""
    T, conv, _ = cycle_reduction_core(A, B, C, 50, 1e-9)
    if not conv:
        return T, False
    return T, certificate(B, C, T, tol)
