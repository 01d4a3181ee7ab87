"""Systems on which gensys's verdict depends on the SCALE of the equations (round 6).  The reference
decides by absolute tolerances: a QZ diagonal pair with |alpha|, |beta| < tol is "coincident zeros" (gensys.py:243-265), existence is
the rank of Q2 pi by singular values > tol (:276-283), a column of C enters the pencil iff sum|C_ij| > tol (:587).  A regular model
with one equation multiplied by 1e-9, the whole system multiplied by 1e-8, or one variable measured in units of 1e9 therefore FAILS
in the reference; the certificate of csrc/dsge_gensys_doubling.hpp must not accept it.  Shared by tests/test_device_models.py (CPU
model of the certificate against the oracle) and tests/test_gpu_gensys_doubling.py (device against the oracle)."""
import numpy as np

from geconpy_amd import workloads as wl


def scaled_system(rng, kind, n=None, log10_lo=-11.0, log10_hi=-5.0):
    """One SW-shaped system (random small size unless n is given) with a scale defect of the given kind:
    "row": one equation multiplied by e; "global": every equation; "col": one variable's columns of A, B, C; "regular": none.
    Returns (A, B, C, D, e)."""
    if n is None:
        n = int(rng.integers(4, 26))
    ns = int(rng.integers(1, max(2, n // 2)))
    nl = int(rng.integers(1, max(2, n // 3)))
    k = int(rng.integers(1, min(n, 4) + 1))
    A, B, C, D, _ = wl.sw_shaped_system(int(rng.integers(1 << 30)), n=n, n_state=ns, n_lead=nl, k=k)
    e = 10.0 ** rng.uniform(log10_lo, log10_hi)
    A, B, C, D = A.copy(), B.copy(), C.copy(), D.copy()
    if kind == "row":
        r = int(rng.integers(n))
        for X in (A, B, C, D):
            X[r] *= e
    elif kind == "global":
        for X in (A, B, C, D):
            X *= e
    elif kind == "col":
        j = int(rng.integers(n))
        for X in (A, B, C):
            X[:, j] *= e
    elif kind != "regular":
        raise ValueError(kind)
    return A, B, C, D, e


def existence_sweep_system(seed, target, n=14, n_state=5, n_lead=4, k=3):
    """A regular system whose smallest singular value of gensys's Q2 pi equals `target` (to rounding): equation r is scaled until
    1 / sqrt(1 + sigma_max(N_L)^2) = target (N_L = lead rows of (B + C T)^-1; tests/device_models/spectral_division_model.py
    ::q2pi_singular_values).  Scaling row r by e multiplies column r of M^-1 by 1 / e, so sigma_max(N_L) is monotone in 1 / e."""
    A, B, C, D, Tst = wl.sw_shaped_system(seed, n=n, n_state=n_state, n_lead=n_lead, k=k)
    L = np.arange(n - n_lead, n)
    M = B + C @ Tst
    Mi = np.linalg.inv(M)
    r = int(np.argmax(np.abs(Mi[L]).max(axis=0)))  # the equation the lead rows load on most

    def smin(e):
        Ms = Mi.copy()
        Ms[:, r] /= e
        return 1.0 / np.sqrt(1.0 + np.linalg.svd(Ms[L], compute_uv=False)[0] ** 2)

    lo, hi = 1e-16, 1.0  # smin is increasing in e
    if smin(hi) < target:
        raise ValueError("target above the unscaled system's sigma_min")
    for _ in range(200):
        mid = np.sqrt(lo * hi)
        if smin(mid) < target:
            lo = mid
        else:
            hi = mid
    e = np.sqrt(lo * hi)
    out = []
    for X in (A, B, C, D):
        X = X.copy()
        X[r] *= e
        out.append(X)
    return (*out, e)


def lead_column_sweep_system(seed, colsum, n=12, n_state=5, n_lead=4, k=3):
    """A regular system in which ONE lead column of C has sum|C_ij| = colsum exactly-ish (gensys.py:587 keeps it iff > tol)."""
    A, B, C, D, _ = wl.sw_shaped_system(seed, n=n, n_state=n_state, n_lead=n_lead, k=k)
    C = C.copy()
    j = n - 1
    C[:, j] *= colsum / np.abs(C[:, j]).sum()
    return A, B, C, D


def lapack_margins(A, B, C, D, tol=1e-8):
    """(smallest singular value of Q2 @ pi, smallest max(|alpha_i|, |beta_i|) over the diagonal pairs) of LAPACK's ordered QZ of
    gensys_setup's pencil (gensys.py:227-235, 243, 267-283): how far the reference's two absolute-tolerance tests are from tipping.
    Both numbers depend on the ORDER in which LAPACK leaves the eigenvalues on the diagonal (swapping two pairs changes their
    magnitudes, not their ratios), so another QZ -- the device's -- may name a system within a factor of a few of the tolerance
    differently; the certificate's guards (spectral_division_model.scale_guards) are order-independent bounds."""
    import scipy.linalg as sla

    from oracle.gensys_qz import gensys_setup

    g0, g1, c, psi, pi = gensys_setup(A, B, C, D, tol)
    _, _, alpha, beta, Qraw, _ = sla.ordqz(g0.astype(complex), g1.astype(complex), sort="ouc", output="complex")
    aa, bb = np.abs(alpha), np.abs(beta)
    stable = ((bb < tol) & (aa >= tol)) | ((bb >= tol) & (aa > bb))
    nu = int(np.sum(~stable))
    Q2 = Qraw.conj().T[len(alpha) - nu:]
    sv = sla.svd(Q2 @ pi, compute_uv=False) if nu > 0 and pi.shape[1] > 0 else np.array([1.0])
    return float(np.min(sv)) if sv.size else 1.0, float(np.maximum(aa, bb).min())
