"""Executable model (plain numpy, scalar loops) of the DEVICE algorithm for gensys.

This is NOT the oracle (the oracle calls LAPACK's zgges/ztgsen through scipy exactly as the
reference does, oracle/gensys_qz.py).  It restates, step for step, what `gensys_kernel` in
geconpy_amd/csrc/dsge_gensys.hpp executes on one wavefront, so that the algorithm can be
validated on the CPU against the oracle and the golden vectors before/independently of the HIP
port, and so that a kernel bug can be told apart from an algorithm bug:

  1. pencil (G0, G1), X = Pi, Ztop = I[:n]             gensys.py:568-614 (index arithmetic only)
  2. G1 -> upper triangular by Givens rotations on rows (left transforms also hit G0 and X)
  3. G0 -> upper Hessenberg keeping G1 triangular (row + column Givens pairs)
  4. complex single-shift QZ with the LAPACK zhgeqz deflation logic (small sub-diagonal of H,
     negligible diagonal of T incl. "chase the zero" procedures, Wilkinson / exceptional shifts)
  5. reorder: stable roots (gensys.py:246 criterion, `realsmall` thresholds) to the top by adjacent
     1x1 swaps (ztgex2 for complex triangular pencils)
  6. gensys post-processing in the partitioned basis (gensys.py:267-343): Jacobi SVDs of Q2 Pi and
     Q1 Pi, existence / uniqueness codes, Phi, and  T = Re(Ztop[:, :ns] A11^-1 [B11, B12 - Phi B22] Ztop^H)
     -- G_0 of gensys.py:322-330 is upper triangular in this basis, so its LU is a back-substitution.
"""
from __future__ import annotations

import numpy as np

SAFMIN = np.finfo(np.float64).tiny
ULP = np.finfo(np.float64).eps  # dlamch('E') * dlamch('B')


def lartg(f, g):
    """c (real), s, r with  [c s; -conj(s) c] [f; g] = [r; 0]."""
    if g == 0:
        return 1.0, 0j, f
    if f == 0:
        ag = abs(g)
        return 0.0, np.conj(g) / ag, ag + 0j
    af, ag = abs(f), abs(g)
    d = np.hypot(af, ag)
    ph = f / af
    return af / d, ph * np.conj(g) / d, ph * d


def rot(x, y, c, s):
    """x' = c x + s y ; y' = c y - conj(s) x   (vectors, in place)."""
    tx = c * x + s * y
    y[:] = c * y - np.conj(s) * x
    x[:] = tx


class Pencil:
    """H (from G0), T (from G1), X = (left transform) Pi, Z = top n rows of the right transform."""

    def __init__(self, G0, G1, n):
        N = G0.shape[0]
        self.N, self.n = N, n
        self.H = G0.astype(np.complex128).copy()
        self.T = G1.astype(np.complex128).copy()
        ell = N - n
        self.X = np.zeros((N, ell), np.complex128)
        self.X[n:, :] = np.eye(ell)
        self.Z = np.eye(N, dtype=np.complex128)[:n].copy()
        self.n_rot = 0
        self.n_refl = 0

    def rot_rows(self, i, k, c, s):
        """rows i (x) and k (y) of H, T, X."""
        for M in (self.H, self.T, self.X):
            rot(M[i], M[k], c, s)
        self.n_rot += 1

    def rot_cols(self, i, k, c, s):
        """columns i (x) and k (y) of H, T, Z."""
        for M in (self.H, self.T, self.Z):
            rot(M[:, i], M[:, k], c, s)
        self.n_rot += 1


def _householder(x):
    """Real reflector H = I - tau v v' with H x = beta e1 (LAPACK dlarfg); v[0] = 1."""
    alpha = x[0]
    xnorm = np.linalg.norm(x[1:])
    if xnorm == 0.0:
        return np.zeros_like(x), 0.0, alpha
    beta = -np.copysign(np.hypot(alpha, xnorm), alpha)
    tau = (beta - alpha) / beta
    v = x / (alpha - beta)
    v[0] = 1.0
    return v, tau, beta


def _apply_left(P, j0, v, tau):
    """rows j0.. of H, T, X  <-  (I - tau v v') rows   (real data at this stage)."""
    if tau == 0.0:
        return
    for M in (P.H, P.T, P.X):
        w = v @ M[j0:, :]
        M[j0:, :] -= tau * np.outer(v, w)
    P.n_refl += 1


def deflate_zero_columns(P, z):
    """The first z columns of T are exactly zero (non-state variables, moved to the front by the
    column permutation).  A QR of H[:, :z] turns the pencil into  [[R0, *], [0, H']] , [[0, *], [0, T']]:
    z roots (alpha = R0_ii, beta = 0) are deflated before any QZ work, exactly, and (H', T') has
    dimension N - z."""
    H = P.H
    for j in range(z):
        v, tau, beta = _householder(H[j:, j].real.copy())
        _apply_left(P, j, v, tau)
        H[j, j] = beta
        H[j + 1 :, j] = 0


def hessenberg_triangular(P, ilo=0):
    N, H, T = P.N, P.H, P.T
    # T[ilo:, ilo:] -> upper triangular by Householder reflectors on rows >= ilo
    for j in range(ilo, N - 1):
        v, tau, beta = _householder(T[j:, j].real.copy())
        _apply_left(P, j, v, tau)
        T[j, j] = beta
        T[j + 1 :, j] = 0
    # H[ilo:, ilo:] -> upper Hessenberg, T stays triangular
    for j in range(ilo, N - 2):
        for i in range(N - 1, j + 1, -1):
            if H[i, j] == 0:
                continue
            c, s, r = lartg(H[i - 1, j], H[i, j])
            P.rot_rows(i - 1, i, c, s)
            H[i - 1, j] = r
            H[i, j] = 0
            if T[i, i - 1] != 0:
                c, s, r = lartg(T[i, i], T[i, i - 1])
                P.rot_cols(i, i - 1, c, s)
                T[i, i] = r
                T[i, i - 1] = 0


def abs1(z):
    return abs(z.real) + abs(z.imag)


def qz_iterate(P, ilo=0, max_it_factor=30):
    """LAPACK zhgeqz (JOB='S') on the active block ilo..N-1 (rows/columns < ilo are already
    triangular).  Returns True on convergence."""
    N, H, T = P.N, P.H, P.T
    if N - ilo <= 1:
        return True
    anorm = np.linalg.norm(H[ilo:, ilo:])
    bnorm = np.linalg.norm(T[ilo:, ilo:])
    atol = max(SAFMIN, ULP * anorm)
    btol = max(SAFMIN, ULP * bnorm)
    ascale = 1.0 / max(SAFMIN, anorm)
    bscale = 1.0 / max(SAFMIN, bnorm)
    ilast = N - 1
    iiter = 0
    eshift = 0j
    maxit = max_it_factor * (N - ilo)
    for _jiter in range(maxit):
        # ---- deflation tests ------------------------------------------------------------
        action = None  # "split60" | "zeroT50" | ("qz", ifirst)
        if ilast == ilo:
            action = "split60"
        elif abs1(H[ilast, ilast - 1]) <= max(SAFMIN, ULP * (abs1(H[ilast, ilast]) + abs1(H[ilast - 1, ilast - 1]))):
            H[ilast, ilast - 1] = 0
            action = "split60"
        elif abs(T[ilast, ilast]) <= btol:
            T[ilast, ilast] = 0
            action = "zeroT50"
        else:
            for j in range(ilast - 1, ilo - 1, -1):
                if j == ilo:
                    ilazro = True
                elif abs1(H[j, j - 1]) <= max(SAFMIN, ULP * (abs1(H[j, j]) + abs1(H[j - 1, j - 1]))):
                    H[j, j - 1] = 0
                    ilazro = True
                else:
                    ilazro = False
                if abs(T[j, j]) < btol:
                    T[j, j] = 0
                    ilazr2 = False
                    if not ilazro:
                        if abs1(H[j, j - 1]) * (ascale * abs1(H[j + 1, j])) <= abs1(H[j, j]) * (ascale * atol):
                            ilazr2 = True
                    if ilazro or ilazr2:
                        # chase the zero of T down; each step splits off / moves the zero
                        for jch in range(j, ilast):
                            c, s, r = lartg(H[jch, jch], H[jch + 1, jch])
                            P.rot_rows(jch, jch + 1, c, s)
                            H[jch, jch] = r
                            H[jch + 1, jch] = 0
                            if ilazr2:
                                H[jch, jch - 1] = H[jch, jch - 1] * c
                            ilazr2 = False
                            if abs1(T[jch + 1, jch + 1]) >= btol:
                                if jch + 1 >= ilast:
                                    action = "split60"
                                else:
                                    action = ("qz", jch + 1)
                                break
                            T[jch + 1, jch + 1] = 0
                        else:
                            action = "zeroT50"
                        break
                    # only the T test passed: chase the zero to T[ilast, ilast]
                    for jch in range(j, ilast):
                        c, s, r = lartg(T[jch, jch + 1], T[jch + 1, jch + 1])
                        P.rot_rows(jch, jch + 1, c, s)
                        T[jch, jch + 1] = r
                        T[jch + 1, jch + 1] = 0
                        c, s, r = lartg(H[jch + 1, jch], H[jch + 1, jch - 1])
                        P.rot_cols(jch, jch - 1, c, s)
                        H[jch + 1, jch] = r
                        H[jch + 1, jch - 1] = 0
                    action = "zeroT50"
                    break
                if ilazro:
                    action = ("qz", j)
                    break
            if action is None:
                return False  # cannot happen (j == ilo always sets ilazro)
        if action == "zeroT50":
            c, s, r = lartg(H[ilast, ilast], H[ilast, ilast - 1])
            P.rot_cols(ilast, ilast - 1, c, s)
            H[ilast, ilast] = r
            H[ilast, ilast - 1] = 0
            action = "split60"
        if action == "split60":
            ilast -= 1
            if ilast < ilo:
                return True
            iiter = 0
            eshift = 0j
            continue
        # ---- one single-shift QZ sweep on ifirst..ilast ------------------------------------
        ifirst = action[1]
        iiter += 1
        if iiter % 10 != 0:
            u12 = (bscale * T[ilast - 1, ilast]) / (bscale * T[ilast, ilast])
            ad11 = (ascale * H[ilast - 1, ilast - 1]) / (bscale * T[ilast - 1, ilast - 1])
            ad21 = (ascale * H[ilast, ilast - 1]) / (bscale * T[ilast - 1, ilast - 1])
            ad12 = (ascale * H[ilast - 1, ilast]) / (bscale * T[ilast, ilast])
            ad22 = (ascale * H[ilast, ilast]) / (bscale * T[ilast, ilast])
            abi22 = ad22 - u12 * ad21
            t1 = 0.5 * (ad11 + abi22)
            rtdisc = np.sqrt(t1 * t1 + ad12 * ad21 - ad11 * ad22 + 0j)
            temp = (t1 - abi22).real * rtdisc.real + (t1 - abi22).imag * rtdisc.imag
            shift = t1 + rtdisc if temp <= 0 else t1 - rtdisc
        else:
            eshift = eshift + (ascale * H[ilast, ilast - 1]) / (bscale * T[ilast - 1, ilast - 1])
            shift = eshift
        istart = ifirst
        ctemp = ascale * H[ifirst, ifirst] - shift * (bscale * T[ifirst, ifirst])
        for j in range(ilast - 1, ifirst, -1):
            ct = ascale * H[j, j] - shift * (bscale * T[j, j])
            temp = abs1(ct)
            temp2 = ascale * abs1(H[j + 1, j])
            tempr = max(temp, temp2)
            if tempr < 1.0 and tempr != 0.0:
                temp /= tempr
                temp2 /= tempr
            if abs1(H[j, j - 1]) * temp2 <= temp * atol:
                istart = j
                ctemp = ct
                break
        ctemp2 = ascale * H[istart + 1, istart]
        c, s, _ = lartg(ctemp, ctemp2)
        for j in range(istart, ilast):
            if j > istart:
                c, s, r = lartg(H[j, j - 1], H[j + 1, j - 1])
                P.rot_rows(j, j + 1, c, s)
                H[j, j - 1] = r
                H[j + 1, j - 1] = 0
            else:
                P.rot_rows(j, j + 1, c, s)
            c, s, r = lartg(T[j + 1, j + 1], T[j + 1, j])
            P.rot_cols(j + 1, j, c, s)
            T[j + 1, j + 1] = r
            T[j + 1, j] = 0
    return False


def _house3(x):
    """Reflector I - tau v v' (v[0] = 1) with (I - tau v v') x = beta e1, any length; tau = 0 when x[1:] = 0."""
    alpha = x[0]
    xn = np.sqrt(np.sum(x[1:] ** 2))
    if xn == 0.0:
        return np.zeros_like(x), 0.0
    beta = -np.copysign(np.hypot(alpha, xn), alpha)
    v = x / (alpha - beta)
    v[0] = 1.0
    return v, (beta - alpha) / beta


def real_double_shift_stage(P, ilo=0, max_it=30):
    """ACCELERATOR in front of qz_iterate (round 3): implicit double-shift QZ sweeps in REAL arithmetic (Moler & Stewart 1973;
    Golub & Van Loan, Algorithm 7.7.2) on the real Hessenberg-triangular pencil the reduction hands over, until every
    sub-diagonal block has shrunk to 1 x 1 or 2 x 2 -- or until anything unusual turns up (a negligible diagonal entry of T,
    i.e. an infinite root inside the active block; 30 sweeps without a deflation): the stage then simply stops.  Every
    transformation is an orthogonal equivalence that keeps the Hessenberg-triangular form, so whatever state it leaves is a
    valid input for the complex single-shift iteration (qz_iterate), which owns all the deflation logic of zhgeqz and only has
    to split the remaining 2 x 2 blocks.  A real sweep step costs two 3-wide reflectors in real arithmetic (rows, then columns) and advances two
    shifts; a complex single-shift step costs two complex rotations and advances one.
    Returns (number of sweep steps, number of sweeps)."""
    N = P.N
    H, T, X, Z = P.H, P.T, P.X, P.Z  # complex storage, real content at this stage
    if N - ilo <= 2:
        return 0, 0
    bnorm = np.linalg.norm(T[ilo:, ilo:].real)
    btol = max(SAFMIN, ULP * bnorm)

    def small(j):  # negligible H[j, j-1] (zhgeqz's test)
        return abs(H[j, j - 1].real) <= max(SAFMIN, ULP * (abs(H[j, j].real) + abs(H[j - 1, j - 1].real)))

    def left(k, nrow, x, c0):  # reflector from x on rows k..k+nrow-1, columns c0.. of H, from k of T, all of X
        v, tau = _house3(np.array(x, dtype=float))
        if tau == 0.0:
            return
        for M, cs in ((H, c0), (T, k), (X, 0)):
            blk = M[k:k + nrow, cs:].real
            M[k:k + nrow, cs:] = blk - tau * np.outer(v, v @ blk)

    def right(k, ncol, rowvec, rmaxH, rmaxT):  # [rowvec] Zr = [0 .. 0 *] on columns k..k+ncol-1
        v, tau = _house3(np.array(rowvec[::-1], dtype=float))
        if tau == 0.0:
            return
        v = v[::-1]
        for M, r1 in ((H, rmaxH), (T, rmaxT), (Z, Z.shape[0])):
            blk = M[:r1, k:k + ncol].real
            M[:r1, k:k + ncol] = blk - tau * np.outer(blk @ v, v)

    def right_first(k, w, rmaxH, rmaxT):  # (I - tau v v') e1 = w / beta on columns k..k+2: column k of [.] Zr is [.] w / beta
        v, tau = _house3(np.array(w, dtype=float))
        if tau == 0.0:
            return
        for M, r1 in ((H, rmaxH), (T, rmaxT), (Z, Z.shape[0])):
            blk = M[:r1, k:k + 3].real
            M[:r1, k:k + 3] = blk - tau * np.outer(blk @ v, v)

    ilast = N - 1
    it = 0
    steps = sweeps = 0
    while ilast - ilo >= 2:
        if small(ilast):
            H[ilast, ilast - 1] = 0
            ilast -= 1
            it = 0
            continue
        if small(ilast - 1):
            H[ilast - 1, ilast - 2] = 0
            ilast -= 2  # a 2 x 2 block: the complex iteration splits it
            it = 0
            continue
        ifirst = ilo
        for j in range(ilast - 2, ilo, -1):
            if small(j):
                H[j, j - 1] = 0
                ifirst = j
                break
        if np.any(np.abs(np.diag(T)[ifirst:ilast + 1].real) <= btol):
            break
        it += 1
        if it > max_it:
            break
        m = ilast
        p_, q_, r_, s_ = H[m - 1, m - 1].real, H[m - 1, m].real, H[m, m - 1].real, H[m, m].real
        e_, f_, g_ = T[m - 1, m - 1].real, T[m - 1, m].real, T[m, m].real
        if it % 10 == 0:  # exceptional shifts
            w_ = 1.5 * (abs(r_ / e_) + abs(H[m - 1, m - 2].real / T[m - 2, m - 2].real))
            tr, det = w_, w_ * w_
        else:
            tr = p_ / e_ + (s_ - r_ * f_ / e_) / g_
            det = (p_ * s_ - q_ * r_) / (e_ * g_)
        k = ifirst
        a11, a12, a21, a22, a32 = H[k, k].real, H[k, k + 1].real, H[k + 1, k].real, H[k + 1, k + 1].real, H[k + 2, k + 1].real
        b11, b12, b22 = T[k, k].real, T[k, k + 1].real, T[k + 1, k + 1].real
        m11, m21 = a11 / b11, a21 / b11
        y2 = m21 / b22
        y1 = (m11 - b12 * y2) / b11
        x = a11 * y1 + a12 * y2 - tr * m11 + det
        y = a21 * y1 + a22 * y2 - tr * m21
        z = a32 * y2
        sweeps += 1
        for k in range(ifirst, ilast - 1):
            left(k, 3, [x, y, z], max(k - 1, ifirst) if k > ifirst else k)
            if k > ifirst:
                H[k + 1, k - 1] = 0
                H[k + 2, k - 1] = 0
            # ONE right reflector per step (LAPACK dhgeqz): its first column is the null vector of rows k+1, k+2 of the
            # 3 x 3 block of T -- their cross product --, which clears column k below the diagonal; T[k+2, k+1] stays for
            # the next step's block
            right_first(k, np.cross(T[k + 1, k:k + 3].real, T[k + 2, k:k + 3].real), min(k + 4, ilast + 1), k + 3)
            T[k + 1, k] = 0
            T[k + 2, k] = 0
            x, y = H[k + 1, k].real, H[k + 2, k].real
            if k < ilast - 2:
                z = H[k + 3, k].real
            steps += 1
        k = ilast - 1
        left(k, 2, [x, y], k - 1)
        H[k + 1, k - 1] = 0
        right(k, 2, [T[k + 1, k].real, T[k + 1, k + 1].real], ilast + 1, k + 2)
        T[k + 1, k] = 0
        steps += 1
    return steps, sweeps


def swap_adjacent(P, k):
    """Exchange the 1x1 blocks k and k+1 of the triangular pencil (LAPACK ztgex2)."""
    H, T = P.H, P.T
    f = H[k + 1, k + 1] * T[k, k] - T[k + 1, k + 1] * H[k, k]
    g = H[k + 1, k + 1] * T[k, k + 1] - T[k + 1, k + 1] * H[k, k + 1]
    sa = abs(H[k + 1, k + 1])
    sb = abs(T[k + 1, k + 1])
    c0, s0, _ = lartg(g, f)
    P.rot_cols(k, k + 1, c0, -np.conj(s0))
    if sa >= sb:
        c, s, _ = lartg(H[k, k], H[k + 1, k])
    else:
        c, s, _ = lartg(T[k, k], T[k + 1, k])
    P.rot_rows(k, k + 1, c, s)
    H[k + 1, k] = 0
    T[k + 1, k] = 0


def is_stable(a, b, rs):
    """gensys.py:246 on (alpha, beta) = (H_ii, T_ii)."""
    aa, ab = abs(a), abs(b)
    return (ab < rs and aa >= rs) or (ab >= rs and aa > ab)


def reorder_stable_first(P, rs):
    N, H, T = P.N, P.H, P.T
    ns = 0
    for i in range(N):
        if is_stable(H[i, i], T[i, i], rs):
            for k in range(i - 1, ns - 1, -1):
                swap_adjacent(P, k)
            ns += 1
    return ns


def jacobi_svd(M, max_sweeps=60):
    """One-sided (Hestenes) Jacobi on the columns of M (r x c): returns G = M V (columns sigma_j u_j),
    V (c x c) and sigma (c,).  No sorting."""
    G = M.astype(np.complex128).copy()
    c = G.shape[1]
    V = np.eye(c, dtype=np.complex128)
    for _ in range(max_sweeps):
        rotated = False
        for p in range(c - 1):
            for q in range(p + 1, c):
                alpha = np.vdot(G[:, p], G[:, p]).real
                beta = np.vdot(G[:, q], G[:, q]).real
                gamma = np.vdot(G[:, p], G[:, q])
                ag = abs(gamma)
                if ag < 1e-290 or ag <= 1e-15 * np.sqrt(alpha) * np.sqrt(beta):
                    continue
                rotated = True
                ph = gamma / ag
                zeta = (beta - alpha) / (2.0 * ag)
                t = (1.0 if zeta >= 0 else -1.0) / (abs(zeta) + np.sqrt(1.0 + zeta * zeta))
                cs = 1.0 / np.sqrt(1.0 + t * t)
                sn = cs * t
                for Mx in (G, V):
                    gp = Mx[:, p].copy()
                    gq = Mx[:, q] * np.conj(ph)
                    Mx[:, p] = cs * gp - sn * gq
                    Mx[:, q] = sn * gp + cs * gq
        if not rotated:
            break
    sigma = np.sqrt(np.maximum(0.0, np.einsum("ij,ij->j", G.conj(), G).real))
    return G, V, sigma


def gensys_post(P, ns, rs):
    """gensys.py:267-343 in the partitioned Schur basis -> (T (n x n real), eu)."""
    N, n, H, T, X, Z = P.N, P.n, P.H, P.T, P.X, P.Z
    nu = N - ns
    ell = N - n
    eu = np.zeros(3, dtype=np.int64)
    # coincident zeros (gensys.py:243-244)
    for i in range(N):
        if abs(H[i, i]) < rs and abs(T[i, i]) < rs:
            eu[0] = eu[1] = -2
            return np.zeros((n, n)), eu
    eta2 = X[ns:, :]
    eta1 = X[:ns, :]
    # SVD of eta2 (nu x ell): keep sigma > rs
    if nu > 0:
        G2, V2, s2 = jacobi_svd(eta2)
        keep2 = s2 > rs
    else:
        G2 = np.zeros((0, ell), np.complex128)
        V2 = np.eye(ell, dtype=np.complex128)
        s2 = np.zeros(ell)
        keep2 = np.zeros(ell, bool)
    r2 = int(keep2.sum())
    if r2 >= nu:
        eu[0] = 1
    # eta = [eta1; eta2] = Q Pi has orthonormal columns, so eta1^H eta1 + eta2^H eta2 = I (CS
    # decomposition): the right singular vectors of eta1 are those of eta2 and G1 = eta1 V2 has
    # orthogonal columns with norms sqrt(1 - s2^2).  The reference runs a second gesdd on eta1
    # (gensys.py:291-296); here its singular values are the column norms of eta1 V2 and V1 := V2.
    G1 = eta1 @ V2
    s1 = np.sqrt(np.einsum("ij,ij->j", G1.conj(), G1).real) if ns > 0 else np.zeros(ell)
    keep1 = s1 > rs
    r1 = int(keep1.sum())
    # uniqueness (gensys.py:301-310): V1k - V2k V2k^H V1k keeps exactly the columns v_j with keep1_j and
    # not keep2_j, which are orthonormal, so its rank is their count (each has norm 1 > rs N).
    if r1 == 0:
        unique = True
    else:
        n_loose = int((keep1 & ~keep2).sum())
        eu[2] = n_loose
        unique = n_loose == 0
    if unique:
        eu[1] = 1
    # Phi = eta1_k pinv_k(eta2) = sum_j [keep1_j keep2_j / s2_j^2] (eta1 v_j)(G2_j)^H
    w = np.where(keep1 & keep2, 1.0 / np.maximum(s2, 1e-300) ** 2, 0.0)
    Phi = (G1 * w[None, :]) @ G2.conj().T  # ns x nu
    A11 = H[:ns, :ns]
    rhs = np.hstack((T[:ns, :ns], T[:ns, ns:] - Phi @ T[ns:, ns:]))  # ns x N
    # back-substitution with the upper-triangular A11
    Y = np.zeros((ns, N), np.complex128)
    for i in range(ns - 1, -1, -1):
        Y[i] = (rhs[i] - A11[i, i + 1 :] @ Y[i + 1 :]) / A11[i, i]
    Tm = (Z[:, :ns] @ Y @ Z.conj().T).real
    return Tm, eu


def gensys_device_model(A, B, C, D, tol=1e-8, deflate=True, real_stage=True):
    """(T, eu, info) for one system; mirrors the kernel's control flow."""
    A, B, C = (np.asarray(x, dtype=np.float64) for x in (A, B, C))
    n = A.shape[0]
    lead = np.flatnonzero(np.abs(C).sum(axis=0) > tol)
    ell = lead.size
    N = n + ell
    G0 = np.zeros((N, N))
    G0[:n, :n] = -B
    G0[:n, n:] = -C[:, lead]
    G0[n + np.arange(ell), lead] = 1.0
    G1 = np.zeros((N, N))
    G1[:n, :n] = A
    G1[n:, n:] = np.eye(ell)
    rs = tol if tol > 0 else np.spacing(1.0)
    # structural deflation: columns of G1 that are exactly zero (non-state variables) go first
    zero_cols = [j for j in range(n) if not np.any(A[:, j] != 0.0)]
    z = len(zero_cols) if deflate else 0
    if z:
        colperm = np.array(zero_cols + [j for j in range(N) if j not in set(zero_cols)])
        G0 = G0[:, colperm]
        G1 = G1[:, colperm]
    P = Pencil(G0, G1, n)
    if z:
        P.Z = np.eye(N, dtype=np.complex128)[:n][:, colperm].copy()
        deflate_zero_columns(P, z)
    hessenberg_triangular(P, ilo=z)
    rot_ht = P.n_rot
    real_steps = real_double_shift_stage(P, ilo=z) if real_stage else (0, 0)
    ok = qz_iterate(P, ilo=z)
    rot_qz = P.n_rot - rot_ht
    if not ok:
        return np.zeros((n, n)), np.array([-3, -3, 0]), dict(converged=False)
    ns = reorder_stable_first(P, rs)
    Tm, eu = gensys_post(P, ns, rs)
    info = dict(converged=True, N=N, ns=ns, z=z, real_steps=real_steps, n_refl=P.n_refl, rot_ht=rot_ht, rot_qz=rot_qz, rot_reorder=P.n_rot - rot_ht - rot_qz,
                alpha=np.diag(P.H).copy(), beta=np.diag(P.T).copy())
    return Tm, eu, info
