"""Oracle for SURVEY.md section 8 (f4): second-order perturbation + pruned state space (TEST INFRASTRUCTURE ONLY).

*** parity unpinned against the reference: gEconpy has no second-order solver (it raises at
gEconpy/model/perturbation.py:97-98 and gEconpy/model/model.py:1433-1434, 1614-1615); BASELINE.json configs[4] asks for
one anyway. ***  This module restates the PUBLISHED algorithm (Schmitt-Grohe & Uribe 2004, "Solving dynamic general
equilibrium models using a second-order approximation to the policy function", in the unfolded-tensor form of Dynare's
k-order solver) in gEconpy's notation and pins it on a model whose exact policy function is known in closed form
(Brock-Mirman: log utility, full depreciation; tests/test_oracle_second_order.py), so that a device kernel has something
to be checked against.  No kernel is built on it yet.

Notation (gEconpy/model/perturbation.py:42-46, gEconpy/solvers/shared.py:22-26): the model is
F(y_{t-1}, y_t, y_{t+1}, u_t) = 0 with Jacobians A, B, C, D at the steady state and first-order solution
y_t = T y_{t-1} + R u_t (deviations).  Second order:

    y_t = T y- + R u + 1/2 [ g_yy (y- (x) y-) + 2 g_yu (y- (x) u) + g_uu (u (x) u) + g_ss ]

With z = [y-; y; y+; u] (m = 3n + k), H = the Hessian of F unfolded as n x m^2 (H[i, a m + b] = d2 F_i / dz_a dz_b) and
M = B + C T:

    Zy = [I; T; T T; 0]      Zu = [0; R; T R; I]      Zu' = [0; 0; R; 0]                      (dz / d y-, d u_t, d u_{t+1})
    M g_yy + C g_yy (T (x) T) = -H (Zy (x) Zy)                                                 (generalised Sylvester)
    M g_yu = -[ H (Zy (x) Zu) + C g_yy (T (x) R) ]
    M g_uu = -[ H (Zu (x) Zu) + C g_yy (R (x) R) ]
    (M + C) g_ss = -[ C g_uu + H (Zu' (x) Zu') ] vec(Sigma)
"""
from __future__ import annotations

import numpy as np


def _commutation(p, q):
    """K_{p,q}: K vec(X) = vec(X') for X (p x q), column-major vec; (a (x) b) = K (b (x) a) for a in R^q, b in R^p."""
    K = np.zeros((p * q, p * q))
    for i in range(p):
        for j in range(q):
            K[i * q + j, j * p + i] = 1.0
    return K


def second_order_solution(A, B, C, D, H, T, R, Sigma):
    """-> dict(g_yy (n, n^2), g_yu (n, n k), g_uu (n, k^2), g_ss (n,)); Kronecker products are row-major
    ((a (x) b)[i nb + j] = a_i b_j), matching ``np.kron``."""
    n, k = D.shape
    m = 3 * n + k
    H = np.asarray(H, dtype=np.float64).reshape(n, m * m)
    M = B + C @ T
    Zy = np.vstack([np.eye(n), T, T @ T, np.zeros((k, n))])
    Zu = np.vstack([np.zeros((n, k)), R, T @ R, np.eye(k)])
    Zup = np.vstack([np.zeros((2 * n, k)), R, np.zeros((k, k))])
    # generalised Sylvester  M X + C X (T (x) T) = rhs:  vec form  (I (x) M + (T (x) T)' (x) C) vec(X) = vec(rhs)
    rhs = -H @ np.kron(Zy, Zy)
    TT = np.kron(T, T)
    big = np.kron(np.eye(n * n), M) + np.kron(TT.T, C)
    g_yy = np.linalg.solve(big, rhs.reshape(-1, order="F")).reshape(n, n * n, order="F")
    g_yu = -np.linalg.solve(M, H @ np.kron(Zy, Zu) + C @ g_yy @ np.kron(T, R))
    g_uu = -np.linalg.solve(M, H @ np.kron(Zu, Zu) + C @ g_yy @ np.kron(R, R))
    vS = np.asarray(Sigma, dtype=np.float64).reshape(-1)
    g_ss = -np.linalg.solve(M + C, (C @ g_uu + H @ np.kron(Zup, Zup)) @ vS)
    return dict(g_yy=g_yy, g_yu=g_yu, g_uu=g_uu, g_ss=g_ss)


def second_order_residual(A, B, C, D, H, T, R, Sigma, sol):
    """Max abs residual of the four second-order conditions (a self-check that does not use the solver's own linear
    algebra: it re-inserts the solution into the differentiated model)."""
    n, k = D.shape
    m = 3 * n + k
    H = np.asarray(H, dtype=np.float64).reshape(n, m * m)
    Zy = np.vstack([np.eye(n), T, T @ T, np.zeros((k, n))])
    Zu = np.vstack([np.zeros((n, k)), R, T @ R, np.eye(k)])
    Zup = np.vstack([np.zeros((2 * n, k)), R, np.zeros((k, k))])
    g_yy, g_yu, g_uu, g_ss = sol["g_yy"], sol["g_yu"], sol["g_uu"], sol["g_ss"]
    # d2/dy-dy-:  B g_yy + C (g_yy (T (x) T) + T g_yy) + H (Zy (x) Zy)
    r1 = B @ g_yy + C @ (g_yy @ np.kron(T, T) + T @ g_yy) + H @ np.kron(Zy, Zy)
    r2 = B @ g_yu + C @ (g_yy @ np.kron(T, R) + T @ g_yu) + H @ np.kron(Zy, Zu)
    r3 = B @ g_uu + C @ (g_yy @ np.kron(R, R) + T @ g_uu) + H @ np.kron(Zu, Zu)
    vS = np.asarray(Sigma).reshape(-1)
    r4 = B @ g_ss + C @ (g_ss + T @ g_ss + g_uu @ vS) + H @ np.kron(Zup, Zup) @ vS
    return max(np.abs(r1).max(), np.abs(r2).max(), np.abs(r3).max(), np.abs(r4).max())


def pruned_state_space(T, R, sol, Sigma):
    """Pruned second-order system (Kim, Kim, Schaumburg & Sims 2008; Andreasen, Fernandez-Villaverde & Rubio-Ramirez 2018)
    as ONE linear recursion in the augmented state  z = [x_f; x_s; x_f (x) x_f]  (n + n + n^2):

        x_f' = T x_f + R u
        x_s' = T x_s + 1/2 g_yy (x_f (x) x_f) + g_yu (x_f (x) u) + 1/2 g_uu (u (x) u) + 1/2 g_ss
        (x_f (x) x_f)' = (T (x) T)(x_f (x) x_f) + (T (x) R)(x_f (x) u) + (R (x) T)(u (x) x_f) + (R (x) R)(u (x) u)

    i.e.  z' = c + Az z + xi'  with a martingale-difference xi whose unconditional covariance Qz follows from Gaussian
    fourth moments and P_f = dlyap(T, R Sigma R').  The observed variables are  y = x_f + x_s  (deviations): Z_aug =
    [Z, Z, 0].  -> dict(Az, c, Qz, P_f, mean) with ``mean`` the unconditional mean of z.  A Gaussian filter on this
    system is the quasi-likelihood configs[4] asks for; ``oracle.kalman_filter_logp`` takes Az, c, Qz as T, c, R Q R'."""
    import scipy.linalg as sla

    n, k = R.shape
    Sigma = np.asarray(Sigma, dtype=np.float64)
    g_yy, g_yu, g_uu, g_ss = sol["g_yy"], sol["g_yu"], sol["g_uu"], sol["g_ss"]
    vS = Sigma.reshape(-1)
    N = 2 * n + n * n
    Az = np.zeros((N, N))
    Az[:n, :n] = T
    Az[n:2 * n, n:2 * n] = T
    Az[n:2 * n, 2 * n:] = 0.5 * g_yy
    Az[2 * n:, 2 * n:] = np.kron(T, T)
    c = np.zeros(N)
    c[n:2 * n] = 0.5 * (g_uu @ vS + g_ss)
    c[2 * n:] = np.kron(R, R) @ vS
    P_f = sla.solve_discrete_lyapunov(T, R @ Sigma @ R.T)
    # xi = L1 u + L2 (x_f (x) u) + L3 (u (x) u - vec Sigma), with (u (x) x_f) = K (x_f (x) u)
    Knk = _commutation(n, k)  # (u (x) x) = Knk... maps (x (x) u) -> (u (x) x)
    L1 = np.vstack([R, np.zeros((n, k)), np.zeros((n * n, k))])
    L2 = np.vstack([np.zeros((n, n * k)), g_yu, np.kron(T, R) + np.kron(R, T) @ _xu_to_ux(n, k)])
    L3 = np.vstack([np.zeros((n, k * k)), 0.5 * g_uu, np.kron(R, R)])
    Kkk = _commutation(k, k)
    V2 = np.kron(P_f, Sigma)                                  # E (x (x) u)(x (x) u)'   (x, u independent, zero mean)
    V3 = (np.eye(k * k) + Kkk) @ np.kron(Sigma, Sigma)        # Var(u (x) u), Gaussian u
    Qz = L1 @ Sigma @ L1.T + L2 @ V2 @ L2.T + L3 @ V3 @ L3.T  # odd cross moments vanish
    del Knk
    mean = np.linalg.solve(np.eye(N) - Az, c)
    return dict(Az=Az, c=c, Qz=0.5 * (Qz + Qz.T), P_f=P_f, mean=mean)


def _xu_to_ux(n, k):
    """P with  (u (x) x) = P (x (x) u)  for x in R^n, u in R^k (row-major Kronecker)."""
    P = np.zeros((k * n, n * k))
    for i in range(n):
        for j in range(k):
            P[j * n + i, i * k + j] = 1.0
    return P


def simulate_pruned(T, R, sol, shocks):
    """Direct simulation of the pruned system from x_f = x_s = 0 -> (x_f, x_s) paths (len(shocks) + 1 rows)."""
    n, k = R.shape
    xf = np.zeros((len(shocks) + 1, n))
    xs = np.zeros_like(xf)
    for t, u in enumerate(shocks):
        f = xf[t]
        xf[t + 1] = T @ f + R @ u
        xs[t + 1] = (T @ xs[t] + 0.5 * sol["g_yy"] @ np.kron(f, f) + sol["g_yu"] @ np.kron(f, u) +
                     0.5 * sol["g_uu"] @ np.kron(u, u) + 0.5 * sol["g_ss"])
    return xf, xs


# ======================================================================================================================
# Reduced (minimal-state) formulation -- what the device kernels of geconpy_amd/csrc/dsge_second_order.hpp compute.
#
# The policy function depends on y_{t-1} only through the state variables S (the non-zero columns of A, hence of T), so
# g_yy, g_yu vanish outside S x S / S x u, and the pruned filter only has to carry
#     z = [ x_f[U] ; x_s[U] ; w ],   U = S u O (states first, then the observed non-states),   w_(a<=b) = x_f[S_a] x_f[S_b]
# of dimension 2|U| + s(s+1)/2 (207 for the SW-shaped workload) instead of 2n + n^2 (1680).  With jitter = 0 the reduced and
# the full filter agree to rounding (tests/test_oracle_second_order.py); with the upstream jitter on F and P+ the two are
# different regularisations (the full space carries duplicate and non-state products), and the REDUCED one is the
# definition the device is checked against.  The Hessian arrives as the device takes it: COO entries (equation, z_a, z_b),
# z_a <= z_b, z = [y-; y; y+; u], pattern shared by all draws, one value vector per draw.
# ======================================================================================================================


def hessian_coo_to_dense(n, k, idx, val):
    """(nnz, 3) upper-triangular COO -> dense n x m^2 unfolded Hessian (symmetric in its two z indices)."""
    m = 3 * n + k
    H = np.zeros((n, m, m))
    for (i, a, b), v in zip(np.asarray(idx), np.asarray(val)):
        assert a <= b
        H[i, a, b] = v
        H[i, b, a] = v
    return H.reshape(n, m * m)


def _hess_contract(n, idx, val, ZL, ZR):
    """sum_e val_e (ZL[a_e] (x) ZR[b_e] + [a_e != b_e] ZL[b_e] (x) ZR[a_e]) scattered to row i_e: H (ZL (x) ZR) as (n, cl, cr)."""
    out = np.zeros((n, ZL.shape[1], ZR.shape[1]))
    for (i, a, b), v in zip(np.asarray(idx), np.asarray(val)):
        out[i] += v * np.outer(ZL[a], ZR[b])
        if a != b:
            out[i] += v * np.outer(ZL[b], ZR[a])
    return out


def _sylvester_schur(G, Ts, X0):
    """X_i + sum_j G_ij Ts' X_j Ts = X0_i for X (n, s, s), by a complex Schur form of Ts and back-substitution over the
    (c, d) index pairs (a Bartels-Stewart variant; independent of the doubling iteration the device uses)."""
    import scipy.linalg as sla

    n, s = X0.shape[0], Ts.shape[0]
    U, Q = sla.schur(Ts.astype(np.complex128), output="complex")  # Ts = Q U Q^H
    Y0 = np.einsum("ac,iab,bd->icd", Q, X0.astype(np.complex128), Q)  # Q' X0_i Q
    Y = np.zeros_like(Y0)
    eye = np.eye(n)
    for c in range(s):
        for d in range(s):
            r = np.einsum("a,iab,b->i", U[: c + 1, c], Y[:, : c + 1, : d + 1], U[: d + 1, d])  # (Y[:, c, d] is still 0)
            Y[:, c, d] = np.linalg.solve(eye + U[c, c] * U[d, d] * G, Y0[:, c, d] - G @ r)
    X = np.einsum("ac,icd,bd->iab", Q.conj(), Y, Q.conj())  # conj(Q) Y Q^H
    assert np.abs(X.imag).max() <= 1e-9 * max(1.0, np.abs(X.real).max())
    return X.real


def second_order_solution_reduced(B, C, T, R, hess_idx, hess_val, Sigma, S=None):
    """-> dict(g_yy (n, s, s), g_yu (n, s, k), g_uu (n, k, k), g_ss (n,), S): the four second-order blocks on the state
    columns.  Same equations as ``second_order_solution`` (header of this file) restricted to S."""
    n, k = R.shape
    S = np.flatnonzero((T != 0).any(axis=0)) if S is None else np.asarray(S)
    Ts, Rs = T[np.ix_(S, S)], R[S]
    M = B + C @ T
    Zy = np.vstack([np.eye(n), T, T @ T, np.zeros((k, n))])[:, S]
    Zu = np.vstack([np.zeros((n, k)), R, T @ R, np.eye(k)])
    Zup = np.vstack([np.zeros((2 * n, k)), R, np.zeros((k, k))])
    G = np.linalg.solve(M, C)
    X0 = np.linalg.solve(M, -_hess_contract(n, hess_idx, hess_val, Zy, Zy).reshape(n, -1)).reshape(n, len(S), len(S))
    g_yy = _sylvester_schur(G, Ts, X0)
    rhs_yu = _hess_contract(n, hess_idx, hess_val, Zy, Zu) + np.einsum("ij,jab,ac,bq->icq", C, g_yy, Ts, Rs)
    g_yu = -np.linalg.solve(M, rhs_yu.reshape(n, -1)).reshape(n, len(S), k)
    rhs_uu = _hess_contract(n, hess_idx, hess_val, Zu, Zu) + np.einsum("ij,jab,ap,bq->ipq", C, g_yy, Rs, Rs)
    g_uu = -np.linalg.solve(M, rhs_uu.reshape(n, -1)).reshape(n, k, k)
    Sigma = np.asarray(Sigma, dtype=np.float64)
    hpp = _hess_contract(n, hess_idx, hess_val, Zup, Zup)
    g_ss = -np.linalg.solve(M + C, np.einsum("ij,jpq,pq->i", C, g_uu, Sigma) + np.einsum("ipq,pq->i", hpp, Sigma))
    return dict(g_yy=g_yy, g_yu=g_yu, g_uu=g_uu, g_ss=g_ss, S=S)


def retained_variables(S, obs):
    """U = S followed by the observed variables that are not states (ascending)."""
    S = list(np.asarray(S))
    return np.array(S + sorted(set(int(o) for o in obs) - set(S)), dtype=np.int64)


def half_index(s):
    """The (a, b), a <= b, pairs of the symmetric half in the order the device uses (row-major upper triangle)."""
    return [(a, b) for a in range(s) for b in range(a, s)]


def pruned_state_space_reduced(T, R, sol, Sigma, obs):
    """Minimal pruned system z' = c + Az z + xi on z = [x_f[U]; x_s[U]; w] -> dict(Az, c, Qz, mean, P_f, U, S, m)."""
    n, k = R.shape
    S = np.asarray(sol["S"])
    s = len(S)
    U = retained_variables(S, obs)
    u = len(U)
    pairs = half_index(s)
    q = len(pairs)
    m = 2 * u + q
    Sigma = np.asarray(Sigma, dtype=np.float64)
    Ts, Rs, Tu, Ru = T[np.ix_(S, S)], R[S], T[np.ix_(U, S)], R[U]
    gyy, gyu, guu, gss = sol["g_yy"][U], sol["g_yu"][U], sol["g_uu"][U], sol["g_ss"][U]
    Az = np.zeros((m, m))
    Az[:u, :s] = Tu
    Az[u:2 * u, u:u + s] = Tu
    K2 = np.zeros((q, q))
    Gh = np.zeros((u, q))
    for j, (c_, d_) in enumerate(pairs):
        Gh[:, j] = 0.5 * gyy[:, c_, c_] if c_ == d_ else 0.5 * (gyy[:, c_, d_] + gyy[:, d_, c_])
        for i, (a_, b_) in enumerate(pairs):
            K2[i, j] = Ts[a_, c_] * Ts[b_, d_] + (Ts[a_, d_] * Ts[b_, c_] if c_ != d_ else 0.0)
    Az[u:2 * u, 2 * u:] = Gh
    Az[2 * u:, 2 * u:] = K2
    RSR = Rs @ Sigma @ Rs.T
    c = np.zeros(m)
    c[u:2 * u] = 0.5 * (np.einsum("ipq,pq->i", guu, Sigma) + gss)
    c[2 * u:] = [RSR[a_, b_] for a_, b_ in pairs]
    import scipy.linalg as sla

    P_f = sla.solve_discrete_lyapunov(Ts, RSR)
    # xi = L1 e + L2 (x_f[S] (x) e) + L3 (e (x) e - vec Sigma)
    L1 = np.zeros((m, k))
    L1[:u] = Ru
    L2 = np.zeros((m, s, k))
    L2[u:2 * u] = gyu
    L3 = np.zeros((m, k, k))
    L3[u:2 * u] = 0.5 * guu
    for i, (a_, b_) in enumerate(pairs):
        L2[2 * u + i] = np.outer(Ts[a_], Rs[b_]) + np.outer(Ts[b_], Rs[a_])
        L3[2 * u + i] = np.outer(Rs[a_], Rs[b_])
    L2 = L2.reshape(m, s * k)
    L3 = L3.reshape(m, k * k)
    Kkk = _commutation(k, k)
    V2 = np.kron(P_f, Sigma)
    V3 = (np.eye(k * k) + Kkk) @ np.kron(Sigma, Sigma)
    Qz = L1 @ Sigma @ L1.T + L2 @ V2 @ L2.T + L3 @ V3 @ L3.T
    mean = np.linalg.solve(np.eye(m) - Az, c)
    return dict(Az=Az, c=c, Qz=0.5 * (Qz + Qz.T), mean=mean, P_f=P_f, U=U, S=S, m=m)


def pruned_design(Z, d, U, m):
    """Z_aug = [Z[:, U], Z[:, U], 0] (y = Z (x_f + x_s) + d)."""
    u = len(U)
    Za = np.zeros((Z.shape[0], m))
    Za[:, :u] = Z[:, U]
    Za[:, u:2 * u] = Z[:, U]
    return Za


def pruned_kalman_logp(T, R, sol, Sigma, Z, y, H=None, d=None, jitter=None, return_parts=False, conventions=None):
    """Gaussian quasi-likelihood of the pruned second-order system: the "standard" filter (oracle.statespace) on the reduced
    augmented state, started from its stationary mean and covariance."""
    import scipy.linalg as sla

    from .statespace import JITTER_DEFAULT, kalman_filter_logp

    obs = np.flatnonzero((np.asarray(Z) != 0).any(axis=0))
    ps = pruned_state_space_reduced(T, R, sol, Sigma, obs)
    Za = pruned_design(np.asarray(Z, dtype=np.float64), d, ps["U"], ps["m"])
    P0 = sla.solve_discrete_lyapunov(ps["Az"], ps["Qz"])
    lp = kalman_filter_logp(y, ps["Az"], np.eye(ps["m"]), ps["Qz"], Za, H=H, d=d, c=ps["c"], a0=ps["mean"], P0=P0,
                            jitter=JITTER_DEFAULT if jitter is None else jitter, conventions=conventions)
    return (lp, ps, P0) if return_parts else lp


def solve_second_order_logp(A, B, C, D, hess_idx, hess_val, Sigma, Z, y, H=None, d=None, tol=1e-8, max_iter=1000,
                            jitter=None, conventions=None):
    """One full second-order evaluation (BASELINE configs[4]): A,B,C,D -> T,R (cycle reduction) -> g_yy, g_yu, g_uu, g_ss
    -> pruned state space -> quasi log-likelihood.  -> dict(logp, T, R, sol)."""
    from .cycle_reduction import cycle_reduction_core
    from .shared import compute_selection_matrix

    Tm, ok, _ = cycle_reduction_core(A, B, C, max_iter, tol)
    if not ok:
        return dict(logp=-np.inf, T=Tm, R=None, sol=None)
    Rm = compute_selection_matrix(B, C, D, Tm)
    S = np.flatnonzero((A != 0).any(axis=0))
    sol = second_order_solution_reduced(B, C, Tm, Rm, hess_idx, hess_val, Sigma, S=S)
    lp = pruned_kalman_logp(Tm, Rm, sol, Sigma, Z, y, H=H, d=d, jitter=jitter, conventions=conventions)
    return dict(logp=lp, T=Tm, R=Rm, sol=sol)
