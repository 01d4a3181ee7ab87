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
