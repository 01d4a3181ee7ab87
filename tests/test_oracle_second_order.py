"""the second-order oracle (oracle/second_order.py) pinned on a model with a CLOSED-FORM policy function --
Brock-Mirman growth with log utility and full depreciation:

    1 / c_t = beta E_t[ alpha e^{z_{t+1}} k_t^{alpha-1} / c_{t+1} ],   c_t + k_t = e^{z_t} k_{t-1}^alpha,   z_t = rho z_{t-1} + u_t
    exact:  k_t = alpha beta e^{z_t} k_{t-1}^alpha,   c_t = (1 - alpha beta) e^{z_t} k_{t-1}^alpha      (independent of sigma)

The reference has no second-order solver (gEconpy/model/perturbation.py:97-98), so this is what stands in for golden
vectors: the second derivatives of the exact policy at the steady state, computed symbolically."""
import numpy as np
import pytest
import sympy as sp
from numpy.testing import assert_allclose

import oracle
from oracle import second_order as so


def _brock_mirman(alpha=0.33, beta=0.96, rho=0.9):
    names = ["c", "k", "z"]
    ym = sp.symbols("cm km zm")
    y0 = sp.symbols("c0 k0 z0")
    yp = sp.symbols("cp kp zp")
    u = sp.symbols("u")
    a, b, r = sp.Float(alpha), sp.Float(beta), sp.Float(rho)
    F = sp.Matrix([
        1 / y0[0] - b * a * sp.exp(yp[2]) * y0[1] ** (a - 1) / yp[0],
        y0[0] + y0[1] - sp.exp(y0[2]) * ym[1] ** a,
        y0[2] - r * ym[2] - u,
    ])
    kss = (alpha * beta) ** (1 / (1 - alpha))
    css = (1 - alpha * beta) * kss ** alpha
    ss = {**{s: v for s, v in zip(ym, (css, kss, 0.0))}, **{s: v for s, v in zip(y0, (css, kss, 0.0))},
          **{s: v for s, v in zip(yp, (css, kss, 0.0))}, u: 0.0}
    z = list(ym) + list(y0) + list(yp) + [u]
    J = np.array(F.jacobian(z).subs(ss), dtype=float)
    n, k, m = 3, 1, 10
    H = np.zeros((n, m * m))
    for i in range(n):
        Hi = sp.hessian(F[i], z).subs(ss)
        H[i] = np.array(Hi, dtype=float).reshape(-1)
    A, B, C, D = J[:, :3], J[:, 3:6], J[:, 6:9], J[:, 9:10]
    # exact policy g(y-, u)
    zt = r * ym[2] + u
    g = sp.Matrix([(1 - a * b) * sp.exp(zt) * ym[1] ** a, a * b * sp.exp(zt) * ym[1] ** a, zt])
    s0 = {ym[0]: css, ym[1]: kss, ym[2]: 0.0, u: 0.0}
    T = np.array(g.jacobian(list(ym)).subs(s0), dtype=float)
    R = np.array(g.jacobian([u]).subs(s0), dtype=float)
    x = list(ym) + [u]
    G2 = np.zeros((n, 4, 4))
    for i in range(n):
        G2[i] = np.array(sp.hessian(g[i], x).subs(s0), dtype=float)
    assert np.abs(np.array(F.subs(ss), dtype=float)).max() < 1e-14
    del names
    return A, B, C, D, H, T, R, G2


def test_first_order_of_the_closed_form_is_what_the_solvers_find():
    A, B, C, D, H, T, R, _ = _brock_mirman()
    Tcr, conv, _it = oracle.cycle_reduction_core(A, B, C, 1000, 1e-13)
    assert conv
    assert_allclose(Tcr, T, atol=1e-11)
    assert_allclose(oracle.compute_selection_matrix(B, C, D, Tcr), R, atol=1e-11)
    Tg, ok = oracle.gensys_T_success(A, B, C, D)[:2]
    assert ok
    assert_allclose(Tg, T, atol=1e-10)


@pytest.mark.parametrize("pars", [dict(), dict(alpha=0.25, beta=0.99, rho=0.5), dict(alpha=0.4, beta=0.9, rho=0.97)])
def test_second_order_matches_the_exact_policy(pars):
    A, B, C, D, H, T, R, G2 = _brock_mirman(**pars)
    n, k = 3, 1
    Sigma = np.array([[0.01 ** 2]])
    sol = so.second_order_solution(A, B, C, D, H, T, R, Sigma)
    assert so.second_order_residual(A, B, C, D, H, T, R, Sigma, sol) < 1e-10
    g_yy = sol["g_yy"].reshape(n, n, n)
    g_yu = sol["g_yu"].reshape(n, n, k)
    g_uu = sol["g_uu"].reshape(n, k, k)
    assert_allclose(g_yy, G2[:, :3, :3], atol=1e-9)
    assert_allclose(g_yu[:, :, 0], G2[:, :3, 3], atol=1e-9)
    assert_allclose(g_uu[:, 0, 0], G2[:, 3, 3], atol=1e-9)
    assert_allclose(sol["g_ss"], 0.0, atol=1e-12)  # the exact policy does not depend on the shock variance


def test_pruned_augmented_system_reproduces_the_pruned_simulation():
    """z' = c + Az z + xi: the deterministic part reproduces one step of the pruned simulation exactly when the shock is
    switched off, the unconditional mean and covariance match a long seeded simulation (statistical tolerances), and the
    augmented system is what the Gaussian quasi-likelihood runs on (oracle.kalman_filter_logp accepts it)."""
    A, B, C, D, H, T, R, _ = _brock_mirman()
    n = 3
    Sigma = np.array([[0.05 ** 2]])
    sol = so.second_order_solution(A, B, C, D, H, T, R, Sigma)
    ps = so.pruned_state_space(T, R, sol, Sigma)
    Az, c, Qz = ps["Az"], ps["c"], ps["Qz"]
    assert np.abs(np.linalg.eigvals(Az)).max() < 1.0 and np.linalg.eigvalsh(Qz).min() > -1e-14
    rng = np.random.default_rng(0)
    # exact one-step identity: E[z' | z] = c + Az z
    xf, xs = rng.standard_normal(n) * 0.1, rng.standard_normal(n) * 0.01
    z = np.concatenate([xf, xs, np.kron(xf, xf)])
    us = rng.standard_normal((4000, 1)) * 0.05
    nxt = np.zeros(z.size)
    for u in us:
        f1 = T @ xf + R @ u
        s1 = (T @ xs + 0.5 * sol["g_yy"] @ np.kron(xf, xf) + sol["g_yu"] @ np.kron(xf, u) + 0.5 * sol["g_uu"] @ np.kron(u, u) +
              0.5 * sol["g_ss"])
        nxt += np.concatenate([f1, s1, np.kron(f1, f1)])
    nxt /= len(us)
    pred = c + Az @ z
    assert_allclose(nxt, pred, atol=4 * 0.05 / np.sqrt(len(us)) * 3)
    # unconditional moments against a long simulation
    shocks = rng.standard_normal((200_000, 1)) * 0.05
    xf_p, xs_p = so.simulate_pruned(T, R, sol, shocks)
    zs = np.hstack([xf_p, xs_p, np.einsum("ti,tj->tij", xf_p, xf_p).reshape(len(xf_p), -1)])[1000:]
    assert_allclose(zs.mean(axis=0), ps["mean"], atol=5e-3 * np.abs(ps["mean"]).max() + 2e-4)
    import scipy.linalg as sla

    Pz = sla.solve_discrete_lyapunov(Az, Qz)
    emp = np.cov(zs.T)
    assert np.abs(emp - Pz).max() <= 0.05 * np.abs(Pz).max()
    # the quasi-likelihood: observed y = x_f + x_s (consumption and capital), Gaussian filter on the augmented system
    Z = np.zeros((2, 2 * n + n * n))
    Z[0, 0] = Z[0, n] = 1.0
    Z[1, 1] = Z[1, n + 1] = 1.0
    y = (xf_p + xs_p)[1:201, :2]
    lp = oracle.kalman_filter_logp(y, Az, np.eye(Az.shape[0]), Qz, Z, H=np.diag([1e-6, 1e-6]), c=c)
    assert np.isfinite(lp)
