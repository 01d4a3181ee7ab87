"""CPU model (numpy + mpmath) behind adj_stein_fixed_point (csrc/dsge_kernels.hpp): on the SW-shaped draw 752 -- M = B + C T with
cond 3e8, G = -M^-T C' with entries of 2e7 -- compare, against a 50-digit evaluation of S = sum_k G^k H (T')^k:
  * the reference's Kronecker LU (oracle.policy_function_adjoints, shared.py:12-71),
  * Smith doubling with the explicit G (the device's first pass),
  * the plain fixed point with the explicit G,
  * the fixed point with an LU solve per sweep (what the device's second pass falls back to).
Usage: python tools/adjoint_fixed_point_model.py [draw]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings

import mpmath as mp
import numpy as np
import scipy.linalg as sl

import oracle
from geconpy_amd import workloads as wl


def exact_stein(B, C, T, T_bar, terms=100, dps=50):
    """S = sum_k G^k H (T')^k, G = -M^-T C', H = -M^-T T_bar, in `dps`-digit arithmetic."""
    mp.mp.dps = dps
    M = mp.matrix((B + C @ T).tolist())
    Mi = mp.inverse(M.T)
    G = -(Mi * mp.matrix(C.T.tolist()))
    Tt = mp.matrix(T.T.tolist())
    term = -(Mi * mp.matrix(T_bar.tolist()))
    S = term.copy()
    for _ in range(terms):
        term = G * term * Tt
        S = S + term
    return np.array(S.tolist(), dtype=float)


def main():
    draw = int(sys.argv[1]) if len(sys.argv) > 1 else 752
    b = wl.sw_shaped_batch(draw + 1)
    A, B, C = (b[x][draw] for x in "ABC")
    T = oracle.cycle_reduction_numpy(A, B, C, tol=1e-8, max_iter=1000)[0]
    n = T.shape[0]
    T_bar = np.random.default_rng(1).standard_normal((n, n))
    M = B + C @ T
    G = -np.linalg.solve(M.T, C.T)
    H = -np.linalg.solve(M.T, T_bar)
    Sx = exact_stein(B, C, T, T_bar)
    rel = lambda X: float(np.abs(X - Sx).max() / np.abs(Sx).max())
    print(f"draw {draw}: cond(M) {np.linalg.cond(M):.2e}, max|G| {np.abs(G).max():.2e}, rho(G) {np.abs(np.linalg.eigvals(G)).max():.3f}, "
          f"rho(T) {np.abs(np.linalg.eigvals(T)).max():.3f}, max|S| {np.abs(Sx).max():.2e}")
    print("reference (Kronecker LU)                vs exact:", rel(oracle.policy_function_adjoints(A, B, C, T, T_bar)[0]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Sk, Gk, Tk = H.copy(), G.copy(), T.copy()
        for _ in range(12):
            Sk = Sk + Gk @ Sk @ Tk.T
            Gk = Gk @ Gk
            Tk = Tk @ Tk
    print("Smith doubling, explicit G              vs exact:", rel(Sk))
    X = H.copy()
    for _ in range(200):
        X = H + G @ (X @ T.T)
    print("fixed point, explicit G, 200 sweeps      vs exact:", rel(X))
    lu = sl.lu_factor(M.T)
    X = np.zeros((n, n))
    best, since = np.inf, 0
    for j in range(200):
        Xn = -sl.lu_solve(lu, T_bar + C.T @ X @ T.T)
        d = np.abs(Xn - X).max()
        X = Xn
        if d < 0.9 * best:  # the device's stopping rule: the step has not improved by 10 % for eight sweeps
            best, since = d, 0
        else:
            since += 1
        if d <= 1e-15 * np.abs(X).max() or (d <= 1e-6 * np.abs(X).max() and since >= 8):
            break
    print(f"fixed point, LU solve per sweep, {j + 1} sweeps vs exact:", rel(X))


if __name__ == "__main__":
    main()
