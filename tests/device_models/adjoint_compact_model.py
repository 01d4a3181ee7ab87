"""Executable model of adj_stein_solve_compact (csrc/dsge_kernels.hpp): the adjoint of the policy equation A + B T + C T T = 0,

    M' S + C' S T' = -T_bar,   M = B + C T        (gEconpy/solvers/shared.py:12-71 solves it as an n^2 x n^2 Kronecker system)

written as the Stein equation S = H + G S T' (H = -M^-T T_bar, G = -M^-T C') and solved by doubling -- on the full n x n form
(rounds 2-5: adj_stein_solve) and on the compact form of round 6:  C has non-zero columns only for the variables with a lead (L),
T only for the states (St), so G = -Wm V with Wm = M^-T E_L, V = C_L', and Z = V S restricted to the columns St solves the
nl x ns equation  Zs = V H[:, St] + Gs Zs Tss',  Gs = -V Wm,  Tss = T[St, St];  S = H - (Wm Zs) T[:, St]'."""
import numpy as np


def kronecker_solve(M, C, T, T_bar):
    """The reference's route: (I kron M' + T kron C') vec(S) = -vec(T_bar), column-major vec."""
    n = M.shape[0]
    K = np.kron(np.eye(n), M.T) + np.kron(T, C.T)
    return np.linalg.solve(K, -T_bar.flatten(order="F")).reshape((n, n), order="F")


def full_doubling(M, C, T, T_bar, max_doublings=64):
    """-> (S, largest |entry| met among the powers G^(2^k))."""
    Mit = np.linalg.inv(M.T)
    S, G, F = -Mit @ T_bar, -Mit @ C.T, T.T.copy()
    growth = np.abs(G).max()
    with np.errstate(all="ignore"):
        for _ in range(max_doublings):
            inc = G @ S @ F
            S = S + inc
            G, F = G @ G, F @ F
            if not np.isfinite(inc).all() or np.abs(inc).max() <= 1e-17 * np.abs(S).max():
                break
            growth = max(growth, np.abs(G).max())
    return S, growth


def compact_doubling(M, C, T, T_bar, max_doublings=64):
    """-> (S, largest |entry| met among the powers Gs^(2^k), nl, ns)."""
    L = np.flatnonzero(np.any(C != 0, axis=0))
    St = np.flatnonzero(np.any(T != 0, axis=0))
    Mit = np.linalg.inv(M.T)
    H, Wm, CL = -Mit @ T_bar, Mit[:, L], C[:, L]
    Gs, Z, Fk = -CL.T @ Wm, CL.T @ H[:, St], T[np.ix_(St, St)].copy()
    growth = np.abs(Gs).max()
    for _ in range(max_doublings):
        inc = Gs @ (Z @ Fk.T)
        Z = Z + inc
        Gs, Fk = Gs @ Gs, Fk @ Fk
        if np.abs(inc).max() <= 1e-17 * np.abs(Z).max():
            break
        growth = max(growth, np.abs(Gs).max())
    return H - (Wm @ Z) @ T[:, St].T, growth, len(L), len(St)
