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


def two_step_pullback(B, C, T, R, R_bar, T_bar):
    """The reference's order (shared.py:74-75 then :12-71): the pullback of R = -(B + C T)^-1 D first -- X = M^-T R_bar, D_bar = -X,
    M_bar = -X R', which sends M_bar to B, M_bar T' to C and C' M_bar to T -- then the policy adjoints of the total T_bar."""
    M = B + C @ T
    X = np.linalg.solve(M.T, R_bar)
    M_bar = -X @ R.T
    S = kronecker_solve(M, C, T, T_bar + C.T @ M_bar)
    return dict(A_bar=S, B_bar=M_bar + S @ T.T, C_bar=M_bar @ T.T + S @ T.T @ T.T, D_bar=-X)


def fused_pullback(B, C, T, R, R_bar, T_bar, max_doublings=64):
    """adjoint_kernel<BS, false, true>: ONE elimination of M' with the right-hand sides [T_bar | E_L | R_bar]; the cotangent C' M_bar
    enters through  H = H_f + Wm (C_L' X) R'  (M^-T C' = Wm C_L');  B_bar = -X R' + S T',  C_bar = B_bar T'."""
    M = B + C @ T
    L = np.flatnonzero(np.any(C != 0, axis=0))
    St = np.flatnonzero(np.any(T != 0, axis=0))
    sol = np.linalg.solve(M.T, np.hstack([T_bar, np.eye(M.shape[0])[:, L], R_bar]))
    n, nl = M.shape[0], len(L)
    H_f, Wm, X = -sol[:, :n], sol[:, n:n + nl], sol[:, n + nl:]
    CL = C[:, L]
    H = H_f + Wm @ (CL.T @ X) @ R.T
    Gs, Z, Fk = -CL.T @ Wm, CL.T @ H[:, St], T[np.ix_(St, St)].copy()
    for _ in range(max_doublings):
        inc = Gs @ (Z @ Fk.T)
        Z = Z + inc
        Gs, Fk = Gs @ Gs, Fk @ Fk
        if np.abs(inc).max() <= 1e-17 * np.abs(Z).max():
            break
    S = H - (Wm @ Z) @ T[:, St].T
    B_bar = -X @ R.T + S @ T.T
    return dict(A_bar=S, B_bar=B_bar, C_bar=B_bar @ T.T, D_bar=-X)
