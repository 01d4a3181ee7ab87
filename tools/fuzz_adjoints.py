"""GPU box: the standalone pullbacks at random sizes up to 56 variables against the oracle: policy adjoints (Kronecker
solve in numpy) and the pullback of the selection matrix (closed form)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from geconpy_amd import batched, workloads as wl
import oracle


def run(seed, trials, verbose=True):
    rng = np.random.default_rng(seed)
    bad = 0
    for trial in range(trials):
        n = int(rng.integers(3, 57)) if rng.random() < 0.7 else int(rng.integers(44, 57))
        ns = int(rng.integers(1, max(2, n // 2)))
        nl = int(rng.integers(1, max(2, n // 3)))
        k = int(rng.integers(1, min(n, 9) + 1))
        nb = 2
        try:
            sysm = [wl.sw_shaped_system(int(rng.integers(1 << 30)), n=n, n_state=ns, n_lead=nl, k=k) for _ in range(nb)]
        except Exception:
            continue
        A, B, C, D, T = (np.stack([s_[j] for s_ in sysm]) for j in range(5))
        R = np.stack([oracle.compute_selection_matrix(B[i], C[i], D[i], T[i]) for i in range(nb)])
        T_bar = rng.standard_normal(T.shape) * (T != 0).any(axis=1)[:, None, :]   # cotangent on the state columns
        R_bar = rng.standard_normal(R.shape)
        Ab, Bb, Cb, st = batched.policy_adjoints_batched(B, C, T, T_bar)
        Bs, Cs, Ds, Ts = batched.selection_adjoints_batched(B, C, T, R, R_bar)
        for i in range(nb):
            S, ST, STT = oracle.policy_function_adjoints(A[i], B[i], C[i], T[i], T_bar[i])
            scale = max(1.0, np.abs(S).max())
            e1 = max(np.abs(Ab[i] - S).max(), np.abs(Bb[i] - ST).max(), np.abs(Cb[i] - STT).max()) / scale
            M = C[i] @ T[i] + B[i]
            G = -np.linalg.solve(M.T, R_bar[i])
            GR = G @ R[i].T
            ref = (GR, GR @ T[i].T, G, C[i].T @ GR)
            e2 = max(np.abs(x - r_).max() / max(1.0, np.abs(r_).max()) for x, r_ in zip((Bs[i], Cs[i], Ds[i], Ts[i]), ref))
            # The device sums the series S = sum_k G^k H (T')^k by doubling; a draw whose Stein residual shows lost digits (a
            # non-normal G = -(B + C T)^-T C') gets one step of iterative refinement in a second pass (adjoint_kernel<BS, true>):
            # the reference's Kronecker LU level, 1e-9 asserted.
            if st[i] == 0 and e1 > 1e-10 and verbose:
                res = lambda S_: np.abs(M.T @ S_ + C[i].T @ S_ @ T[i].T + T_bar[i]).max() / max(1.0, np.abs(T_bar[i]).max())
                print("  policy adjoints differ by", f"{e1:.1e}", "residual device", f"{res(Ab[i]):.1e}", "oracle", f"{res(S):.1e}", dict(n=n, ns=ns, nl=nl))
            if st[i] != 0 or not (e1 <= 1e-9 and e2 <= 1e-9):
                bad += 1
                if verbose:
                    print("MISMATCH", dict(n=n, ns=ns, nl=nl, k=k, draw=i), st[i], e1, e2)
    if verbose:
        print("trials done, mismatches:", bad)
    return bad


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 40)
