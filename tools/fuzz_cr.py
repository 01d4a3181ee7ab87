"""GPU box: cycle reduction alone at random sizes against the oracle: T, status, iteration counts, for the default kernels
and with the structure-exploiting ones switched off."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from geconpy_amd import batched, workloads as wl
import oracle


def cycle_reduction_exact(A, B, C, tol, digits=40):
    """40-digit cycle reduction with the reference's stopping rule (cycle_reduction.py:171-177) -> (T as float64, iterations)."""
    from mpmath import mp, matrix, mpf

    mp.dps = digits
    M = lambda a: matrix(a.tolist())
    norm1 = lambda X: max(sum(abs(X[i, j]) for i in range(X.rows)) for j in range(X.cols))
    A0, A1, A2, Ah = M(A), M(B), M(C), M(B)
    for it in range(1000):
        A1i = A1 ** -1
        X0, X2 = A1i * A0, A1i * A2
        m00, m02, m20, m22 = A0 * X0, A0 * X2, A2 * X0, A2 * X2
        A1, Ah, A0, A2 = A1 - m02 - m20, Ah - m20, -m00, -m22
        if norm1(A0) < mpf(tol) and norm1(A2) < mpf(tol):
            return np.array((-(Ah ** -1) * M(A)).tolist(), dtype=float), it + 1
    raise RuntimeError("no convergence")


def run(seed, trials, verbose=True):
    rng = np.random.default_rng(seed)
    bad = 0
    for trial in range(trials):
        n = int(rng.integers(3, 65))
        ns = int(rng.integers(1, max(2, n // 2)))
        nl = int(rng.integers(1, max(2, n // 3)))
        nb = 4
        try:
            sysm = [wl.sw_shaped_system(int(rng.integers(1 << 30)), n=n, n_state=ns, n_lead=nl, k=1) for _ in range(nb)]
        except Exception:
            continue
        A, B, C = (np.stack([s_[j] for s_ in sysm]) for j in range(3))
        tol = float(rng.choice([1e-6, 1e-9, 1e-12]))
        ref = [oracle.cycle_reduction_core(A[i], B[i], C[i], 200, tol) for i in range(nb)]
        exact = {}
        for opts in ({}, {"cr_four_waves": 0}, {"cr_compact": 0}):
            T, st, it = batched.cycle_reduction_batched(A, B, C, max_iter=200, tol=tol, options=opts)
            for i in range(nb):
                Tc, conv, itc = ref[i]
                # fixed bars: status and iteration count exact, |T - T_oracle| <= 1e-9 (ill-conditioned intermediates are
                # refined on the device: CR_REFINE_PIVOT_RATIO, dsge_cr_compact.hpp / dsge_cr_wide.hpp)
                ok = (st[i] == 0) == bool(conv) and (not conv or (it[i] == itc and np.abs(T[i] - Tc).max() <= 1e-9))
                if not ok and conv and st[i] == 0 and it[i] == itc:
                    # The float64 oracle (the reference's LAPACK path) is itself uncertain at this level on some draws (seed 11,
                    # n = 27: 6.4e-9 from a 40-digit evaluation).  Adjudicate against exact arithmetic: the device must be within
                    # 1e-9 of the exact T, or at least as close to it as the reference's own float64 result.
                    Tx, itx = exact.setdefault(i, cycle_reduction_exact(A[i], B[i], C[i], tol))
                    e_dev, e_orc = np.abs(T[i] - Tx).max(), np.abs(Tc - Tx).max()
                    if itx == itc and e_dev <= max(1e-9, e_orc):
                        ok = True
                        if verbose:
                            print("adjudicated against 40-digit arithmetic", opts, dict(n=n, tol=tol, draw=i),
                                  f"|T_dev - T_exact| = {e_dev:.2e}, |T_oracle - T_exact| = {e_orc:.2e}")
                if not ok:
                    bad += 1
                    if verbose:
                        print("MISMATCH", opts, dict(n=n, ns=ns, nl=nl, tol=tol, draw=i), st[i], it[i], itc,
                              np.abs(T[i] - Tc).max() if conv else None)
                        if i in exact:
                            Tx, itx = exact[i]
                            print("   against 40-digit arithmetic: iterations", itx, f"|T_dev - T_exact| = {np.abs(T[i] - Tx).max():.2e},",
                                  f"|T_oracle - T_exact| = {np.abs(Tc - Tx).max():.2e}, max|T| = {np.abs(Tx).max():.2e},",
                                  f"cond(B + C T) = {np.linalg.cond(B[i] + C[i] @ Tx):.2e}")
                            if os.environ.get("FUZZ_CR_DUMP"):
                                np.savez(os.environ["FUZZ_CR_DUMP"], A=A[i], B=B[i], C=C[i], tol=tol, T_dev=T[i], T_oracle=Tc, T_exact=Tx)
    if verbose:
        print("trials done, mismatches:", bad)
    return bad


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 60)
