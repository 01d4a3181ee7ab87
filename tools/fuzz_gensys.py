"""GPU box: gensys alone at random sizes against the oracle (LAPACK ordered QZ): T, eu; and the Blanchard-Kahn counts."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from geconpy_amd import batched, workloads as wl
import oracle


def run(seed, trials, verbose=True):
    rng = np.random.default_rng(seed)
    bad = 0
    for trial in range(trials):
        n = int(rng.integers(3, 49))
        ns = int(rng.integers(1, max(2, n // 2)))
        nl = int(rng.integers(1, max(2, n // 3)))
        if n + nl > 60:
            continue
        k = int(rng.integers(1, min(n, 6) + 1))
        nb = 3
        try:
            sysm = [wl.sw_shaped_system(int(rng.integers(1 << 30)), n=n, n_state=ns, n_lead=nl, k=k) for _ in range(nb)]
        except Exception:
            continue
        A, B, C, D, Tst = (np.stack([s_[j] for s_ in sysm]) for j in range(5))
        if rng.random() < 0.2:  # an explosive draw: no stable solution (eu != [1, 1])
            A[1] = A[1] * 30.0
        out = batched.gensys_batched(A, B, C, D)
        bk = batched.bk_eigenvalues_batched(A, B, C)
        for i in range(nb):
            Tref, succ, eu = oracle.gensys_T_success(A[i], B[i], C[i], D[i])
            ok_d = bool(out["success"][i])
            good = (ok_d == succ) and (list(out["eu"][i][:2]) == [int(eu[0]), int(eu[1])] or not succ)
            if succ and ok_d:
                err = np.abs(out["T"][i] - Tref).max()
                if err > 1e-7:
                    # accepted only if the problem itself is that uncertain: the oracle's two algorithms (LAPACK QZ and cycle
                    # reduction) disagree by a comparable amount (seed 14, n = 22: 1.1e-6 between them, device 2.2e-6)
                    Tcr, conv, _ = oracle.cycle_reduction_core(A[i], B[i], C[i], 500, 1e-13)
                    spread = np.abs(Tref - Tcr).max() if conv else 0.0
                    if err <= 10.0 * spread:
                        if verbose:
                            print("conditioning outlier", dict(n=n, draw=i), f"|dT| = {err:.2e}, oracle QZ vs oracle cycle reduction = {spread:.2e}")
                    else:
                        good = False
            sat, n_fwd, n_unst = oracle.check_bk_condition(A[i], B[i], C[i], D[i])
            good = good and int(bk["n_forward"][i]) == n_fwd and int(bk["n_unstable"][i]) == n_unst
            if not good:
                bad += 1
                if verbose:
                    print("MISMATCH", dict(n=n, ns=ns, nl=nl, k=k, draw=i), ok_d, succ, out["eu"][i], eu,
                          np.abs(out["T"][i] - Tref).max() if succ and ok_d else None, (int(bk["n_forward"][i]), n_fwd),
                          (int(bk["n_unstable"][i]), n_unst))
                    if os.environ.get("FUZZ_GENSYS_DUMP"):
                        np.savez(os.environ["FUZZ_GENSYS_DUMP"], A=A[i], B=B[i], C=C[i], D=D[i], T_dev=out["T"][i], T_oracle=Tref if succ else 0.0)
    if verbose:
        print("trials done, mismatches:", bad)
    return bad


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 40)
