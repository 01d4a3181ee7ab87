"""GPU box: the raw-pencil gensys entry point on random pencils against the oracle: the pencils gensys_setup builds for
random models, and the same pencils under a random equivalence transformation (P g0, P g1, P psi, P pi -- the solution is
invariant) so that the structural sparsity of the model pencil is gone."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from geconpy_amd import batched, workloads as wl
import oracle
from oracle.gensys_qz import gensys_setup


def run(seed, trials, verbose=True):
    rng = np.random.default_rng(seed)
    bad = 0
    for trial in range(trials):
        n = int(rng.integers(3, 37))
        ns = int(rng.integers(1, max(2, n // 2)))
        nl = int(rng.integers(1, max(2, n // 3)))
        if n + nl > 50:
            continue
        k = int(rng.integers(1, min(n, 5) + 1))
        nb = 2
        try:
            sysm = [wl.sw_shaped_system(int(rng.integers(1 << 30)), n=n, n_state=ns, n_lead=nl, k=k) for _ in range(nb)]
        except Exception:
            continue
        pencils = [gensys_setup(*s_[:4]) for s_ in sysm]
        N = pencils[0][0].shape[0]
        if any(pc[0].shape[0] != N or pc[4].shape != pencils[0][4].shape for pc in pencils):
            continue
        mix = rng.random() < 0.5
        g0, g1, cc, psi, pi = [], [], [], [], []
        for pc in pencils:
            P = np.eye(N) + (0.3 * rng.standard_normal((N, N)) if mix else 0.0)
            g0.append(P @ pc[0]); g1.append(P @ pc[1]); cc.append(P @ np.asarray(pc[2]).reshape(N)); psi.append(P @ pc[3]); pi.append(P @ pc[4])
        g0, g1, cc, psi, pi = (np.stack(x) for x in (g0, g1, cc, psi, pi))
        out = batched.gensys_pencil_batched(g0, g1, psi, pi, c=cc)
        for i in range(nb):
            r = oracle.gensys(g0[i], g1[i], cc[i].reshape(N, 1), psi[i], pi[i])
            eu = r[7]
            good = list(out["eu"][i][:2]) == [eu[0], eu[1]]
            if eu[0] == 1 and eu[1] == 1:
                good = good and np.abs(out["G1"][i] - r[0]).max() <= 1e-7 and np.abs(out["impact"][i] - r[2]).max() <= 1e-7
            if not good:
                bad += 1
                if verbose:
                    print("MISMATCH", dict(n=n, ns=ns, nl=nl, k=k, N=N, mix=mix, draw=i), out["eu"][i], eu,
                          np.abs(out["G1"][i] - r[0]).max() if eu[0] == 1 and eu[1] == 1 else None)
    if verbose:
        print("trials done, mismatches:", bad)
    return bad


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 40)
