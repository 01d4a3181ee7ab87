"""GPU box: the standalone filter entry point (T, R, Q given) at random sizes against the oracle: sparse and dense
transition matrices, selector / dense / batched design matrices, p up to 16, diagonal and full Q, missing data."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from geconpy_amd import batched
import oracle


def run(seed, trials, verbose=True):
    rng = np.random.default_rng(seed)
    bad = 0
    for trial in range(trials):
        m = int(rng.integers(2, 65))
        k = int(rng.integers(1, min(m, 10) + 1))
        p = int(rng.integers(1, min(m, 16) + 1))
        T_len = int(rng.choice([1, 2, 9, 50]))
        nb = 3
        T = rng.standard_normal((nb, m, m))
        if rng.random() < 0.6:  # a DSGE-like transition: only some columns (states) non-zero
            cols = rng.random(m) < rng.uniform(0.2, 0.7)
            cols[rng.integers(m)] = True
            T = T * cols[None, None, :]
        for i in range(nb):
            rad = np.max(np.abs(np.linalg.eigvals(T[i])))
            T[i] *= rng.uniform(0.2, 0.97) / max(rad, 1e-12)
        R = rng.standard_normal((nb, m, k))
        if rng.random() < 0.5:
            Q = rng.uniform(0.5, 2.0, (nb, k)); Qor = [np.diag(Q[i]) for i in range(nb)]; qm = "diag_batched"
        else:
            L = rng.standard_normal((nb, k, k)); Q = L @ np.swapaxes(L, 1, 2) + 0.1 * np.eye(k); Qor = list(Q); qm = "full_batched"
        zk = int(rng.integers(0, 3))
        if zk == 0 and p <= m:  # selector
            Z = np.zeros((p, m)); Z[np.arange(p), rng.choice(m, p, replace=False)] = rng.uniform(0.5, 1.5, p); Zor = [Z] * nb
        elif zk == 1:
            Z = rng.standard_normal((p, m)) * (rng.random((p, m)) < 0.5); Z[np.arange(p), rng.choice(m, p, replace=False)] += 1.0; Zor = [Z] * nb
        else:
            Z = rng.standard_normal((nb, p, m)); Zor = list(Z)
        d = rng.standard_normal(p) * 0.1
        H = rng.uniform(0.1, 1.0, p)
        y = rng.standard_normal((T_len, p))
        if T_len > 2:
            y[1, 0] = np.nan
            if rng.random() < 0.3: y[2, :] = np.nan
        logp, st = batched.kalman_logp_batched(T, R, Q, Z, y, d=d, Hdiag=H, q_mode=qm)
        for i in range(nb):
            r = oracle.kalman_filter_logp(y, T[i], R[i], Qor[i], Zor[i], H=np.diag(H), d=d)
            ref = r["logp"] if isinstance(r, dict) else r
            if st[i] != 0 or not abs(logp[i] - ref) <= 1e-8 * max(1.0, abs(ref)):
                bad += 1
                if verbose:
                    print("MISMATCH", dict(m=m, k=k, p=p, T_len=T_len, zk=zk, qm=qm, draw=i), st[i], logp[i], ref)
    if verbose:
        print("trials done, mismatches:", bad)
    return bad


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 60)
