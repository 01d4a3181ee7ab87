"""GPU box: autocorrelation matrices and the stationary covariance at random sizes against the oracle."""
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
        k = int(rng.integers(1, min(m, 8) + 1))
        nb = 2
        T = rng.standard_normal((nb, m, m)) * (rng.random((1, 1, m)) < rng.uniform(0.3, 1.0))
        for i in range(nb):
            rad = np.max(np.abs(np.linalg.eigvals(T[i])))
            T[i] *= rng.uniform(0.2, 0.95) / max(rad, 1e-12)
        R = rng.standard_normal((nb, m, k))
        Q = rng.uniform(0.5, 2.0, (nb, k))
        n_lags = int(rng.integers(1, 9)); lag_step = int(rng.integers(1, 4)); corr = bool(rng.integers(0, 2))
        obs = rng.random() < 0.5
        Z = H = None
        if obs:
            p = int(rng.integers(1, min(m, 8) + 1))
            Z = rng.standard_normal((p, m)); H = rng.uniform(0.1, 1.0, p)
        acf, st, sig = batched.autocorrelation_matrices_batched(T, R, Q, n_lags=n_lags, lag_step=lag_step, Z=Z, Hdiag=H,
                                                               correlation=corr, q_mode="diag_batched", return_sigma=True)
        P0, RQR, st2 = batched.lyapunov_batched(T, R, Q, q_mode="diag_batched")
        for i in range(nb):
            ref = oracle.autocorrelation_matrices(T[i], R[i], np.diag(Q[i]), n_lags=n_lags, lag_step=lag_step, Z=Z,
                                                  H=None if H is None else np.diag(H), correlation=corr)
            Sref = oracle.solve_discrete_lyapunov(T[i], R[i] @ np.diag(Q[i]) @ R[i].T)
            sc = max(1.0, np.abs(Sref).max())
            e = max(np.abs(acf[i] - ref).max() / max(1.0, np.abs(ref).max()), np.abs(sig[i] - Sref).max() / sc, np.abs(P0[i] - Sref).max() / sc)
            if st[i] != 0 or st2[i] != 0 or not e <= 1e-9:
                bad += 1
                if verbose:
                    print("MISMATCH", dict(m=m, k=k, n_lags=n_lags, lag_step=lag_step, corr=corr, obs=obs, draw=i), st[i], st2[i], e)
    if verbose:
        print("trials done, mismatches:", bad)
    return bad


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 40)
