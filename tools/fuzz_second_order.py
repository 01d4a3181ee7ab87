"""GPU box: the second-order path (coefficients + pruned-state-space likelihood) at random sizes and structures against
oracle/second_order.py: 4..34 variables, 1..14 states (pruned state up to 208), observed states and non-states, missing
observations, short and long samples (the steady-state switch), with and without an observation intercept."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from geconpy_amd import batched, workloads as wl
from oracle import second_order as so


def run(seed, trials, verbose=True):
    rng = np.random.default_rng(seed)
    bad = done = 0
    while done < trials:
        n = int(rng.integers(4, 35))
        ns = int(rng.integers(1, min(15, max(2, n // 2)) + 0))
        nl = int(rng.integers(1, max(2, n // 3)))
        k = int(rng.integers(1, min(ns, 6) + 1))
        p = int(rng.integers(1, min(5, n) + 1))
        obs = np.sort(rng.choice(n, size=p, replace=False))
        u = ns + int(np.sum(obs >= ns))  # the generator puts the states first
        if 2 * u + ns * (ns + 1) // 2 > 208 or u > 40:
            continue
        nb = 2
        try:
            sysm = [wl.sw_shaped_system(int(rng.integers(1 << 30)), n=n, n_state=ns, n_lead=nl, k=k) for _ in range(nb)]
        except Exception:  # noqa: BLE001  (the generator rejects some shapes)
            continue
        done += 1
        A, B, C, D = (np.stack([s_[j] for s_ in sysm]) for j in range(4))
        idx = wl.second_order_hessian_pattern(A[0], C[0], k, nnz_per_eq=int(rng.integers(2, 8)), seed=int(rng.integers(1 << 30)))
        val = rng.standard_normal((nb, len(idx)))
        q = rng.uniform(0.5e-4, 4e-4, (nb, k))
        Z = np.zeros((p, n))
        Z[np.arange(p), obs] = rng.choice([1.0, 0.5, 2.0], size=p)
        T_len = int(rng.choice([12, 40, 120]))
        y = rng.normal(0, 0.02, (T_len, p))
        if T_len > 12:
            y[5, 0] = np.nan
            y[9] = np.nan
        H = rng.uniform(0.5e-5, 2e-5, p)
        d = rng.normal(0, 0.01, p) if rng.random() < 0.5 else None
        out = batched.second_order_logp_batched(A, B, C, D, idx, val, q, Z, y, d=d, Hdiag=H, tol=1e-12, return_solution=True)
        for i in range(nb):
            r = so.solve_second_order_logp(A[i], B[i], C[i], D[i], idx, val[i], np.diag(q[i]), Z, y, H=np.diag(H), d=d, tol=1e-12)
            errs = {key: float(np.abs(out[key][i] - r["sol"][key]).max() / max(1.0, np.abs(r["sol"][key]).max()))
                    for key in ("g_yy", "g_yu", "g_uu", "g_ss")}
            el = abs(out["logp"][i] - r["logp"]) / abs(r["logp"])
            if out["status"][i] != 0 or max(errs.values()) > 1e-9 or not el <= 1e-8:
                bad += 1
                if verbose:
                    print("MISMATCH", dict(n=n, ns=ns, nl=nl, k=k, p=p, T_len=T_len, draw=i), int(out["status"][i]), errs, el)
    if verbose:
        print("trials done, mismatches:", bad)
    return bad


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 20)
