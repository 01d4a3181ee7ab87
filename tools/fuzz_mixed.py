"""GPU box: batches whose draws do NOT share their structure -- more states, fewer static variables than the hints say --
so that the second passes of every kernel cascade run (flagged draws: dense cycle reduction, larger filter tile, general
filter, R Q R' for handed-on draws).  Every draw against the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from geconpy_amd import batched, workloads as wl
import oracle


def run(seed, trials, verbose=True):
    rng = np.random.default_rng(seed)
    bad = 0
    for trial in range(trials):
        n = int(rng.integers(12, 57))
        ns = int(rng.integers(2, max(3, n // 3)))
        nl = int(rng.integers(1, max(2, n // 4)))
        extra = int(rng.integers(1, 6))       # some draws have this many more states (and fewer static variables)
        k = int(rng.integers(1, 8))
        p = int(rng.integers(1, min(k, 7) + 1))
        nb = 48
        which = rng.random(nb) < 0.25
        try:
            sysm = [wl.sw_shaped_system(int(rng.integers(1 << 30)), n=n, n_state=ns + (extra if which[i] else 0), n_lead=nl, k=k)
                    for i in range(nb)]
        except Exception:
            continue
        A, B, C, D = (np.stack([s_[j] for s_ in sysm]) for j in range(4))
        q = rng.uniform(0.5, 2.0, (nb, k)) * 1e-4
        Z = np.zeros((p, n)); Z[np.arange(p), rng.choice(n, p, replace=False)] = 1.0
        T_len = int(rng.choice([3, 25]))
        y = rng.normal(0, 0.02, (T_len, p))
        H = np.full(p, 1e-4)
        # hints of the MAJORITY structure: the minority violates them
        ns_hint = int((A[~which][0] != 0).any(axis=0).sum()) if (~which).any() else None
        h_hint = int((~(A[~which][0] != 0).any(axis=0) & ~(C[~which][0] != 0).any(axis=0)).sum()) if (~which).any() else -1
        out = batched.solve_kalman_logp_batched(A, B, C, D, q, Z, y, Hdiag=H, tol=1e-10, max_iter=1000, q_mode="diag_batched",
                                                n_state_hint=ns_hint, z_selector_hint=1, options={"n_static_hint": h_hint})
        for i in rng.choice(nb, 10, replace=False):
            r = oracle.solve_kalman_logp(A[i], B[i], C[i], D[i], np.diag(q[i]), Z, y, H=np.diag(H), tol=1e-10, max_iter=1000)
            if out["status"][i] != 0 or not abs(out["logp"][i] - r["logp"]) <= 1e-8 * max(1.0, abs(r["logp"])):
                bad += 1
                if verbose:
                    print("MISMATCH", dict(n=n, ns=ns, nl=nl, extra=extra, k=k, p=p, T_len=T_len, draw=int(i), minority=bool(which[i])),
                          out["status"][i], out["logp"][i], r["logp"])
    if verbose:
        print("trials done, mismatches:", bad)
    return bad


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 30)
