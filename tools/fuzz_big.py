"""GPU box: the 65 .. 96-variable path (csrc/dsge_big.hpp) at random sizes and shapes against the oracle: cycle reduction (T,
status, iteration counts; both stopping rules), the selection matrix, and the fused solve + Kalman logp (selector and dense Z).
python tools/fuzz_big.py [seed] [trials]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import oracle
from geconpy_amd import _lib, batched
from geconpy_amd import workloads as wl


def run(seed, trials):
    rng = np.random.default_rng(seed)
    bad = 0
    worst_T = worst_lp = 0.0
    for trial in range(trials):
        n = int(rng.integers(65, 97))
        ns = int(rng.integers(8, 50))
        nl = int(rng.integers(4, max(5, n // 3)))
        k = int(rng.integers(1, 13))
        nb = 3
        try:
            sysm = [wl.sw_shaped_system(int(rng.integers(1 << 30)), n=n, n_state=ns, n_lead=nl, k=k) for _ in range(nb)]
        except Exception:
            continue
        A, B, C, D = (np.stack([s_[j] for s_ in sysm]) for j in range(4))
        tol = float(rng.choice([1e-6, 1e-8, 1e-11]))
        T, st, it = batched.cycle_reduction_batched(A, B, C, max_iter=200, tol=tol)
        R, resid = batched.selection_batched(B, C, D, T, A=A)
        for i in range(nb):
            Tc, conv, itc = oracle.cycle_reduction_core(A[i], B[i], C[i], 200, tol)
            ok = (st[i] == 0) == bool(conv)
            if conv and ok:
                eT = np.abs(T[i] - Tc).max() / max(1.0, np.abs(Tc).max())
                worst_T = max(worst_T, eT)
                Rc = oracle.compute_selection_matrix(B[i], C[i], D[i], Tc)
                eR = np.abs(R[i] - Rc).max() / max(1.0, np.abs(Rc).max())
                ok = it[i] == itc and eT <= 1e-8 and eR <= 1e-7
            if not ok:
                bad += 1
                print("CR MISMATCH", dict(seed=seed, trial=trial, n=n, ns=ns, nl=nl, k=k, tol=tol, draw=i, status=int(st[i]),
                                          conv=bool(conv), it=(int(it[i]), itc)))
        if trial % 3 == 0:
            Ts, sts, its = batched.scan_cycle_reduction_batched(A, B, C, max_iter=40, tol=1e-8)
            for i in range(nb):
                Tc, steps = oracle.scan_cycle_reduction(A[i], B[i], C[i], max_iter=40, tol=1e-8)
                if not (steps == its[i] and np.abs(Ts[i] - Tc).max() <= 1e-8 * max(1.0, np.abs(Tc).max())):
                    bad += 1
                    print("SCAN MISMATCH", dict(seed=seed, trial=trial, n=n, draw=i, steps=(int(its[i]), steps)))
        # fused logp: p observed series, selector on random variables or a dense design matrix
        p = int(rng.integers(1, 9))
        T_len = int(rng.integers(5, 40))
        obs = rng.choice(n, p, replace=False)
        if ns + p > 64:
            continue
        dense = bool(rng.integers(0, 3) == 0)
        Z = np.zeros((p, n))
        if dense:
            for s_ in range(p):
                Z[s_, rng.choice(min(n, ns + 10), 2, replace=False)] = rng.uniform(0.5, 1.5, 2)
        else:
            Z[np.arange(p), obs] = 1.0
        y = rng.standard_normal((T_len, p)) * 0.05
        if T_len > 6:
            y[3, 0] = np.nan
        Hd = rng.uniform(1e-4, 1e-2, p)
        q = rng.uniform(1e-4, 4e-4, (nb, k))
        try:
            r = batched.solve_kalman_logp_batched(A, B, C, D, q, Z, y, Hdiag=Hd, tol=1e-8, max_iter=1000, q_mode="diag_batched")
        except _lib.DsgeTooLargeError:
            continue
        for i in range(nb):
            ref = oracle.solve_kalman_logp(A[i], B[i], C[i], D[i], np.diag(q[i]), Z, y, H=np.diag(Hd), tol=1e-8, max_iter=1000)
            if not ref["success"]:
                ok = r["status"][i] != 0
            else:
                e = abs(r["logp"][i] - ref["logp"]) / max(1.0, abs(ref["logp"]))
                worst_lp = max(worst_lp, e)
                ok = r["status"][i] == 0 and e <= 1e-8
            if not ok:
                bad += 1
                print("LOGP MISMATCH", dict(seed=seed, trial=trial, n=n, ns=ns, k=k, p=p, dense=dense, draw=i, status=int(r["status"][i]),
                                            logp=float(r["logp"][i]), ref=float(ref["logp"])))
    print(f"seed {seed}: {trials} trials, {bad} mismatches; worst |T - T_oracle| {worst_T:.2e}, worst rel logp error {worst_lp:.2e}")
    return bad


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    sys.exit(1 if run(seed, trials) else 0)
