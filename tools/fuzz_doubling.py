"""GPU box: dsge_options.gensys_doubling against the oracle (tools/fuzz_gensys.py under the option) and against the QZ path on batches
with a random mix of regular, explosive, indeterminate, near-unit-root and rank-deficient systems: eu and status EXACTLY equal."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from geconpy_amd import _lib, batched, workloads as wl
import fuzz_gensys

DBL = {"gensys_doubling": 1}


def mixed(seed, trials):
    rng = np.random.default_rng(seed)
    bad = 0
    for trial in range(trials):
        n = int(rng.integers(4, 45)); ns = int(rng.integers(1, max(2, n // 2))); nl = int(rng.integers(1, max(2, n // 3)))
        k = int(rng.integers(1, min(n, 6) + 1)); nb = 24
        sysm = [wl.sw_shaped_system(int(rng.integers(1 << 30)), n=n, n_state=ns, n_lead=nl, k=k) for _ in range(nb)]
        A, B, C, D, Tst = (np.stack([s_[j] for s_ in sysm]) for j in range(5))
        for i in range(nb):
            u = rng.random()
            M = B[i] + C[i] @ Tst[i]
            if u < 0.15:      # explosive state block
                A[i] *= rng.uniform(5, 40)
            elif u < 0.30:    # indeterminacy: a root of G beyond the unit circle
                G = np.linalg.solve(M, C[i]); G *= rng.uniform(1.05, 3.0) / np.max(np.abs(np.linalg.eigvals(G)))
                C[i] = M @ G; B[i] = M - C[i] @ Tst[i]
            elif u < 0.40:    # roots near the unit circle, on either side
                T2 = Tst[i].copy(); S = T2[:ns, :ns]
                T2[:, :ns] *= (1.0 + rng.choice([-1, 1]) * 10.0 ** rng.uniform(-9, -3)) / np.max(np.abs(np.linalg.eigvals(S)))
                A[i] = -M @ T2; B[i] = M - C[i] @ T2
            elif u < 0.45:    # a zero equation: coincident zeros
                r = int(rng.integers(n)); A[i, r] = B[i, r] = C[i, r] = 0.0
            elif u < 0.50:    # a lead column below the tolerance
                c = n - 1; C[i][:, c] *= 1e-12
        qz = batched.gensys_batched(A, B, C, D, tol=1e-8, options={"gensys_doubling": 0})
        db = batched.gensys_batched(A, B, C, D, tol=1e-8, options=DBL)
        same = np.array_equal(qz["eu"], db["eu"]) and np.array_equal(qz["status"], db["status"])
        ok = qz["success"]
        dT = np.abs(qz["T"] - db["T"]).reshape(nb, -1).max(axis=1)
        scale = np.maximum(1.0, np.abs(qz["T"]).reshape(nb, -1).max(axis=1))
        if not same or (ok.any() and (dT[ok] / scale[ok]).max() > 1e-7):
            bad += 1
            print("MISMATCH", dict(seed=seed, trial=trial, n=n, ns=ns, nl=nl), "eu rows differing:", np.flatnonzero((qz["eu"] != db["eu"]).any(axis=1)).tolist(),
                  "max dT", float((dT[ok] / scale[ok]).max()) if ok.any() else None)
    return bad


if __name__ == "__main__":
    for seed in [int(a) for a in sys.argv[1:]] or [9101, 9102]:
        t0 = time.time()
        with _lib.options_scope(DBL):
            b1 = fuzz_gensys.run(seed, 1500, verbose=False)
        print(f"fuzz_gensys under gensys_doubling seed={seed} trials=1500: mismatches={b1}  ({time.time() - t0:.0f} s)", flush=True)
        t0 = time.time()
        b2 = mixed(seed, 150)
        print(f"mixed batches (24 draws each, half of them non-regular) seed={seed} trials=150: mismatches={b2}  ({time.time() - t0:.0f} s)", flush=True)
