"""GPU box: the chunked host path (batch >= 512: two, >= 2048: four chunks on two streams) against the device-resident
single pipeline: bit-identical logp / status at random sizes, solver variants included."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import batched, workloads as wl
from geconpy_amd.engine import LogpEngine


def run(seed, trials, verbose=True):
    rng = np.random.default_rng(seed)
    eng = LogpEngine(torch.device("cuda", 0))
    bad = 0
    for trial in range(trials):
        n = int(rng.integers(8, 57))
        ns = int(rng.integers(2, max(3, n // 2)))
        nl = int(rng.integers(1, max(2, n // 3)))
        k = int(rng.integers(1, 8)); p = int(rng.integers(1, min(k, 7) + 1))
        nb = int(rng.choice([520, 2100]))
        base = [wl.sw_shaped_system(int(rng.integers(1 << 30)), n=n, n_state=ns, n_lead=nl, k=k) for _ in range(8)]
        idx = rng.integers(0, 8, nb)
        A, B, C, D = (np.stack([base[j][q] for j in idx]) for q in range(4))
        A = A * (1.0 + 1e-4 * rng.standard_normal((nb, 1, 1)))   # distinct draws
        q = rng.uniform(0.5, 2.0, (nb, k)) * 1e-4
        Z = np.zeros((p, n)); Z[np.arange(p), rng.choice(n, p, replace=False)] = 1.0
        y = rng.normal(0, 0.02, (20, p)); H = np.full(p, 1e-4)
        solver = str(rng.choice(["cycle_reduction", "gensys"])) if n + nl <= 56 else "cycle_reduction"
        out = batched.solve_kalman_logp_batched(A, B, C, D, q, Z, y, Hdiag=H, tol=1e-9, max_iter=500, q_mode="diag_batched", solver=solver)
        dev = [eng.to_device(x) for x in (A, B, C, D)]
        hints = eng.structure_hints(dev[0], eng.to_device(Z))
        lp, st = eng.solve_kalman_logp(*dev, eng.to_device(q), eng.to_device(Z), eng.to_device(y), Hdiag=eng.to_device(H), q_mode=1,
                                       tol=1e-9, max_iter=500, n_state_hint=hints[0], z_selector_hint=hints[1], solver=solver)
        torch.cuda.synchronize()
        lp, st = lp.cpu().numpy(), st.cpu().numpy()
        same = np.array_equal(st, out["status"]) and np.array_equal(lp, out["logp"], equal_nan=True)
        if not same:
            d = np.abs(lp - out["logp"]) / np.maximum(1.0, np.abs(lp))
            close = np.array_equal(st, out["status"]) and np.nanmax(d) <= 1e-10
            if not close:
                bad += 1
            if verbose:
                print("DIFF" if not close else "rounding-level", dict(n=n, ns=ns, nl=nl, k=k, p=p, nb=nb, solver=solver), float(np.nanmax(d)),
                      int((st != out["status"]).sum()))
    if verbose:
        print("trials done, mismatches:", bad)
    return bad


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 12)
