"""CPU, numpy only: the device's blocked Gauss-Jordan (model of gauss_jordan_blocked, as in tools/blocked_elimination_model.py) inside
cycle reduction on the draws tools/fuzz_cr.py flagged at the 1e-9 bar (seeds 11 and 23), with the refinement rules that were
considered: per iteration [it: pivot ratio / cond(A1), * = refined].  Output kept in profiles/r3/cr_refine_model.txt."""
import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from geconpy_amd import workloads as wl
import oracle
def blocked_solve(M, R, BS=8):
    W = np.hstack((M, R)).astype(np.float64); n = M.shape[0]
    used = np.zeros(n, bool); prow = np.zeros(n, int); nsteps = (n + BS - 1) // BS
    pmin, pmax = 1e300, 0.0
    for kb in range(nsteps):
        j0 = kb * BS; bw = min(BS, n - j0)
        pw = W[:, j0:j0 + bw].copy(); idm = np.zeros((n, bw)); rsel = []; inv_own = np.ones(n)
        for c in range(bw):
            r = int(np.argmax(np.where(~used, np.abs(pw[:, c]), -1.0))); used[r] = True; rsel.append(r); idm[r, c] = 1.0
            inv = 1.0 / pw[r, c]; pmin = min(pmin, abs(inv)); pmax = max(pmax, abs(inv))
            f = pw[:, c].copy(); f[r] = 0.0
            inv_own[r] = inv
            for c2 in range(bw):
                if c2 > c: pw[:, c2] -= f * (pw[r, c2] * inv)
                if c2 <= c: idm[:, c2] -= f * (idm[r, c2] * inv)
        idm *= inv_own[:, None]
        lh = -idm
        for a, r in enumerate(rsel): lh[r, a] += 1.0
        Y = W[rsel, :].copy()
        prow[j0:j0 + bw] = rsel
        W[:, j0 + bw:] -= lh @ Y[:, j0 + bw:]
    return W[prow, n:], pmax / pmin
def find(seed, tgt):
    rng = np.random.default_rng(seed)
    for trial in range(3000):
        n = int(rng.integers(3, 65)); ns = int(rng.integers(1, max(2, n // 2))); nl = int(rng.integers(1, max(2, n // 3))); nb = 4
        try: sysm = [wl.sw_shaped_system(int(rng.integers(1 << 30)), n=n, n_state=ns, n_lead=nl, k=1) for _ in range(nb)]
        except Exception: continue
        tol = float(rng.choice([1e-6, 1e-9, 1e-12]))
        if (n, ns, nl) == tgt[:3]: return sysm[tgt[3]][:3], tol
def cr(A, B, C, tol, solve, refine_rule):
    n = A.shape[0]
    A0, A1, A2, Ah = A.copy(), B.copy(), C.copy(), B.copy()
    info = []
    for it in range(200):
        R = np.hstack((A0, A2))
        X, ratio = solve(A1, R)
        ref = refine_rule(it, ratio)
        if ref:
            Xc, _ = solve(A1, R - A1 @ X); X = X + Xc
        info.append((it, ratio, np.linalg.cond(A1), ref))
        X0, X2 = X[:, :n], X[:, n:]
        m00, m02, m20, m22 = A0 @ X0, A0 @ X2, A2 @ X0, A2 @ X2
        A1 = A1 - m02 - m20; Ah = Ah - m20; A0 = -m00; A2 = -m22
        if np.abs(A0).sum(axis=0).max() < tol and np.abs(A2).sum(axis=0).max() < tol: break
    Xf, ratio = solve(Ah, A)
    reff = refine_rule(99, ratio)
    if reff:
        Xc, _ = solve(Ah, A - Ah @ Xf); Xf = Xf + Xc
    info.append((99, ratio, np.linalg.cond(Ah), reff))
    return -Xf, info
for seed, tgt, BS in ((23, (39, 15, 4, 3), 5), (23, (57, 23, 15, 0), 4), (23, (57, 23, 15, 0), 8), (11, (27, 9, 1, 0), 4)):
    (A, B, C), tol = find(seed, tgt)
    Tc, conv, itc = oracle.cycle_reduction_core(A, B, C, 200, tol)
    sol = lambda M, R: blocked_solve(M, R, BS)
    for name, rule in (("no refine", lambda it, r: False), ("it<2 & r>1e3", lambda it, r: it < 2 and r > 1e3), ("any it r>1e3", lambda it, r: r > 1e3), ("any it r>3e2", lambda it, r: r > 3e2), ("all", lambda it, r: True)):
        T, info = cr(A, B, C, tol, sol, rule)
        print(tgt, "BS", BS, f"{name:14s} |T - T_oracle| = {np.abs(T - Tc).max():.2e}", " ".join(f"[{i}:{r:.0f}/{c:.0e}{'*' if f else ''}]" for i, r, c, f in info))
