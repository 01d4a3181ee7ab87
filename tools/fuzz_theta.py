"""GPU box: random parameter programs through geconpy_amd.jacobian_codegen -- the generated kernels theta -> A, B, C, D, q,
theta -> Z, d and both pullbacks against sympy's own lambdify of the same expressions (and of their derivatives).
Parameter names include the kernel's own identifiers (theta, A, q, draw, x0, ...) and names that are not C identifiers."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

HOSTILE = ["theta", "A", "B", "C", "D", "q", "draw", "batch", "th", "tb", "x0", "z1", "w0", "v2", "g0_1", "gd_0", "Ad", "qb",
           "par1", "lambda", "rho^A", "sigma.e", "beta", "gamma", "Z", "d", "d_bar", "theta_bar", "alpha", "delta"]


def random_expr(rng, ps, depth=0):
    """An expression of positive parameters that stays finite and smooth on (0.3, 1.8)^npar."""
    import sympy as sp

    r = rng.random()
    if depth >= 3 or r < 0.25:
        p = ps[rng.integers(len(ps))]
        return p if rng.random() < 0.7 else sp.Rational(int(rng.integers(1, 9)), int(rng.integers(1, 9)))
    a = random_expr(rng, ps, depth + 1)
    b = random_expr(rng, ps, depth + 1)
    kind = rng.integers(9)
    if kind == 0:
        return a + b
    if kind == 1:
        return a * b
    if kind == 2:
        return a / (1 + b * b)
    if kind == 3:
        return sp.exp(-a * a / 4)
    if kind == 4:
        return sp.log(1 + a * a)
    if kind == 5:
        return sp.sqrt(1 + a * a)
    if kind == 6:
        return (1 + a * a) ** sp.Rational(int(rng.integers(1, 5)), int(rng.integers(2, 6)))
    if kind == 7:
        return a - b
    return a ** int(rng.integers(2, 4))


def run(seed, trials, verbose=True):
    import sympy as sp
    import torch

    from geconpy_amd.engine import LogpEngine
    from geconpy_amd.jacobian_codegen import JacobianProgram

    rng = np.random.default_rng(seed)
    eng = LogpEngine(0)
    bad = 0
    for trial in range(trials):
        n, k, npar = int(rng.integers(2, 8)), int(rng.integers(1, 4)), int(rng.integers(1, 7))
        p_obs = int(rng.integers(1, 4))
        names = list(rng.choice(HOSTILE, size=npar, replace=False))
        ps = [sp.Symbol(nm, positive=True) for nm in names]

        def mat(rows, cols, fill):
            M = sp.zeros(rows, cols)
            for r in range(rows):
                for c in range(cols):
                    if rng.random() < fill:
                        M[r, c] = random_expr(rng, ps)
            return M

        A, B, Cm, D = mat(n, n, 0.3), mat(n, n, 0.6), mat(n, n, 0.3), mat(n, k, 0.5)
        with_q, with_Z, with_d = rng.random() < 0.8, rng.random() < 0.6, rng.random() < 0.6
        q = [1 + random_expr(rng, ps) ** 2 for _ in range(k)] if with_q else None
        Z = mat(p_obs, n, 0.4) if with_Z else None
        d = [random_expr(rng, ps) for _ in range(p_obs)] if with_d else None
        prog = JacobianProgram(f"fuzz{seed}_{trial}", ps, A, B, Cm, D, q=q, Z=Z, d=d)
        nb = int(rng.choice([1, 3, 257, 700]))
        theta = rng.uniform(0.3, 1.8, (nb, npar))
        d_theta = eng.to_device(theta)
        got = eng.jacobians_from_theta(prog, d_theta)
        torch.cuda.synchronize()
        errs = []
        cols = [theta[:, i] for i in range(npar)]
        mats = [A, B, Cm, D] + ([sp.Matrix([q])] if with_q else [])
        for mi, M in enumerate(mats):
            g = got[mi].cpu().numpy().reshape(nb, -1)
            flat = list(M)
            for e_i, e in enumerate(flat):
                ref = np.broadcast_to(np.asarray(sp.lambdify(ps, e, "numpy")(*cols), dtype=float), (nb,))
                if e == 0:
                    errs.append(0.0 if np.all(g[:, e_i] == 0.0) else 1.0)  # structural zeros are exact zeros
                else:
                    errs.append(float(np.max(np.abs(g[:, e_i] - ref) / (1.0 + np.abs(ref)))))
        # pullback: random cotangents against the lambdified derivative
        bars = [rng.normal(size=(nb,) + tuple(M.shape)) for M in (A, B, Cm, D)] + [rng.normal(size=(nb, k))]
        d_bars = [eng.to_device(b) for b in bars]
        tb = torch.zeros((nb, npar), dtype=torch.float64, device=eng.device)
        prog.launch_vjp(d_theta.data_ptr(), nb, *[b.data_ptr() for b in d_bars[:4]], d_bars[4].data_ptr() if with_q else None,
                        tb.data_ptr(), eng._stream())
        torch.cuda.synchronize()
        ref_tb = np.zeros((nb, npar))
        for mi, M in enumerate(mats):
            bflat = bars[mi].reshape(nb, -1)
            for e_i, e in enumerate(list(M)):
                if e == 0:
                    continue
                for pi, p_ in enumerate(ps):
                    de = sp.diff(e, p_)
                    if de != 0:
                        ref_tb[:, pi] += bflat[:, e_i] * np.broadcast_to(np.asarray(sp.lambdify(ps, de, "numpy")(*cols), dtype=float), (nb,))
        scale = 1e-300 + np.max(np.abs(ref_tb)) if ref_tb.size else 1.0
        errs.append(float(np.max(np.abs(tb.cpu().numpy() - ref_tb)) / scale) * 1e-2)  # sums of many terms: 1e-11 relative to the largest
        if with_Z or with_d:
            Zg, dg = eng.observation_from_theta(prog, d_theta)
            torch.cuda.synchronize()
            if with_Z:
                g = Zg.cpu().numpy().reshape(nb, -1)
                for e_i, e in enumerate(list(Z)):
                    ref = np.broadcast_to(np.asarray(sp.lambdify(ps, e, "numpy")(*cols), dtype=float), (nb,))
                    errs.append(float(np.max(np.abs(g[:, e_i] - ref) / (1.0 + np.abs(ref)))) if e != 0 else float(np.any(g[:, e_i] != 0)))
            if with_Z:  # the pullback of Z (round 3: dsge_jac_obs_z_vjp_launch) ACCUMULATES into theta_bar
                zbar = rng.normal(size=(nb, p_obs, n))
                tb3 = torch.full((nb, npar), -0.25, dtype=torch.float64, device=eng.device)
                prog.launch_obs_z_vjp(d_theta.data_ptr(), nb, eng.to_device(zbar).data_ptr(), tb3.data_ptr(), eng._stream())
                torch.cuda.synchronize()
                ref3 = np.full((nb, npar), -0.25)
                zflat = zbar.reshape(nb, -1)
                for e_i, e in enumerate(list(Z)):
                    if e == 0:
                        continue
                    for pi, p_ in enumerate(ps):
                        de = sp.diff(e, p_)
                        if de != 0:
                            ref3[:, pi] += zflat[:, e_i] * np.broadcast_to(np.asarray(sp.lambdify(ps, de, "numpy")(*cols), dtype=float), (nb,))
                errs.append(float(np.max(np.abs(tb3.cpu().numpy() - ref3)) / np.max(np.abs(ref3))) * 1e-2)
            if with_d:
                g = dg.cpu().numpy()
                for e_i, e in enumerate(d):
                    ref = np.broadcast_to(np.asarray(sp.lambdify(ps, e, "numpy")(*cols), dtype=float), (nb,))
                    errs.append(float(np.max(np.abs(g[:, e_i] - ref) / (1.0 + np.abs(ref)))))
                dbar = rng.normal(size=(nb, p_obs))
                tb2 = torch.full((nb, npar), 0.5, dtype=torch.float64, device=eng.device)  # the obs pullback ACCUMULATES
                prog.launch_obs_vjp(d_theta.data_ptr(), nb, eng.to_device(dbar).data_ptr(), tb2.data_ptr(), eng._stream())
                torch.cuda.synchronize()
                ref2 = np.full((nb, npar), 0.5)
                for e_i, e in enumerate(d):
                    for pi, p_ in enumerate(ps):
                        de = sp.diff(e, p_)
                        if de != 0:
                            ref2[:, pi] += dbar[:, e_i] * np.broadcast_to(np.asarray(sp.lambdify(ps, de, "numpy")(*cols), dtype=float), (nb,))
                errs.append(float(np.max(np.abs(tb2.cpu().numpy() - ref2)) / np.max(np.abs(ref2))) * 1e-2)
        worst = max(errs) if errs else 0.0
        ok = worst < 1e-12
        bad += (not ok)
        if verbose or not ok:
            print(f"trial {trial}: n={n} k={k} npar={npar} names={names} nb={nb} q={with_q} Z={with_Z} d={with_d} worst={worst:.2e}"
                  f"{'' if ok else '  MISMATCH'}", flush=True)
    return bad


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    print("mismatches:", run(seed, trials))
