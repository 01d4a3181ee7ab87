"""theta -> A,B,C,D codegen.  CPU: the symbolic RBC twin against the numpy closed form, the
generated source, and that the cross-compiled library exports its C ABI.  GPU: the kernel's matrices
against the numpy closed form and the theta -> logp path against the matrix-fed pipeline."""
import ctypes
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose

from geconpy_amd import workloads as wl
from geconpy_amd.jacobian_codegen import JacobianProgram, rbc_linearized_program


def _theta(nb, seed=3):
    th = wl.rbc_prior_draws(nb, seed=seed)
    names = ["sigma", "phi", "alpha", "beta", "delta", "rho_A", "sigma_A"]
    return th, np.ascontiguousarray(np.stack([th[k] for k in names], axis=1))


def test_symbolic_rbc_matches_numpy_closed_form():
    import sympy as sp

    prog = rbc_linearized_program()
    th, theta = _theta(16)
    A, B, C, D = wl.rbc_linearized_jacobians(**th)
    fns = [sp.lambdify(prog.params, M, "numpy") for M in prog.mats]
    for i in range(16):
        for F, ref in zip(fns, (A[i], B[i], C[i], D[i])):
            assert_allclose(np.asarray(F(*theta[i]), dtype=float), ref, rtol=1e-13, atol=0)
    # structure: exactly the entries the closed form fills
    nz = {(mi, flat) for mi, flat, _ in prog.nonzero_entries() if mi < 4}
    ref_nz = set()
    for mi, M in enumerate((A, B, C, D)):
        for flat in np.flatnonzero(np.any(M.reshape(16, -1) != 0, axis=0)):
            ref_nz.add((mi, int(flat)))
    assert nz == ref_nz


def test_generated_source_and_library():
    prog = rbc_linearized_program()
    src = prog.source()
    assert "__global__" in src and "dsge_jac_launch" in src and "JAC_NPAR 7" in src
    assert src == rbc_linearized_program().source()  # deterministic (the library is cached by source hash)
    prog.build()  # hipcc cross-compiles without a GPU
    lib = prog.load()  # (imports torch first: one HIP runtime per process, as geconpy_amd._lib.load does)
    dims = [ctypes.c_int() for _ in range(4)]
    assert lib.dsge_jac_dims(*[ctypes.byref(d) for d in dims]) == 0
    assert [d.value for d in dims] == [8, 1, 7, 1]
    assert hasattr(lib, "dsge_jac_launch") and hasattr(lib, "dsge_jac_vjp_launch")


def test_program_validation():
    import sympy as sp

    a, b = sp.symbols("a b")
    with pytest.raises(ValueError):
        JacobianProgram("bad", [a], sp.Matrix([[a * b]]), sp.eye(1), sp.zeros(1, 1), sp.ones(1, 1))  # b is no parameter
    with pytest.raises(ValueError):
        JacobianProgram("bad", [a], sp.eye(2), sp.eye(1), sp.zeros(1, 1), sp.ones(1, 1))  # shape mismatch


@pytest.mark.gpu
def test_kernel_matches_closed_form_and_feeds_the_pipeline():
    import torch

    import oracle
    from geconpy_amd.engine import LogpEngine

    nb = 1000  # not a multiple of the 256-thread block
    prog = rbc_linearized_program()
    th, theta = _theta(nb)
    eng = LogpEngine(0)
    d_theta = eng.to_device(theta)
    A, B, C, D, q = eng.jacobians_from_theta(prog, d_theta)
    torch.cuda.synchronize()
    refs = wl.rbc_linearized_jacobians(**th)
    for got, ref in zip((A, B, C, D), refs):
        got = got.cpu().numpy()
        assert np.array_equal(got == 0, ref == 0)  # same sparsity, exact zeros
        assert_allclose(got, ref, rtol=1e-13, atol=0)
    assert_allclose(q.cpu().numpy()[:, 0], th["sigma_A"] ** 2, rtol=1e-15)
    # theta -> logp on the device vs the pipeline fed with host-built matrices
    Z = np.zeros((1, 8))
    Z[0, wl.RBC_VARIABLES.index("Y")] = 1.0
    y = np.random.default_rng(0).normal(0, 0.05, (100, 1))
    dZ, dy = eng.to_device(Z), eng.to_device(y)
    logp, st = eng.logp_from_theta(prog, d_theta, dZ, dy, tol=1e-8, max_iter=1000)
    dev = [eng.to_device(x) for x in refs]
    logp2, st2 = eng.solve_kalman_logp(*dev, eng.to_device((th["sigma_A"] ** 2)[:, None]), dZ, dy, q_mode=1, tol=1e-8,
                                       max_iter=1000)
    torch.cuda.synchronize()
    assert torch.equal(st, st2) and int((st != 0).sum()) == 0
    assert_allclose(logp.cpu().numpy(), logp2.cpu().numpy(), rtol=1e-11)
    for i in (0, 499, 999):
        ref = oracle.solve_kalman_logp(refs[0][i], refs[1][i], refs[2][i], refs[3][i], np.array([[th["sigma_A"][i] ** 2]]), Z, y)
        assert_allclose(logp[i].item(), ref["logp"], rtol=1e-9)


@pytest.mark.gpu
def test_theta_gradient_end_to_end():
    """theta -> (logp, d logp / d theta) on the device (generated Jacobian kernel, logp + reverse-mode pipeline,
    generated pullback kernel) against central finite differences of the CPU oracle in theta."""
    import torch

    import oracle
    from geconpy_amd.engine import LogpEngine

    nb = 5
    prog = rbc_linearized_program()
    th, theta = _theta(nb, seed=9)
    names = ["sigma", "phi", "alpha", "beta", "delta", "rho_A", "sigma_A"]
    Z = np.zeros((2, 8))
    Z[0, wl.RBC_VARIABLES.index("Y")] = 1.0
    Z[1, wl.RBC_VARIABLES.index("I")] = 1.0
    y = np.random.default_rng(5).normal(0, 0.05, (60, 2))
    h = np.array([1e-4, 1e-3])
    eng = LogpEngine(0)
    logp, st, theta_bar, g = eng.logp_and_grad_from_theta(prog, eng.to_device(theta), eng.to_device(Z), eng.to_device(y),
                                                          Hdiag=eng.to_device(h), tol=1e-13, max_iter=200, n_filter_hint=4)
    torch.cuda.synchronize()
    assert int((st != 0).sum()) == 0
    theta_bar = theta_bar.cpu().numpy()

    def f(row):
        kw = dict(zip(names, row))
        A, B, C, D = wl.rbc_linearized_jacobians(**kw)
        return oracle.solve_kalman_logp(A, B, C, D, np.array([[kw["sigma_A"] ** 2]]), Z, y, H=np.diag(h), tol=1e-13,
                                        max_iter=200)["logp"]

    for i in range(nb):
        assert_allclose(logp[i].item(), f(theta[i]), rtol=1e-9)
        for j in range(7):
            e = 1e-6 * max(abs(theta[i, j]), 1e-2)
            tp, tm = theta[i].copy(), theta[i].copy()
            tp[j] += e
            tm[j] -= e
            fd = (f(tp) - f(tm)) / (2 * e)
            assert_allclose(theta_bar[i, j], fd, rtol=5e-5, atol=1e-5 * max(1.0, np.abs(theta_bar[i]).max()))


@pytest.mark.gpu
def test_sw_shaped_theta_program_on_device():
    """The 37-parameter affine family of the SW-shaped workload (bench.py --from-theta on BASELINE configs[2]): the generated
    kernel reproduces the host twin bit for bit on the entries it writes (one multiply-add each), keeps the zero-column
    structure of the base system, and theta -> logp through the fused call matches the oracle on the host twin."""
    import torch

    import oracle
    from geconpy_amd import workloads as wl
    from geconpy_amd.engine import LogpEngine
    from geconpy_amd.jacobian_codegen import sw_shaped_program

    prog = sw_shaped_program()
    assert (prog.n, prog.k, len(prog.params)) == (40, 7, 37)
    nb = 96
    th = wl.sw_theta_draws(nb)
    A, B, C, D, q = wl.sw_theta_jacobians(th)
    eng = LogpEngine(0)
    d_th = eng.to_device(th)
    dA, dB, dC, dD, dq = eng.jacobians_from_theta(prog, d_th)
    torch.cuda.synchronize()
    for got, want in ((dA, A), (dB, B), (dC, C), (dD, D), (dq, q)):
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=4e-16, atol=0)
    assert np.array_equal(dA.cpu().numpy() != 0, A != 0) and np.array_equal(dC.cpu().numpy() != 0, C != 0)
    om = wl.sw_shaped_observation_model()
    dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
    ns, zs = eng.structure_hints(dA, dZ)
    assert (ns, zs, eng.static_hint(dA, dC)) == (18, 1, 10)
    logp, status = eng.logp_from_theta(prog, d_th, dZ, dy, Hdiag=dH, tol=1e-8, max_iter=1000, n_state_hint=ns, z_selector_hint=zs,
                                       options={"n_static_hint": 10})
    torch.cuda.synchronize()
    logp, status = logp.cpu().numpy(), status.cpu().numpy()
    assert not status.any()
    for i in (0, 31, 95):
        ref = oracle.solve_kalman_logp(A[i], B[i], C[i], D[i], np.diag(q[i]), om["Z"], om["y"], H=np.diag(om["Hdiag"]), tol=1e-8,
                                       max_iter=1000)
        assert abs(logp[i] - ref["logp"]) <= 1e-8 * abs(ref["logp"])
    # d logp / d theta through the generated pullback kernel against central differences of the fused call
    lp, st, tb, _g = eng.logp_and_grad_from_theta(prog, d_th, dZ, dy, Hdiag=dH, tol=1e-10, max_iter=1000)
    torch.cuda.synchronize()
    assert not st.cpu().numpy().any()
    tb = tb.cpu().numpy()
    for j in (0, 17, 20, 33):
        h = 1e-2 if j < 30 else 1e-6  # (a column scaling moves by 0.02 h; sigma ~ 1e-2)
        tp, tm = th.copy(), th.copy()
        tp[:, j] += h
        tm[:, j] -= h
        lps = []
        for tt in (tp, tm):
            l_, _s = eng.logp_from_theta(prog, eng.to_device(tt), dZ, dy, Hdiag=dH, tol=1e-10, max_iter=1000)
            torch.cuda.synchronize()
            lps.append(l_.cpu().numpy())
        fd = (lps[0] - lps[1]) / (2 * h)
        np.testing.assert_allclose(tb[:, j], fd, rtol=2e-3, atol=2e-3 * np.abs(fd).max())


def test_sample_parameters_methods():
    from scipy import stats

    from geconpy_amd.diagnostics import sample_parameters

    priors = {"beta": stats.beta(10, 1), "sigma": stats.gamma(4, scale=0.5), "rho": (0.0, 0.99)}
    for method in ("random", "lhs", "sobol", "halton", "sobol_ppf", "lhs_ppf"):
        names, x = sample_parameters(priors, 128, seed=3, method=method)
        assert names == ["beta", "sigma", "rho"] and x.shape == (128, 3) and np.isfinite(x).all()
        assert (x[:, 0] > 0).all() and (x[:, 0] <= 1).all() and (x[:, 2] >= 0).all() and (x[:, 2] <= 0.99).all()
        names2, x2 = sample_parameters(priors, 128, seed=3, method=method)
        assert np.array_equal(x, x2)  # seeded
    _n, x = sample_parameters(priors, 256, seed=0, method="lhs", hdi_prob=0.9)
    lo, hi = stats.gamma(4, scale=0.5).ppf([0.05, 0.95])
    assert x[:, 1].min() >= lo and x[:, 1].max() <= hi
    with pytest.raises(ValueError):
        sample_parameters(priors, 8, method="nope")


@pytest.mark.gpu
def test_prior_solvability_check_batched_rbc():
    """prior_solvability_check (perturbation_diagnostics.py:526-579) on the generated RBC kernel: draws inside the priors'
    bulk all solve; widening delta / rho_A far outside produces labelled failures, in input order."""
    from scipy import stats

    from geconpy_amd.diagnostics import prior_solvability_check_batched

    prog = rbc_linearized_program()
    priors = {"sigma": stats.gamma(8, scale=0.25), "phi": stats.gamma(9, scale=1 / 3), "alpha": stats.beta(5, 9),
              "beta": stats.beta(40, 1), "delta": stats.beta(2, 40), "rho_A": stats.beta(3, 2)}
    df = prior_solvability_check_batched(prog, 512, priors, seed=1, method="sobol", defaults={"sigma_A": 0.01}, tol=1e-8,
                                         max_iter=1000)
    assert list(df.columns) == list(priors) + ["failure_step", "norm_deterministic", "norm_stochastic"] and len(df) == 512
    assert df["failure_step"].isna().all() or (df["failure_step"].isna().mean() > 0.98)
    ok = df["failure_step"].isna()
    assert (df.loc[ok, "norm_deterministic"] < 1e-8).all() and (df.loc[ok, "norm_stochastic"] < 1e-8).all()
    wild = dict(priors, rho_A=(0.5, 1.5), alpha=(-0.2, 0.9))
    dfw = prior_solvability_check_batched(prog, 512, wild, seed=1, method="lhs", defaults={"sigma_A": 0.01}, tol=1e-8,
                                          max_iter=200)
    labels = set(dfw["failure_step"].dropna())
    assert labels and labels <= {"steady_state", "perturbation", "blanchard-kahn", "deterministic_norm", "stochastic_norm"}
    explosive = dfw["rho_A"] > 1.0
    assert dfw.loc[explosive & (dfw["alpha"] > 0.05), "failure_step"].notna().all()  # an explosive shock process never passes
    assert dfw.loc[dfw["alpha"] < 0, "failure_step"].eq("steady_state").all()        # (alpha / R)^(...) of a negative number


def _rbc_obs_program():
    """RBC with a parameter-dependent observation equation: log-level output and consumption,
    y_obs = ln X_ss(theta) + x_hat  (Z rows select Y and C with unit loadings scaled by a parameter-dependent factor to
    exercise Z(theta) as well: the second row loads 1 / sigma on C)."""
    import sympy as sp

    base = rbc_linearized_program()
    sigma, phi, alpha, beta, delta, rho_A, sigma_A = base.params
    R = 1 / beta - (1 - delta)
    W = (1 - alpha) ** (1 / (1 - alpha)) * (alpha / R) ** (alpha / (1 - alpha))
    Y = (R / (R - delta * alpha)) ** (sigma / (sigma + phi)) * ((1 - alpha) ** (-phi) * W ** (1 + phi)) ** (1 / (sigma + phi))
    Cs = Y - delta * alpha * Y / R
    Z = sp.zeros(2, 8)
    Z[0, 7] = 1
    Z[1, 1] = 1 / sigma
    return JacobianProgram("rbc_obs", base.params, *base.mats, q=base.q, Z=Z, d=[sp.log(Y), sp.log(Cs)])


def test_observation_equation_codegen_source():
    prog = _rbc_obs_program()
    src = prog.source()
    assert "jac_obs_kernel" in src and "dsge_jac_obs_launch" in src and "dsge_jac_obs_vjp_launch" in src and prog.p == 2
    assert "jac_obs_kernel" not in rbc_linearized_program().source()  # programs without Z / d are unchanged


@pytest.mark.gpu
def test_observation_equation_from_theta_on_device():
    """Z(theta), d(theta) generated on the device (statespace.py:298-388) and used by the fused call as batched design
    matrix / intercept: values against the numpy closed form, logp against the oracle with that draw's Z and d."""
    import torch

    import oracle
    from geconpy_amd.engine import LogpEngine

    prog = _rbc_obs_program()
    nb = 48
    th, theta = _theta(nb)
    eng = LogpEngine(0)
    d_th = eng.to_device(theta)
    Zb, db = eng.observation_from_theta(prog, d_th)
    torch.cuda.synchronize()
    ss = wl.rbc_steady_state(th["sigma"], th["phi"], th["alpha"], th["beta"], th["delta"])
    assert_allclose(db.cpu().numpy(), np.stack([np.log(ss["Y"]), np.log(ss["C"])], axis=1), rtol=1e-11, atol=1e-13)
    Zh = Zb.cpu().numpy()
    assert_allclose(Zh[:, 1, 1], 1.0 / th["sigma"], rtol=1e-15)
    assert np.all(Zh[:, 0, 7] == 1.0) and np.count_nonzero(Zh) == 2 * nb
    y = np.stack([np.log(ss["Y"][0]), np.log(ss["C"][0])]) + np.random.default_rng(0).normal(0, 0.02, (60, 2))
    dy, dH = eng.to_device(y), eng.to_device(np.array([1e-4, 1e-4]))
    logp, status = eng.logp_from_theta(prog, d_th, None, dy, Hdiag=dH, tol=1e-10, max_iter=1000)
    torch.cuda.synchronize()
    assert not status.cpu().numpy().any()
    A, B, C, D = wl.rbc_linearized_jacobians(**th)
    for i in (0, 20, 47):
        ref = oracle.solve_kalman_logp(A[i], B[i], C[i], D[i], np.array([[th["sigma_A"][i] ** 2]]), Zh[i], y,
                                       H=np.diag([1e-4, 1e-4]), d=db.cpu().numpy()[i], tol=1e-10, max_iter=1000)
        assert abs(logp[i].item() - ref["logp"]) <= 1e-8 * abs(ref["logp"])


@pytest.mark.gpu
def test_theta_gradient_through_a_parameter_dependent_observation_equation():
    """d logp / d theta when BOTH the design matrix and the intercept depend on theta (statespace.py:298-388: observation
    equations are differentiated by pytensor upstream): generated Z(theta), d(theta) kernels -> dense-Z gradient entry point
    (Z_bar) -> generated pullbacks of Z and d, against central finite differences of the oracle in theta."""
    import torch

    import oracle
    from geconpy_amd.engine import LogpEngine

    prog = _rbc_obs_program()
    assert "dsge_jac_obs_z_vjp_launch" in prog.source()
    nb = 4
    th, theta = _theta(nb, seed=11)
    names = ["sigma", "phi", "alpha", "beta", "delta", "rho_A", "sigma_A"]
    ss0 = wl.rbc_steady_state(th["sigma"], th["phi"], th["alpha"], th["beta"], th["delta"])
    y = np.stack([np.log(ss0["Y"][0]), np.log(ss0["C"][0])]) + np.random.default_rng(2).normal(0, 0.02, (50, 2))
    h = np.array([1e-4, 2e-4])
    eng = LogpEngine(0)
    logp, st, theta_bar, g = eng.logp_and_grad_from_theta(prog, eng.to_device(theta), None, eng.to_device(y), Hdiag=eng.to_device(h),
                                                          tol=1e-13, max_iter=300, n_filter_hint=2)
    torch.cuda.synchronize()
    assert int((st != 0).sum()) == 0 and "Z_bar" in g
    theta_bar = theta_bar.cpu().numpy()

    def f(row):
        kw = dict(zip(names, row))
        A, B, C, D = wl.rbc_linearized_jacobians(**kw)
        ss = wl.rbc_steady_state(kw["sigma"], kw["phi"], kw["alpha"], kw["beta"], kw["delta"])
        Z = np.zeros((2, 8))
        Z[0, 7] = 1.0
        Z[1, 1] = 1.0 / kw["sigma"]
        d = np.array([np.log(ss["Y"]), np.log(ss["C"])]).reshape(2)
        return oracle.solve_kalman_logp(A, B, C, D, np.array([[kw["sigma_A"] ** 2]]), Z, y, H=np.diag(h), d=d, tol=1e-13,
                                        max_iter=300)["logp"]

    for i in range(nb):
        assert_allclose(logp[i].item(), f(theta[i]), rtol=1e-9)
        for j in range(7):
            e = 1e-6 * max(abs(theta[i, j]), 1e-2)
            tp, tm = theta[i].copy(), theta[i].copy()
            tp[j] += e
            tm[j] -= e
            fd = (f(tp) - f(tm)) / (2 * e)
            assert_allclose(theta_bar[i, j], fd, rtol=5e-5, atol=1e-5 * max(1.0, np.abs(theta_bar[i]).max()))


def test_parameter_names_never_meet_the_kernel_identifiers():
    """A parameter called theta / A / q / draw / x0, or one that is no C identifier, is printed as par<i>."""
    import sympy as sp

    from geconpy_amd.jacobian_codegen import JacobianProgram

    names = ["theta", "A", "q", "draw", "x0", "rho^A", "sigma.e", "par1"]
    ps = [sp.Symbol(nm, positive=True) for nm in names]
    A, B, C, D = sp.zeros(2, 2), sp.zeros(2, 2), sp.zeros(2, 2), sp.zeros(2, 1)
    A[0, 0] = ps[0] * ps[1] + sp.exp(ps[4])
    B[1, 1] = ps[2] ** sp.Rational(1, 3) / ps[3]
    C[1, 0] = ps[5] * ps[6] / ps[7]
    D[0, 0] = ps[2]
    prog = JacobianProgram("hostile_names", ps, A, B, C, D, q=[ps[2] ** 2], Z=sp.Matrix([[ps[0], 0]]), d=[sp.log(ps[1])])
    src = prog.source()
    for i in range(len(names)):
        assert f"const double par{i} = th[{i}];" in src
    assert "const double theta =" not in src and "const double A =" not in src and "rho^A" not in src
    assert os.path.exists(prog.build())  # hipcc accepts it (cross-compiles without a GPU)


def test_integers_beyond_32_bits_are_double_literals():
    """sympy folds rational coefficients into integers of any size; clang rejects a literal beyond 64 bits (found by
    tools/fuzz_theta.py).  Small integers stay integers (pow(x, 2) keeps its integer exponent)."""
    import sympy as sp

    from geconpy_amd.jacobian_codegen import JacobianProgram

    a = sp.Symbol("a", positive=True)
    M = sp.Matrix([[339799298607853600768 * a / (68719476736 * a + 424010647921) ** 2 + sp.exp(a) ** 2]])
    src = JacobianProgram("bigint", [a], M, M, M, M).source()
    assert "339799298607853600768.0" in src and "68719476736.0" in src and "424010647921.0" in src
    assert ", 2)" in src and ", 2.0)" not in src
