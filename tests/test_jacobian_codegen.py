"""theta -> A,B,C,D codegen.  CPU: the symbolic RBC twin against the numpy closed form, the
generated source, and that the cross-compiled library exports its C ABI.  GPU: the kernel's matrices
against the numpy closed form and the theta -> logp path against the matrix-fed pipeline."""
import ctypes

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
