"""GPU: reverse-mode gradient of logp against central finite differences of the CPU oracle.

The oracle (numpy/scipy: cycle reduction -> R -> bilinear Lyapunov -> Joseph-form filter) shares no code with the
device path (structure-reduced filter, doubling Lyapunov, hand-written reverse sweep), so agreement of directional
derivatives in random directions checks the whole chain A,B,C,D,q,d,H -> logp."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import oracle
from geconpy_amd import _lib, batched
from geconpy_amd import workloads as wl

pytestmark = pytest.mark.gpu


def _oracle_logp(A, B, C, D, q, Z, y, d, h):
    return oracle.solve_kalman_logp(A, B, C, D, np.diag(q), Z, y, H=np.diag(h), d=d, tol=1e-13, max_iter=200)["logp"]


def _directional_check(A, B, C, D, q, Z, y, d, h, g, rng, n_dirs=3, eps=1e-6, rtol=2e-5):
    """<grad, direction> vs the central difference of the oracle along random directions that respect the
    structural zeros of A (the contract of the gradient entry point)."""
    maskA = (A != 0).any(axis=0)[None, :] * np.ones_like(A)
    for _ in range(n_dirs):
        dA = rng.standard_normal(A.shape) * maskA * 0.1
        dB, dC, dD = (rng.standard_normal(M.shape) * 0.1 for M in (B, C, D))
        dq = rng.standard_normal(q.shape) * q * 0.3
        dd = rng.standard_normal(d.shape) * 0.1
        dh = rng.standard_normal(h.shape) * h * 0.3
        analytic = ((g["A_bar"] * dA).sum() + (g["B_bar"] * dB).sum() + (g["C_bar"] * dC).sum() + (g["D_bar"] * dD).sum()
                    + (g["q_bar"] * dq).sum() + (g["d_bar"] * dd).sum() + (g["h_bar"] * dh).sum())
        fp = _oracle_logp(A + eps * dA, B + eps * dB, C + eps * dC, D + eps * dD, q + eps * dq, Z, y, d + eps * dd, h + eps * dh)
        fm = _oracle_logp(A - eps * dA, B - eps * dB, C - eps * dC, D - eps * dD, q - eps * dq, Z, y, d - eps * dd, h - eps * dh)
        fd = (fp - fm) / (2 * eps)
        assert_allclose(analytic, fd, rtol=rtol, atol=1e-6 * max(1.0, abs(fd)))
    # each block on its own (a cancellation between blocks would hide an error in one of them)
    for name, M in (("A_bar", A), ("B_bar", B), ("C_bar", C), ("D_bar", D)):
        dM = rng.standard_normal(M.shape) * 0.1
        if name == "A_bar":
            dM = dM * maskA
        args = dict(A=A, B=B, C=C, D=D)
        key = name[0]
        fp = _oracle_logp(*[args[x] + (eps * dM if x == key else 0) for x in "ABCD"], q, Z, y, d, h)
        fm = _oracle_logp(*[args[x] - (eps * dM if x == key else 0) for x in "ABCD"], q, Z, y, d, h)
        fd = (fp - fm) / (2 * eps)
        assert_allclose((g[name] * dM).sum(), fd, rtol=rtol, atol=1e-6 * max(1.0, abs(fd)))


def test_gradient_rbc():
    rng = np.random.default_rng(0)
    nb = 6
    th = wl.rbc_prior_draws(nb, seed=4)
    A, B, C, D = wl.rbc_linearized_jacobians(**th)
    q = (th["sigma_A"] ** 2)[:, None]
    Z = np.zeros((2, 8))
    Z[0, wl.RBC_VARIABLES.index("Y")] = 1.0
    Z[1, wl.RBC_VARIABLES.index("C")] = 0.5
    y = rng.normal(0, 0.05, (40, 2))
    y[7, 0] = np.nan
    y[11, :] = np.nan
    d = np.array([0.01, -0.02])
    h = np.array([1e-4, 2e-4])
    out = batched.solve_kalman_logp_grad_batched(A, B, C, D, q, Z, y, d=d, Hdiag=h, tol=1e-13, max_iter=200)
    assert np.all(out["status"] == 0)
    ref = batched.solve_kalman_logp_batched(A, B, C, D, q, Z, y, d=d, Hdiag=h, tol=1e-13, max_iter=200, q_mode="diag_batched")
    assert_allclose(out["logp"], ref["logp"], rtol=1e-11)
    for i in range(nb):
        g = {k_: v[i] for k_, v in out.items() if k_.endswith("_bar")}
        assert_allclose(out["logp"][i], _oracle_logp(A[i], B[i], C[i], D[i], q[i], Z, y, d, h), rtol=1e-9)
        _directional_check(A[i], B[i], C[i], D[i], q[i], Z, y, d, h, g, rng)


def test_gradient_rbc_against_extrapolated_differences():
    """A check of the reverse sweep three orders tighter than the 2e-5 of the plain central differences: the directional
    derivative of the oracle by Richardson extrapolation of central differences (steps h, h/2, h/4 -> error O(h^6); with
    the oracle's own noise of ~1e-13 relative in logp the extrapolated value is good to ~1e-9 relative), on the RBC model
    with missing observations, every input block moving at once."""
    rng = np.random.default_rng(5)
    nb = 3
    th = wl.rbc_prior_draws(nb, seed=9)
    A, B, C, D = wl.rbc_linearized_jacobians(**th)
    q = (th["sigma_A"] ** 2)[:, None]
    Z = np.zeros((2, 8))
    Z[0, wl.RBC_VARIABLES.index("Y")] = 1.0
    Z[1, wl.RBC_VARIABLES.index("C")] = 0.5
    y = rng.normal(0, 0.05, (40, 2))
    y[5, 1] = np.nan
    d = np.array([0.01, -0.02])
    h = np.array([1e-4, 2e-4])
    out = batched.solve_kalman_logp_grad_batched(A, B, C, D, q, Z, y, d=d, Hdiag=h, tol=1e-14, max_iter=200)
    assert np.all(out["status"] == 0)

    def central(f, step):
        return (f(step) - f(-step)) / (2.0 * step)

    for i in range(nb):
        g = {k_: v[i] for k_, v in out.items() if k_.endswith("_bar")}
        maskA = (A[i] != 0).any(axis=0)[None, :] * np.ones_like(A[i])
        for _ in range(2):
            dA = rng.standard_normal(A[i].shape) * maskA * 0.1
            dB, dC, dD = (rng.standard_normal(M.shape) * 0.1 for M in (B[i], C[i], D[i]))
            dq = rng.standard_normal(q[i].shape) * q[i] * 0.3
            dd = rng.standard_normal(d.shape) * 0.1
            dh = rng.standard_normal(h.shape) * h * 0.3
            analytic = ((g["A_bar"] * dA).sum() + (g["B_bar"] * dB).sum() + (g["C_bar"] * dC).sum() + (g["D_bar"] * dD).sum()
                        + (g["q_bar"] * dq).sum() + (g["d_bar"] * dd).sum() + (g["h_bar"] * dh).sum())

            def f(e):
                return oracle.solve_kalman_logp(A[i] + e * dA, B[i] + e * dB, C[i] + e * dC, D[i] + e * dD, np.diag(q[i] + e * dq),
                                                Z, y, H=np.diag(h + e * dh), d=d + e * dd, tol=1e-15, max_iter=300)["logp"]

            h0 = 8e-3
            d1 = [central(f, h0 / 2 ** j) for j in range(3)]
            d2 = [(4.0 * d1[j + 1] - d1[j]) / 3.0 for j in range(2)]
            d3 = (16.0 * d2[1] - d2[0]) / 15.0
            assert abs(d3 - d2[1]) <= 1e-6 * abs(d3)  # the extrapolation has converged (else the step is too large)
            assert_allclose(analytic, d3, rtol=2e-8, atol=1e-9 * max(1.0, abs(d3)))


def test_gradient_sw_shaped():
    rng = np.random.default_rng(1)
    nb = 3
    b = wl.sw_shaped_batch(nb)
    om = wl.sw_shaped_observation_model()
    q = b["sigma"] ** 2
    y = om["y"][:50].copy()
    y[5, 2] = np.nan
    d = rng.normal(0, 0.01, 7)
    h = om["Hdiag"].copy()
    out = batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], y, d=d, Hdiag=h, tol=1e-13,
                                                 max_iter=200)
    assert np.all(out["status"] == 0)
    # structurally zero columns of A carry no cotangent by contract; T_bar only lives on the state columns
    zero_cols = ~(b["A"][0] != 0).any(axis=0)
    assert zero_cols.sum() == 22
    for i in range(nb):
        g = {k_: v[i] for k_, v in out.items() if k_.endswith("_bar")}
        assert_allclose(out["logp"][i], _oracle_logp(b["A"][i], b["B"][i], b["C"][i], b["D"][i], q[i], om["Z"], y, d, h),
                        rtol=1e-9)
        _directional_check(b["A"][i], b["B"][i], b["C"][i], b["D"][i], q[i], om["Z"], y, d, h, g, rng, n_dirs=2)


def test_gradient_with_the_gensys_solver():
    """The reference's default estimation solver is gensys (`configure(..., solver="gensys")`, statespace.py:832) and its
    gradient is the same implicit-function adjoint of A + B T + C T^2 = 0 whichever solver produced T
    (`o1_policy_function_adjoints`, shared.py:12-71; gensys.py:668-676): logp + gradient through the gensys launches against the
    cycle-reduction route (logp 1e-9, every cotangent 1e-7 of its scale) and against directional differences of the oracle."""
    rng = np.random.default_rng(4)
    nb = 6
    b = wl.sw_shaped_batch(nb, first_draw=40)
    om = wl.sw_shaped_observation_model()
    q = b["sigma"] ** 2
    y = om["y"][:60].copy()
    h = om["Hdiag"].copy()
    d = rng.normal(0, 0.01, 7)
    cr = batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], y, d=d, Hdiag=h, tol=1e-13,
                                                max_iter=200)
    gs = batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], y, d=d, Hdiag=h, tol=1e-8,
                                                solver="gensys")
    assert np.all(cr["status"] == 0) and np.all(gs["status"] == 0)
    assert_allclose(gs["logp"], cr["logp"], rtol=1e-9)
    for key in ("A_bar", "B_bar", "C_bar", "D_bar", "q_bar", "d_bar", "h_bar"):
        sc = np.abs(cr[key]).reshape(nb, -1).max(axis=1).reshape((nb,) + (1,) * (cr[key].ndim - 1))
        assert np.all(np.abs(gs[key] - cr[key]) <= 1e-7 * np.maximum(sc, 1e-300)), key
    g = {k_: v[0] for k_, v in gs.items() if k_.endswith("_bar")}
    _directional_check(b["A"][0], b["B"][0], b["C"][0], b["D"][0], q[0], om["Z"], y, d, h, g, rng, n_dirs=2)


def test_gradient_sw_shaped_against_extrapolated_differences():
    """The same extrapolated-difference check on the 40-variable SW-shaped system (deflated solve, reduced filter with
    steady-state segments, 60 periods with a missing observation): 1e-7 relative."""
    rng = np.random.default_rng(8)
    b = wl.sw_shaped_batch(2, first_draw=300)
    om = wl.sw_shaped_observation_model()
    q = b["sigma"] ** 2
    y = om["y"][:60].copy()
    y[9, 2] = np.nan
    d = rng.normal(0, 0.01, 7)
    h = om["Hdiag"].copy()
    out = batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], y, d=d, Hdiag=h, tol=1e-14,
                                                 max_iter=200)
    assert np.all(out["status"] == 0)
    i = 1
    A, B, C, D = (b[x][i] for x in "ABCD")
    g = {k_: v[i] for k_, v in out.items() if k_.endswith("_bar")}
    maskA = (A != 0).any(axis=0)[None, :] * np.ones_like(A)
    dA = rng.standard_normal(A.shape) * maskA * 0.1
    dB, dC, dD = (rng.standard_normal(M.shape) * 0.1 for M in (B, C, D))
    dq = rng.standard_normal(q[i].shape) * q[i] * 0.3
    dd = rng.standard_normal(d.shape) * 0.1
    dh = rng.standard_normal(h.shape) * h * 0.3
    analytic = ((g["A_bar"] * dA).sum() + (g["B_bar"] * dB).sum() + (g["C_bar"] * dC).sum() + (g["D_bar"] * dD).sum()
                + (g["q_bar"] * dq).sum() + (g["d_bar"] * dd).sum() + (g["h_bar"] * dh).sum())

    def f(e):
        return oracle.solve_kalman_logp(A + e * dA, B + e * dB, C + e * dC, D + e * dD, np.diag(q[i] + e * dq), om["Z"], y,
                                        H=np.diag(h + e * dh), d=d + e * dd, tol=1e-15, max_iter=300)["logp"]

    h0 = 2e-3
    d1 = [(f(h0 / 2 ** j) - f(-h0 / 2 ** j)) / (2.0 * h0 / 2 ** j) for j in range(3)]
    d2 = [(4.0 * d1[j + 1] - d1[j]) / 3.0 for j in range(2)]
    d3 = (16.0 * d2[1] - d2[0]) / 15.0
    assert abs(d3 - d2[1]) <= 1e-5 * abs(d3)
    assert_allclose(analytic, d3, rtol=1e-7, atol=1e-8 * max(1.0, abs(d3)))


@pytest.mark.parametrize("n,ns,nl", [(52, 23, 15), (56, 25, 16)])
def test_gradient_on_the_56_wide_tile(n, ns, nl):
    """Systems of 49..56 variables (round 2: the 7 x 7-block instances of adjoint_kernel and grad_assemble_kernel, 101 and
    153 KB of LDS): logp against the oracle, directional derivatives against central differences of the oracle."""
    rng = np.random.default_rng(n)
    k = p = 7
    nb = 2
    sysm = [wl.sw_shaped_system(6100 + 3 * n + i, n=n, n_state=ns, n_lead=nl, k=k) for i in range(nb)]
    A, B, C, D = (np.stack([s_[j] for s_ in sysm]) for j in range(4))
    q = np.full((nb, k), 1e-4) * rng.uniform(0.5, 2.0, (nb, k))
    Z = np.zeros((p, n))
    Z[np.arange(p), np.arange(p)] = 1.0
    y = rng.normal(0, 0.02, (30, p))
    y[4, 1] = np.nan
    d = rng.normal(0, 0.01, p)
    h = np.full(p, 1e-4)
    out = batched.solve_kalman_logp_grad_batched(A, B, C, D, q, Z, y, d=d, Hdiag=h, tol=1e-13, max_iter=200)
    assert np.all(out["status"] == 0)
    for i in range(nb):
        g = {k_: v[i] for k_, v in out.items() if k_.endswith("_bar")}
        assert_allclose(out["logp"][i], _oracle_logp(A[i], B[i], C[i], D[i], q[i], Z, y, d, h), rtol=1e-9)
        _directional_check(A[i], B[i], C[i], D[i], q[i], Z, y, d, h, g, rng, n_dirs=1)


@pytest.mark.parametrize("n,ns,nl,jumps,u", [(20, 9, 5, False, 9), (24, 14, 6, False, 14), (40, 18, 12, False, 18), (40, 22, 10, False, 22),
                                             (40, 18, 12, True, 25), (44, 30, 8, False, 30), (48, 34, 6, False, 34),
                                             # more states than the compact Stein tile holds: the fused assembly + adjoint launch hands
                                             # these draws to the two-kernel path (24 of 40, 16 of 32)
                                             (40, 26, 10, False, 26), (30, 20, 6, False, 20)])
def test_gradient_across_the_tiles_of_the_reverse_sweep(n, ns, nl, jumps, u):
    """Every tile count of the reverse sweep's matrix-core products (kg_cov_products_mf: 2 BS - 1 or 2 BS tiles of four by the
    number u of retained variables: 3, 4 | 5, 6 | 7, 8; the 40-wide tile keeps the register-block products) and both forward sweeps
    with record output (kalman_mf_kernel for u <= 20 on the 24-wide tile, kalman_nt_kernel otherwise), against central differences of
    the CPU oracle; the compact Stein equation of the policy adjoints at the same sizes."""
    rng = np.random.default_rng(n + ns)
    shape = dict(n=n, n_state=ns, n_lead=nl, k=5, p=7, T_len=40)
    b = wl.sw_shaped_batch(2, seed0=9100 + n + ns, **shape)
    observed = tuple(range(ns, ns + 7)) if jumps else None
    om = wl.sw_shaped_observation_model(seed0=9100 + n + ns, observed=observed, **shape)
    assert np.count_nonzero((b["A"][0] != 0).any(axis=0) | (om["Z"] != 0).any(axis=0)) == u
    q = b["sigma"] ** 2
    y = om["y"].copy()
    y[7, 1] = np.nan
    d = rng.normal(0, 0.01, 7)
    h = om["Hdiag"].copy()
    out = batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], y, d=d, Hdiag=h, tol=1e-13, max_iter=200)
    assert np.all(out["status"] == 0)
    for i in range(2):
        g = {k_: v[i] for k_, v in out.items() if k_.endswith("_bar")}
        assert_allclose(out["logp"][i], _oracle_logp(b["A"][i], b["B"][i], b["C"][i], b["D"][i], q[i], om["Z"], y, d, h), rtol=1e-9)
        _directional_check(b["A"][i], b["B"][i], b["C"][i], b["D"][i], q[i], om["Z"], y, d, h, g, rng, n_dirs=1)


def test_gradient_fused_adjoint_launch_matches_the_two_kernel_path():
    """Round 6: the gradient pipeline's last launch is the reverse of the assembly AND the policy adjoints on one elimination of
    B + C T (adjoint_kernel<BS, false, true>).  Forcing the refinement rule on every draw (debug mode 1) makes that launch hand every
    draw to the two-kernel path (grad_assemble_kernel + adjoint_kernel with only_flag, then the refinement pass): same cotangents."""
    lib = _lib.load()
    b = wl.sw_shaped_batch(96)
    om = wl.sw_shaped_observation_model()
    q = b["sigma"] ** 2
    kw = dict(Hdiag=om["Hdiag"], tol=1e-10, max_iter=1000)
    fused = batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], om["y"][:60], **kw)
    try:
        _lib.check(lib.dsge_debug_adjoint_refine(1))
        two = batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], om["y"][:60], **kw)
    finally:
        _lib.check(lib.dsge_debug_adjoint_refine(0))
    assert not fused["status"].any() and not two["status"].any()
    assert np.array_equal(fused["logp"], two["logp"])
    for key in ("A_bar", "B_bar", "C_bar", "D_bar", "q_bar"):
        sc = np.abs(two[key]).reshape(96, -1).max(axis=1)
        er = np.abs(fused[key] - two[key]).reshape(96, -1).max(axis=1) / np.maximum(sc, 1e-300)
        assert er.max() <= 1e-8 and np.median(er) <= 1e-11, (key, float(er.max()), int(er.argmax()))


def test_gradient_failed_and_unsupported_draws():
    b = wl.sw_shaped_batch(3)
    om = wl.sw_shaped_observation_model()
    q = b["sigma"] ** 2
    A = b["A"].copy()
    A[1, 0, 0] = np.nan  # solver fails on draw 1
    out = batched.solve_kalman_logp_grad_batched(A, b["B"], b["C"], b["D"], q, om["Z"], om["y"][:20], Hdiag=om["Hdiag"],
                                                 tol=1e-10, max_iter=100)
    assert out["status"][0] == 0 and out["status"][2] == 0 and out["status"][1] != 0
    assert out["logp"][1] == -np.inf and np.all(out["D_bar"][1] == 0) and np.all(out["q_bar"][1] == 0)
    assert np.all(np.isfinite(out["A_bar"][[0, 2]]))
    # a dense design matrix handed to the SELECTOR entry point is flagged, never mis-evaluated (the Python wrapper routes it to
    # the dense-Z entry point by itself: dense_z=False forces the selector one)
    Zd = np.random.default_rng(2).standard_normal((7, 40))
    out = batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], q, Zd, om["y"][:20], tol=1e-10, max_iter=100,
                                                 dense_z=False)
    assert np.all(out["status"] & _lib.ST_GRAD_UNSUPPORTED) and np.all(np.isnan(out["logp"]))
    # ... and through the dense-Z entry point it is an ordinary evaluation; a failed draw there gives -inf and zero cotangents
    out = batched.solve_kalman_logp_grad_batched(A, b["B"], b["C"], b["D"], q, Zd, om["y"][:20], Hdiag=om["Hdiag"], tol=1e-10,
                                                 max_iter=100, return_Z_bar=True)
    assert out["status"][0] == 0 and out["status"][2] == 0 and out["status"][1] != 0
    assert out["logp"][1] == -np.inf and np.all(out["D_bar"][1] == 0) and np.all(out["Z_bar"][1] == 0)
    assert np.all(np.isfinite(out["A_bar"][[0, 2]])) and np.all(np.isfinite(out["Z_bar"][[0, 2]]))


def test_gradient_with_steady_state_segments():
    """Long sample: the gradient kernel's forward sweep switches to the steady-state recursion (only a_t stored), a
    missing entry later forces a full step and a second steady segment; the reverse sweep must walk both segments.
    Checked against finite differences of the oracle (which never switches) and against tol = 0."""
    rng = np.random.default_rng(3)
    nb = 2
    b = wl.sw_shaped_batch(nb)
    om = wl.sw_shaped_observation_model()
    q = b["sigma"] ** 2
    y = om["y"][:130].copy()
    y[95, 1] = np.nan
    y[96, :] = np.nan
    d = rng.normal(0, 0.01, 7)
    h = om["Hdiag"].copy()
    kw = dict(d=d, Hdiag=h, tol=1e-13, max_iter=200)
    out = batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], y, **kw)
    with _lib.options_scope({"kalman_steady_tol": 0.0}):
        out0 = batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], y, **kw)
    assert np.all(out["status"] == 0) and np.all(out0["status"] == 0)
    assert_allclose(out["logp"], out0["logp"], rtol=1e-12)
    for key in ("A_bar", "B_bar", "C_bar", "D_bar", "q_bar", "d_bar", "h_bar"):
        scale = np.abs(out0[key]).max()
        assert_allclose(out[key], out0[key], atol=1e-8 * scale, rtol=1e-7, err_msg=key)
    g = {k_: v[0] for k_, v in out.items() if k_.endswith("_bar")}
    _directional_check(b["A"][0], b["B"][0], b["C"][0], b["D"][0], q[0], om["Z"], y, d, h, g, rng, n_dirs=1)


def test_gradient_is_repeatable_and_independent_of_the_batch_size():
    """Regression (round 2): the per-draw record of the reverse sweep is T_len step records + P_0; round 1 strode the
    store by the step records alone, so the P_0 of draw d sat on the step-0 record of draw d + 1 -- a cross-workgroup race
    that made the cotangents of A, B, C non-repeatable (and wrong by tens of percent) for batches of ~100 draws and more
    while logp, q_bar and D_bar stayed exact.  A draw's gradient must not depend on the batch it travels in."""
    om = wl.sw_shaped_observation_model()
    nb = 256
    b = wl.sw_shaped_batch(nb)
    q = b["sigma"] ** 2
    kw = dict(Hdiag=om["Hdiag"], tol=1e-10, max_iter=1000)
    g1 = batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], om["y"], **kw)
    g2 = batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], om["y"], **kw)
    assert not g1["status"].any()
    for key in ("logp", "A_bar", "B_bar", "C_bar", "D_bar", "q_bar", "h_bar"):
        assert np.array_equal(g1[key], g2[key]), key
    sub = batched.solve_kalman_logp_grad_batched(b["A"][:40], b["B"][:40], b["C"][:40], b["D"][:40], q[:40], om["Z"], om["y"], **kw)
    for key in ("logp", "A_bar", "B_bar", "C_bar", "D_bar", "q_bar"):
        assert np.array_equal(g1[key][:40], sub[key]), key
    # every draw of the batch against a central difference of the device's own logp along one random direction of A, C
    rng = np.random.default_rng(0)
    dA = rng.standard_normal(b["A"].shape) * (b["A"] != 0)
    dC = rng.standard_normal(b["C"].shape) * (b["C"] != 0)
    h = 1e-6
    lp = batched.solve_kalman_logp_batched(b["A"] + h * dA, b["B"], b["C"] + h * dC, b["D"], q, om["Z"], om["y"], **kw)["logp"]
    lm = batched.solve_kalman_logp_batched(b["A"] - h * dA, b["B"], b["C"] - h * dC, b["D"], q, om["Z"], om["y"], **kw)["logp"]
    fd = (lp - lm) / (2 * h)
    an = np.einsum("bij,bij->b", g1["A_bar"], dA) + np.einsum("bij,bij->b", g1["C_bar"], dC)
    assert np.max(np.abs(fd - an) / np.maximum(np.abs(fd), 1.0)) < 1e-3  # (central differences, h = 1e-6; the race gave tens of percent)


@pytest.mark.parametrize("n,ns,nl", [(12, 5, 3), (30, 13, 8), (40, 18, 12), (47, 14, 8), (48, 23, 4), (53, 20, 16)])
def test_policy_adjoint_refinement_pass(n, ns, nl):
    """adjoint_kernel<BS, true>, the out-of-line refinement pass of the Stein solve (one step of iterative refinement for draws
    whose residual T_bar + M' S + C' S T' shows lost digits): forced on every draw (debug mode 1) it must reproduce the
    oracle's Kronecker LU (shared.py:53-71) to 1e-10 or as well as the unrefined solve (mode 2), and the default rule (mode 0)
    must be inside 1e-9 on every draw."""
    from geconpy_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(n)
    nb = 6
    sysm = [wl.sw_shaped_system(4100 + 17 * n + i, n=n, n_state=ns, n_lead=nl, k=3) for i in range(nb)]
    A, B, C, D, T = (np.stack([s_[j] for s_ in sysm]) for j in range(5))
    T_bar = rng.standard_normal(T.shape) * (T != 0).any(axis=1)[:, None, :]
    ref = [oracle.policy_function_adjoints(A[i], B[i], C[i], T[i], T_bar[i]) for i in range(nb)]
    errs = {}
    try:
        for mode in (2, 1, 0):
            _lib.check(lib.dsge_debug_adjoint_refine(mode))
            Ab, Bb, Cb, st = batched.policy_adjoints_batched(B, C, T, T_bar)
            assert (st == 0).all(), (mode, st)
            errs[mode] = np.array([max(np.abs(x[i] - r_).max() for x, r_ in zip((Ab, Bb, Cb), ref[i])) /
                                   max(1.0, np.abs(ref[i][0]).max()) for i in range(nb)])
    finally:
        _lib.check(lib.dsge_debug_adjoint_refine(0))
    # (in working precision a refinement step cannot go below ~cond x u: it may move a 1e-12 solution to 1e-11)
    assert (errs[1] <= np.maximum(errs[2], 1e-10)).all(), errs
    assert (errs[0] <= 1e-9).all() and (errs[1] <= 1e-9).all(), errs


def test_policy_adjoints_of_a_nearly_singular_draw():
    """SW-shaped draw 752: M = B + C T has cond 3e8 and G = -M^-T C' entries of 2e7, so G is known to ~0.6 in absolute terms,
    its float64 powers explode (true |G^16| = 1.7e5, computed 1e25) and the doubling of the first pass breaks down.  The
    second pass then falls back to X <- -M^-T (T_bar + C' X T') with an elimination per sweep (adj_stein_fixed_point;
    tools/adjoint_fixed_point_model.py is its CPU model).  Adjudicated in 50-digit arithmetic: the reference's Kronecker LU
    (shared.py:53-71) is 1.5e-8 from the exact S on this draw; the device has to be within 5e-8 of the exact S, T-side
    products included, and the gradient entry point must return a clear status and finite cotangents for the draw."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from adjoint_fixed_point_model import exact_stein

    b = wl.sw_shaped_batch(760)
    om = wl.sw_shaped_observation_model()
    sl_ = slice(748, 756)
    A, B, C, D = (b[x][sl_] for x in "ABCD")
    T, st, _ = batched.cycle_reduction_batched(A, B, C, tol=1e-8, max_iter=1000)
    assert (st == 0).all()
    T_bar = np.random.default_rng(1).standard_normal(T.shape)
    Ab, Bb, Cb, st = batched.policy_adjoints_batched(B, C, T, T_bar)
    assert (st == 0).all(), st
    i = 4  # draw 752
    Sx = exact_stein(B[i], C[i], T[i], T_bar[i], terms=90)
    ref = oracle.policy_function_adjoints(A[i], B[i], C[i], T[i], T_bar[i])
    sc = np.abs(Sx).max()
    err_ref = np.abs(ref[0] - Sx).max() / sc
    err_dev = np.abs(Ab[i] - Sx).max() / sc
    assert err_ref <= 5e-8 and err_dev <= 5e-8, (err_ref, err_dev)
    assert np.abs(Bb[i] - Sx @ T[i].T).max() <= 5e-8 * np.abs(Sx @ T[i].T).max()
    assert np.abs(Cb[i] - Sx @ T[i].T @ T[i].T).max() <= 5e-8 * np.abs(Sx @ T[i].T @ T[i].T).max()
    # the neighbours are ordinary draws: 1e-9 against the reference
    for j in (0, 3, 5):
        r = oracle.policy_function_adjoints(A[j], B[j], C[j], T[j], T_bar[j])
        assert np.abs(Ab[j] - r[0]).max() <= 1e-9 * max(1.0, np.abs(r[0]).max())
    g = batched.solve_kalman_logp_grad_batched(A, B, C, D, b["sigma"][sl_] ** 2, om["Z"], om["y"], Hdiag=om["Hdiag"], tol=1e-8,
                                               max_iter=1000)
    assert (g["status"] == 0).all(), g["status"]
    for key in ("A_bar", "B_bar", "C_bar", "D_bar", "q_bar"):
        assert np.isfinite(g[key]).all(), key


@pytest.mark.parametrize("batched_q", [False, True])
def test_gradient_with_a_full_shock_covariance(batched_q):
    """full_covariance (statespace.py:247-251: ``state_cov`` is a full k x k matrix): Q (k, k) or (batch, k, k) instead of the
    diagonal variances.  logp equals the oracle's with that Q; Q_bar = R' Gbar R is checked -- together with every other input
    block, which now flows through 2 (Gbar R) Q -- against Richardson-extrapolated central differences of the oracle along
    random SYMMETRIC directions dQ (1e-7 relative, SW-shaped 40-variable system), and it is symmetric; with a diagonal Q the
    diagonal of Q_bar and every other cotangent equal the diagonal path's bit for bit."""
    rng = np.random.default_rng(21)
    nb = 3
    b = wl.sw_shaped_batch(nb, first_draw=40)
    om = wl.sw_shaped_observation_model()
    k = b["D"].shape[2]
    y = om["y"][:50].copy()
    y[7, 1] = np.nan
    h = om["Hdiag"].copy()
    Ls = [np.diag(b["sigma"][i]) + 0.25 * np.tril(rng.standard_normal((k, k)), -1) * b["sigma"][i].mean() for i in range(nb)]
    Q = np.stack([L_ @ L_.T for L_ in Ls])
    Qarg = Q if batched_q else Q[0]
    out = batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], None, om["Z"], y, Hdiag=h, tol=1e-14, max_iter=200,
                                                 Q=Qarg)
    assert np.all(out["status"] == 0) and out["Q_bar"].shape == (nb, k, k)
    assert np.abs(out["Q_bar"] - out["Q_bar"].transpose(0, 2, 1)).max() <= 1e-9 * np.abs(out["Q_bar"]).max()
    i = 1
    Qi = Q[i] if batched_q else Q[0]
    A, B, C, D = (b[x][i] for x in "ABCD")
    ref = oracle.solve_kalman_logp(A, B, C, D, Qi, om["Z"], y, H=np.diag(h), tol=1e-15, max_iter=300)["logp"]
    assert abs(out["logp"][i] - ref) <= 1e-9 * abs(ref)
    g = {k_: v[i] for k_, v in out.items() if k_.endswith("_bar")}
    maskA = (A != 0).any(axis=0)[None, :] * np.ones_like(A)
    dA = rng.standard_normal(A.shape) * maskA * 0.1
    dB, dC, dD = (rng.standard_normal(M.shape) * 0.1 for M in (B, C, D))
    dQ = rng.standard_normal((k, k)) * np.abs(Qi).max() * 0.2
    dQ = 0.5 * (dQ + dQ.T)
    analytic = ((g["A_bar"] * dA).sum() + (g["B_bar"] * dB).sum() + (g["C_bar"] * dC).sum() + (g["D_bar"] * dD).sum()
                + (g["Q_bar"] * dQ).sum())

    def f(e):
        return oracle.solve_kalman_logp(A + e * dA, B + e * dB, C + e * dC, D + e * dD, Qi + e * dQ, om["Z"], y, H=np.diag(h),
                                        tol=1e-15, max_iter=300)["logp"]

    def central(step):
        return (f(step) - f(-step)) / (2.0 * step)

    h0 = 2e-3
    d1, d2, d3 = central(h0), central(h0 / 2), central(h0 / 4)
    r1, r2 = (4 * d2 - d1) / 3, (4 * d3 - d2) / 3
    fd = (16 * r2 - r1) / 15
    assert abs(analytic - fd) <= 1e-7 * max(1.0, abs(fd)), (analytic, fd)
    # a diagonal Q through the full-Q path against the diagonal path
    qd = b["sigma"] ** 2
    diag_path = batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], qd, om["Z"], y, Hdiag=h, tol=1e-14, max_iter=200)
    full_path = batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], None, om["Z"], y, Hdiag=h, tol=1e-14,
                                                       max_iter=200, Q=np.stack([np.diag(v) for v in qd]))
    assert np.allclose(full_path["logp"], diag_path["logp"], rtol=1e-12)
    for key in ("A_bar", "B_bar", "C_bar", "D_bar"):
        sc = np.abs(diag_path[key]).max()
        assert np.abs(full_path[key] - diag_path[key]).max() <= 1e-9 * sc, key
    dq = np.diagonal(full_path["Q_bar"], axis1=1, axis2=2)
    assert np.abs(dq - diag_path["q_bar"]).max() <= 1e-9 * np.abs(diag_path["q_bar"]).max()


@pytest.mark.parametrize("batched_z", [False, True])
def test_gradient_with_a_dense_design_matrix(batched_z):
    """Observation equations (statespace.py:298-332): rows of Z that are linear combinations of variables.  The dense-Z entry
    point carries o_t = Z x_t as p extra variables (T_aug = [[T, 0], [Z T, 0]], R_aug = [R; Z R]) and maps the cotangents back;
    logp equals the oracle's dense-Z filter (1e-9), every cotangent -- A, B, C, D, q, d, H and Z itself -- is checked against
    Richardson-extrapolated central differences of the oracle (1e-7 relative, 40-variable SW-shaped system with missing
    observations), and on a selector Z the dense route reproduces the selector route (1e-9)."""
    rng = np.random.default_rng(33)
    nb = 3
    b = wl.sw_shaped_batch(nb, first_draw=80)
    om = wl.sw_shaped_observation_model()
    n = b["A"].shape[1]
    p = om["Z"].shape[0]
    q = b["sigma"] ** 2
    y = om["y"][:50].copy()
    y[7, 1] = np.nan
    y[13] = np.nan
    h = om["Hdiag"].copy()
    d = rng.normal(0, 0.01, p)
    Zs = np.stack([om["Z"] + 0.15 * rng.standard_normal(om["Z"].shape) * (rng.random(om["Z"].shape) < 0.15) for _ in range(nb)])
    Zarg = Zs if batched_z else Zs[0]
    out = batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], q, Zarg, y, d=d, Hdiag=h, tol=1e-14, max_iter=200,
                                                 return_Z_bar=True)
    assert np.all(out["status"] == 0) and out["Z_bar"].shape == (nb, p, n)
    i = 1
    Zi = Zs[i] if batched_z else Zs[0]
    A, B, C, D = (b[x][i] for x in "ABCD")
    ref = oracle.solve_kalman_logp(A, B, C, D, np.diag(q[i]), Zi, y, H=np.diag(h), d=d, tol=1e-15, max_iter=300)["logp"]
    assert abs(out["logp"][i] - ref) <= 1e-9 * abs(ref), (out["logp"][i], ref)
    g = {k_: v[i] for k_, v in out.items() if k_.endswith("_bar")}
    maskA = (A != 0).any(axis=0)[None, :] * np.ones_like(A)
    dA = rng.standard_normal(A.shape) * maskA * 0.1
    dB, dC, dD = (rng.standard_normal(M.shape) * 0.1 for M in (B, C, D))
    dq = rng.standard_normal(q[i].shape) * q[i] * 0.3
    dd = rng.standard_normal(d.shape) * 0.1
    dh = rng.standard_normal(h.shape) * h * 0.3
    dZ = rng.standard_normal(Zi.shape) * 0.1
    analytic = ((g["A_bar"] * dA).sum() + (g["B_bar"] * dB).sum() + (g["C_bar"] * dC).sum() + (g["D_bar"] * dD).sum()
                + (g["q_bar"] * dq).sum() + (g["d_bar"] * dd).sum() + (g["h_bar"] * dh).sum() + (g["Z_bar"] * dZ).sum())
    only_Z = (g["Z_bar"] * dZ).sum()

    def f(e, z_only=False):
        if z_only:
            return oracle.solve_kalman_logp(A, B, C, D, np.diag(q[i]), Zi + e * dZ, y, H=np.diag(h), d=d, tol=1e-15, max_iter=300)["logp"]
        return oracle.solve_kalman_logp(A + e * dA, B + e * dB, C + e * dC, D + e * dD, np.diag(q[i] + e * dq), Zi + e * dZ, y,
                                        H=np.diag(h + e * dh), d=d + e * dd, tol=1e-15, max_iter=300)["logp"]

    def extrapolated(z_only):
        h0 = 2e-3
        d1 = [(f(h0 / 2 ** j, z_only) - f(-h0 / 2 ** j, z_only)) / (2.0 * h0 / 2 ** j) for j in range(3)]
        d2 = [(4.0 * d1[j + 1] - d1[j]) / 3.0 for j in range(2)]
        return (16.0 * d2[1] - d2[0]) / 15.0

    fd_all, fd_z = extrapolated(False), extrapolated(True)
    assert abs(analytic - fd_all) <= 1e-7 * max(1.0, abs(fd_all)), (analytic, fd_all)
    assert abs(only_Z - fd_z) <= 1e-7 * max(1.0, abs(fd_z)), (only_Z, fd_z)
    # a selector through the dense route = the selector route
    sel = batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], y, d=d, Hdiag=h, tol=1e-14, max_iter=200)
    den = batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], y, d=d, Hdiag=h, tol=1e-14, max_iter=200,
                                                 dense_z=True)
    assert np.all(den["status"] == 0)
    assert np.abs(den["logp"] - sel["logp"]).max() <= 1e-10 * np.abs(sel["logp"]).max()
    for key in ("A_bar", "B_bar", "C_bar", "D_bar", "q_bar", "d_bar", "h_bar"):
        assert np.abs(den[key] - sel[key]).max() <= 1e-8 * np.abs(sel[key]).max(), key


def test_gradient_split_sweeps_match_the_one_kernel_path():
    """dsge_options.kalman_grad_split (round 5; default 2 since round 6): the forward sweep runs as a logp kernel with record output
    (kalman_nt_kernel<.., REC>), the reverse sweep as a kernel of its own.  Same recursion and the same records as the one-kernel path
    (kalman_grad_split = 0): logp to rounding, every cotangent to 1e-10 of its scale in the median and 1e-7 at worst -- on 768 distinct SW-shaped draws (the nearly
    singular draw 752 among them) with complete data, with scattered missing entries (the mask changes: several steady segments)
    and with kalman_steady_tol = 0, and on RBC-sized systems (the 16-wide tile)."""
    rng = np.random.default_rng(8)
    om = wl.sw_shaped_observation_model()
    b = wl.sw_shaped_batch(768)
    q = b["sigma"] ** 2
    y_miss = om["y"].copy()
    y_miss[rng.random(y_miss.shape) < 0.03] = np.nan
    y_miss[120] = np.nan
    d = rng.normal(0, 0.01, 7)
    cases = [("complete", om["y"], {}), ("missing", y_miss, {}), ("full recursion", om["y"][:60], {"kalman_steady_tol": 0.0})]
    for name, y, extra in cases:
        kw = dict(d=d, Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000)
        one = batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], y, options=dict(extra, kalman_grad_split=0), **kw)
        two = batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], y, options=dict(extra, kalman_grad_split=2), **kw)
        mid = batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], y, options=dict(extra, kalman_grad_split=1), **kw)
        assert np.array_equal(one["status"], two["status"]) and not one["status"].any(), name
        # the tail kernel (2) runs the steps of the reverse sweep (1) it replaces operation by operation: only the ORDER in which the
        # segment's sums meet the earlier ones differs
        for key in ("A_bar", "B_bar", "C_bar", "D_bar", "q_bar", "d_bar", "h_bar"):
            sc = np.abs(mid[key]).reshape(len(q), -1).max(axis=1)
            er = np.abs(two[key] - mid[key]).reshape(len(q), -1).max(axis=1) / np.maximum(sc, 1e-300)
            assert er.max() <= 1e-7 and np.median(er) <= 1e-11, (name, key, float(er.max()), int(er.argmax()))
        assert np.array_equal(two["logp"], mid["logp"])
        assert_allclose(two["logp"], one["logp"], rtol=1e-12, err_msg=name)
        for key in ("A_bar", "B_bar", "C_bar", "D_bar", "q_bar", "d_bar", "h_bar"):
            scale = np.abs(one[key]).reshape(len(q), -1).max(axis=1)
            err = np.abs(two[key] - one[key]).reshape(len(q), -1).max(axis=1)
            rel = err / np.maximum(scale, 1e-300)
            # (the two forward sweeps sum in different orders: 1e-13 on the filter's cotangents, which the policy adjoints of an
            #  ill-conditioned draw amplify -- 1e-8 on A_bar of draw 752, cond(B + C T) = 3e8; the typical draw sits at 1e-11)
            assert rel.max() <= 1e-7 and np.median(rel) <= 1e-10, (name, key, float(rel.max()), int(rel.argmax()), float(np.median(rel)))
    sysm = [wl.sw_shaped_system(9000 + i, n=12, n_state=5, n_lead=4, k=3) for i in range(64)]
    A, B, C, D = (np.stack([s_[j] for s_ in sysm]) for j in range(4))
    Z = np.zeros((2, 12))
    Z[0, 1] = 1.0
    Z[1, 7] = 1.0  # a state and a jump variable
    y = rng.standard_normal((80, 2)) * 0.05
    qs = np.full((64, 3), 0.01)
    kw = dict(Hdiag=np.full(2, 1e-4), tol=1e-10, max_iter=500)
    # (kalman_steady_tol = 0: the two forward sweeps test for the steady state on different quantities -- P+ against P -- and may
    #  switch a step apart, which on data the model did not generate moves logp by 1e-10)
    one = batched.solve_kalman_logp_grad_batched(A, B, C, D, qs, Z, y, options={"kalman_grad_split": 0, "kalman_steady_tol": 0.0}, **kw)
    two = batched.solve_kalman_logp_grad_batched(A, B, C, D, qs, Z, y, options={"kalman_grad_split": 2, "kalman_steady_tol": 0.0}, **kw)
    assert np.array_equal(one["status"], two["status"])
    ok = one["status"] == 0
    assert ok.sum() >= 48
    assert_allclose(two["logp"][ok], one["logp"][ok], rtol=1e-12)
    for key in ("A_bar", "B_bar", "C_bar", "D_bar", "q_bar", "h_bar"):
        scale = np.abs(one[key][ok]).max()
        assert_allclose(two[key][ok], one[key][ok], atol=1e-9 * scale, rtol=1e-8, err_msg=key)
