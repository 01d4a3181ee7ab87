"""GPU parity: HIP engine (through the C ABI / ctypes) vs the oracle and the golden vectors.

Tolerances (stated per SURVEY.md §8d): T, R max-abs <= 1e-8 (the reference's own
cross-solver tolerance, tests/model/test_perturbation.py:205-206 -- observed ~1e-13);
logp relative <= 1e-8 (BASELINE.json north_star -- observed ~1e-13); iteration counts and
status codes exact; draw indexing bit-exact.
"""
import os
import numpy as np
import pytest
from numpy.testing import assert_allclose

import oracle
from geconpy_amd import _lib, batched
from geconpy_amd import workloads as wl

pytestmark = pytest.mark.gpu

# Kernel-variant switches of a test go through dsge_options (per call / per host thread: dsge_options_push / _pop; ABI 8 has no
# process-wide setters): _set_option replaces this thread's pushed record by one that carries every override made so far
# (a test's `finally` sets its switch back to the default, so nothing leaks into the next test).
_OPTION_OVERRIDES = {}
_OPTION_PUSHED = [False]


def _set_option(name, value):
    import ctypes as _ct

    lib = _lib.load()
    _OPTION_OVERRIDES[name] = float(value) if name in ("kalman_steady_tol", "jitter_F", "jitter_P") else int(value)
    if _OPTION_PUSHED[0]:
        _lib.check(lib.dsge_options_pop())
        _OPTION_PUSHED[0] = False
    defaults = _lib.make_options()
    for key in [k for k, v in _OPTION_OVERRIDES.items() if getattr(defaults, k) == v]:
        del _OPTION_OVERRIDES[key]  # back at the default: no record stays pushed
    if _OPTION_OVERRIDES:
        rec = _lib.make_options(dict(_OPTION_OVERRIDES))
        _lib.check(lib.dsge_options_push(_ct.addressof(rec)))
        _OPTION_PUSHED[0] = True



def _set_cr_deflation(on):
    """cr_deflation switch + forget the measured number of static variables (what the removed dsge_set_cr_deflation did)."""
    _set_option("cr_deflation", on)
    _lib.check(_lib.load().dsge_forget_measured_shapes())


T_ATOL = 1e-10
LOGP_RTOL = 1e-9


def _stack(g, keys):
    return tuple(np.stack([g[f"{k}_{x}"] for k in keys]) for x in "ABCD")


@pytest.mark.parametrize("key", ["one_block", "rbc_2_block", "full_nk"])
def test_cycle_reduction_reference_goldens(ref_goldens, key):
    g = ref_goldens
    A, B, C, D = (g[f"{key}_{x}"][None] for x in "ABCD")
    for tol, it_ref in zip(g["cr_tols"], g[f"{key}_ref_cr_iters"]):
        T, status, n_iter = batched.cycle_reduction_batched(A, B, C, max_iter=1000, tol=float(tol))
        assert status[0] == 0 and n_iter[0] == it_ref
        assert_allclose(T[0], g[f"{key}_ref_cr_T"], atol=T_ATOL, rtol=0)
    T, status, _ = batched.cycle_reduction_batched(A, B, C, max_iter=1000, tol=1e-8)
    R, resid = batched.selection_batched(B, C, D, T, A=A)
    assert_allclose(R[0], g[f"{key}_ref_cr_R"], atol=T_ATOL, rtol=0)
    assert_allclose(T[0], g[f"{key}_ref_gensys_T"], atol=1e-8, rtol=1e-8)  # gensys == CR, reference tolerance
    assert resid[0] < 1e-20


def test_cycle_reduction_rbc_draws(rbc_golden):
    g = rbc_golden
    th = {k[6:]: g[k] for k in g.files if k.startswith("theta_")}
    A, B, C, D = wl.rbc_linearized_jacobians(**th)
    T, status, n_iter = batched.cycle_reduction_batched(A, B, C, max_iter=1000, tol=1e-8)
    assert np.all(status == 0)
    assert np.array_equal(n_iter, g["ref_cr_iters"])
    assert_allclose(T, g["ref_cr_T"], atol=T_ATOL, rtol=0)
    assert_allclose(T, g["ref_gensys_T"], atol=1e-8, rtol=1e-8)
    R = batched.selection_batched(B, C, D, T)
    assert_allclose(R, g["ref_gensys_R"], atol=1e-8, rtol=1e-8)


def test_cycle_reduction_sw_shaped(sw_golden):
    g = sw_golden
    b = wl.sw_shaped_batch(int(g["n_draws"]))
    T, status, n_iter = batched.cycle_reduction_batched(b["A"], b["B"], b["C"], max_iter=1000, tol=1e-8)
    assert np.all(status == 0)
    assert np.array_equal(n_iter, g["ref_cr_iters"])
    assert_allclose(T, g["ref_cr_T"], atol=T_ATOL, rtol=0)
    assert_allclose(T, b["T_star"], atol=1e-10, rtol=0)


def test_failure_cases(failure_golden):
    g = failure_golden
    names = ["ok", "nonunique", "noexist", "coincident"]
    A, B, C, D = _stack(g, names)
    for max_iter, col in ((1000, 0), (50, 1)):
        T, status, n_iter = batched.cycle_reduction_batched(A, B, C, max_iter=max_iter, tol=1e-8)
        for i, name in enumerate(names):
            conv = bool(g[f"{name}_ref_cr_converged"][col])
            assert (status[i] == 0) == conv, name
            if not conv:
                assert np.all(T[i] == 0.0)  # cycle_reduction.py:181
                assert status[i] & _lib.ST_NOT_CONVERGED
            else:
                assert_allclose(T[i], g[f"{name}_ref_cr_T"], atol=T_ATOL)
    # fused path: failed draws give -inf and do not disturb their neighbours
    om = wl.sw_shaped_observation_model()
    q = np.full(7, 1e-4)
    out = batched.solve_kalman_logp_batched(A, B, C, D, q, om["Z"], om["y"], Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000)
    assert np.isfinite(out["logp"][0]) and out["status"][0] == 0
    assert np.all(out["logp"][1:] == -np.inf) and np.all(out["status"][1:] != 0)
    ref = oracle.solve_kalman_logp(A[0], B[0], C[0], D[0], np.diag(q), om["Z"], om["y"], H=np.diag(om["Hdiag"]))
    assert_allclose(out["logp"][0], ref["logp"], rtol=LOGP_RTOL)


def test_nan_inputs_do_not_hang():
    A = np.full((2, 6, 6), np.nan)
    T, status, n_iter = batched.cycle_reduction_batched(A, A, A, max_iter=20, tol=1e-8)
    assert np.all(status != 0) and np.all(T == 0)
    assert np.all(status & _lib.ST_NAN)


@pytest.mark.parametrize("n,k", [(3, 1), (8, 1), (9, 2), (17, 3), (24, 4), (33, 5), (40, 7), (48, 6), (50, 5), (56, 7), (64, 8)])
def test_sizes_cr_selection_lyapunov(n, k):
    nb = 5
    ns = max(1, n // 2)
    nl = max(1, n // 3)
    sysm = [wl.sw_shaped_system(100 + i, n=n, n_state=ns, n_lead=nl, k=k) for i in range(nb)]
    A, B, C, D, Tst = (np.stack([s[j] for s in sysm]) for j in range(5))
    T, status, n_iter = batched.cycle_reduction_batched(A, B, C, max_iter=1000, tol=1e-9)
    assert np.all(status == 0)
    assert_allclose(T, Tst, atol=1e-9)
    R, resid = batched.selection_batched(B, C, D, T, A=A)
    rng = np.random.default_rng(n)
    q = rng.uniform(0.5, 2.0, (nb, k))
    P0, RQR, st = batched.lyapunov_batched(T, R, q, q_mode="diag_batched")
    assert np.all(st == 0)
    for i in range(nb):
        Tc, conv, it = oracle.cycle_reduction_core(A[i], B[i], C[i], 1000, 1e-9)
        assert conv and it == n_iter[i]
        assert_allclose(T[i], Tc, atol=T_ATOL)
        Rc = oracle.compute_selection_matrix(B[i], C[i], D[i], Tc)
        assert_allclose(R[i], Rc, atol=1e-9, rtol=1e-9)
        # the residual arithmetic itself, from the device's own T (a sum of squares of cancelled terms: 1e-5 relative);
        # its size depends on how ill-conditioned the intermediate A1 of the draw are (1e-27 .. 1e-18 here)
        assert_allclose(resid[i], oracle.policy_residual(A[i], B[i], C[i], T[i]), rtol=1e-5, atol=1e-26)
        assert resid[i] < 1e-16
        RQRo = R[i] @ np.diag(q[i]) @ R[i].T  # the assembly arithmetic itself, from the device's own R and T
        assert_allclose(RQR[i], RQRo, atol=1e-12 * np.abs(RQRo).max())
        P0o = oracle.solve_discrete_lyapunov(T[i], RQRo)
        assert_allclose(P0[i], P0o, atol=1e-10 * np.abs(P0o).max())
        assert np.array_equal(P0[i], P0[i].T)


@pytest.mark.parametrize("m,k,p", [(5, 2, 2), (12, 3, 4), (24, 4, 3), (40, 7, 7), (56, 6, 9), (64, 8, 16)])
def test_kalman_sizes_full_q_general_z(m, k, p):
    nb, T_len = 4, 25
    rng = np.random.default_rng(m)
    T = rng.standard_normal((nb, m, m))
    for i in range(nb):
        T[i] *= rng.uniform(0.3, 0.95) / np.max(np.abs(np.linalg.eigvals(T[i])))
    R = rng.standard_normal((nb, m, k))
    L = rng.standard_normal((nb, k, k))
    Q = L @ np.swapaxes(L, 1, 2) + 0.1 * np.eye(k)
    Z = rng.standard_normal((nb, p, m))
    d = rng.standard_normal((nb, p))
    H = rng.uniform(0.05, 0.5, (nb, p))
    y = rng.standard_normal((T_len, p))
    y[3, 0] = np.nan
    if p > 1:
        y[7, 1] = oracle.MISSING_FILL
    y[11, :] = np.nan
    logp, st = batched.kalman_logp_batched(T, R, Q, Z, y, d=d, Hdiag=H)
    assert np.all(st == 0)
    for i in range(nb):
        ref = oracle.kalman_filter_logp(y, T[i], R[i], Q[i], Z[i], H=np.diag(H[i]), d=d[i])
        assert_allclose(logp[i], ref, rtol=LOGP_RTOL)
    # shared Z / d / H / diagonal shared Q
    qd = rng.uniform(0.5, 1.5, k)
    logp2, _ = batched.kalman_logp_batched(T, R, qd, Z[0], y, d=d[0], Hdiag=H[0])
    for i in range(nb):
        ref = oracle.kalman_filter_logp(y, T[i], R[i], np.diag(qd), Z[0], H=np.diag(H[0]), d=d[0])
        assert_allclose(logp2[i], ref, rtol=LOGP_RTOL)


def test_rbc_config1(rbc_golden):
    """BASELINE.json configs[0]: RBC at the calibrated point, observed Y, T_len = 100."""
    g = rbc_golden
    cal = wl.RBC_CALIBRATION
    A, B, C, D = (x[None] for x in wl.rbc_linearized_jacobians(**cal))
    out = batched.solve_kalman_logp_batched(A, B, C, D, np.array([cal["sigma_A"] ** 2]), g["cal_Z"], g["cal_y"],
                                            tol=1e-8, max_iter=1000, return_policy=True)
    assert out["status"][0] == 0 and np.isfinite(out["logp"][0])
    assert_allclose(out["T"][0], g["cal_ref_gensys_T"], atol=1e-8, rtol=1e-8)
    assert_allclose(out["R"][0], g["cal_ref_gensys_R"], atol=1e-8, rtol=1e-8)
    assert_allclose(out["logp"][0], float(g["cal_oracle_logp"]), rtol=LOGP_RTOL)


def test_backward_direct():
    rng = np.random.default_rng(3)
    nb, n, k = 6, 11, 2
    B = np.eye(n) + 0.1 * rng.standard_normal((nb, n, n))
    A = 0.3 * rng.standard_normal((nb, n, n))
    D = rng.standard_normal((nb, n, k))
    T, R = batched.backward_direct_batched(A, B, D)
    for i in range(nb):
        To, Ro = oracle.solve_policy_function_with_backward_direct(A[i], B[i], None, D[i])
        assert_allclose(T[i], To, atol=1e-12)
        assert_allclose(R[i], Ro, atol=1e-12)


def test_fused_sw_shaped_vs_golden(sw_golden):
    g = sw_golden
    nb = int(g["n_draws"])
    b = wl.sw_shaped_batch(nb)
    om = wl.sw_shaped_observation_model()
    out = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], om["y"],
                                            Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000, return_policy=True)
    assert np.all(out["status"] == 0)
    assert_allclose(out["logp"], g["oracle_logp"], rtol=LOGP_RTOL)
    assert_allclose(out["T"], g["ref_cr_T"], atol=T_ATOL)
    assert np.array_equal(out["n_iter"], g["ref_cr_iters"])
    out_m = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], g["y_missing"],
                                              Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000)
    assert_allclose(out_m["logp"], g["oracle_logp_missing"], rtol=LOGP_RTOL)


def test_draw_indexing_bit_exact_and_deterministic():
    """Size-independent properties at a larger batch: results are a pure function of the
    draw (permutation-equivariant, bit-exact) and two runs are bit-identical."""
    nb = 512
    base = wl.sw_shaped_batch(32)
    rng = np.random.default_rng(0)
    idx = rng.integers(0, 32, nb)
    A, B, C, D, q = base["A"][idx], base["B"][idx], base["C"][idx], base["D"][idx], base["sigma"][idx] ** 2
    om = wl.sw_shaped_observation_model()
    run = lambda sel: batched.solve_kalman_logp_batched(  # noqa: E731
        A[sel], B[sel], C[sel], D[sel], q[sel], om["Z"], om["y"], Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000)["logp"]
    ident = np.arange(nb)
    l1 = run(ident)
    l2 = run(ident)
    assert np.array_equal(l1, l2)
    perm = rng.permutation(nb)
    assert np.array_equal(run(perm), l1[perm])
    # identical draws give identical bits wherever they sit in the batch
    for j in range(32):
        vals = l1[idx == j]
        assert np.all(vals == vals[0])


def _kalman_inputs(nb, m, k, p, T_len, n_state, seed, selector=True):
    rng = np.random.default_rng(seed)
    T = np.zeros((nb, m, m))
    cols = np.sort(rng.choice(m, n_state, replace=False))
    for i in range(nb):
        M = rng.standard_normal((m, n_state))
        S = M[cols]
        M *= rng.uniform(0.3, 0.95) / np.max(np.abs(np.linalg.eigvals(S)))
        T[i][:, cols] = M
    R = rng.standard_normal((nb, m, k))
    q = rng.uniform(0.5, 1.5, (nb, k))
    Z = np.zeros((p, m))
    if selector:
        Z[np.arange(p), rng.choice(m, p, replace=False)] = rng.choice([1.0, 0.25, -2.0], p)
    else:
        Z[:] = rng.standard_normal((p, m))
    d = rng.standard_normal(p)
    H = rng.uniform(0.05, 0.5, p)
    y = rng.standard_normal((T_len, p))
    y[2, 0] = np.nan
    y[5, :] = np.nan
    if p > 2:
        y[9, 1:3] = oracle.MISSING_FILL
    return T, R, q, Z, d, H, y


@pytest.mark.parametrize("m,k,p,ns", [(8, 1, 1, 2), (12, 2, 3, 5), (24, 4, 4, 9), (40, 7, 7, 18), (40, 7, 8, 40),
                                      (37, 3, 5, 11), (64, 6, 8, 23)])
def test_kalman_selector_fast_path(m, k, p, ns):
    """kalman_sel_kernel (compact state block + selector Z) vs the oracle, and vs the general
    kernel (hints off)."""
    nb, T_len = 6, 30
    T, R, q, Z, d, H, y = _kalman_inputs(nb, m, k, p, T_len, ns, seed=m + p)
    logp, st = batched.kalman_logp_batched(T, R, q, Z, y, d=d, Hdiag=H, q_mode="diag_batched")
    assert np.all(st == 0)
    logp_gen, st2 = batched.kalman_logp_batched(T, R, q, Z, y, d=d, Hdiag=H, q_mode="diag_batched", n_state_hint=0,
                                                z_selector_hint=0)
    assert np.all(st2 == 0)
    for i in range(nb):
        ref = oracle.kalman_filter_logp(y, T[i], R[i], np.diag(q[i]), Z, H=np.diag(H), d=d)
        assert_allclose(logp[i], ref, rtol=LOGP_RTOL)
        assert_allclose(logp_gen[i], ref, rtol=LOGP_RTOL)


def test_kalman_wrong_hints_are_harmless():
    """Hints are verified on the device: a too-small state hint or a false selector claim must
    not change any result (the flagged draws are re-run by the general kernel)."""
    nb, m, k, p, T_len = 6, 24, 3, 4, 20
    T, R, q, Z, d, H, y = _kalman_inputs(nb, m, k, p, T_len, 9, seed=5)
    good, _ = batched.kalman_logp_batched(T, R, q, Z, y, d=d, Hdiag=H, q_mode="diag_batched")
    too_small, st = batched.kalman_logp_batched(T, R, q, Z, y, d=d, Hdiag=H, q_mode="diag_batched", n_state_hint=3)
    assert np.all(st == 0)
    assert_allclose(too_small, good, rtol=1e-12)
    Tg, Rg, qg, Zg, dg, Hg, yg = _kalman_inputs(nb, m, k, p, T_len, 9, seed=5, selector=False)
    lied, st = batched.kalman_logp_batched(Tg, Rg, qg, Zg, yg, d=dg, Hdiag=Hg, q_mode="diag_batched", z_selector_hint=1)
    honest, _ = batched.kalman_logp_batched(Tg, Rg, qg, Zg, yg, d=dg, Hdiag=Hg, q_mode="diag_batched")
    assert np.all(st == 0)
    assert_allclose(lied, honest, rtol=1e-11)  # general kernel (re-run) vs dense-Z fast path: same filter
    for i in range(nb):
        ref = oracle.kalman_filter_logp(yg, Tg[i], Rg[i], np.diag(qg[i]), Zg, H=np.diag(Hg), d=dg)
        assert_allclose(lied[i], ref, rtol=LOGP_RTOL)
    # mixed batch: one draw with a dense T among compact ones
    T2 = T.copy()
    T2[2] = 0.1 * np.random.default_rng(0).standard_normal((m, m))
    mixed, st = batched.kalman_logp_batched(T2, R, q, Z, y, d=d, Hdiag=H, q_mode="diag_batched", n_state_hint=9)
    assert np.all(st == 0)
    for i in range(nb):
        ref = oracle.kalman_filter_logp(y, T2[i], R[i], np.diag(q[i]), Z, H=np.diag(H), d=d)
        assert_allclose(mixed[i], ref, rtol=LOGP_RTOL)


def test_pivoting_is_exercised():
    """Row-scrambled and badly row-scaled systems have the same solution T; they force the
    blocked Gauss-Jordan to pivot across panels (zero / tiny diagonal entries)."""
    nb = 8
    b = wl.sw_shaped_batch(nb)
    rng = np.random.default_rng(11)
    A, B, C, D = (b[x].copy() for x in "ABCD")
    for i in range(nb):
        perm = rng.permutation(40)
        scale = 10.0 ** rng.uniform(-3, 3, 40)
        for M in (A, B, C, D):
            M[i] = scale[:, None] * M[i][perm]
    T, status, n_iter = batched.cycle_reduction_batched(A, B, C, max_iter=1000, tol=1e-9)
    assert np.all(status == 0)
    assert_allclose(T, b["T_star"], atol=1e-9)
    R = batched.selection_batched(B, C, D, T)
    for i in range(nb):
        assert_allclose(R[i], oracle.compute_selection_matrix(b["B"][i], b["C"][i], b["D"][i], b["T_star"][i]), atol=1e-8)
    # rbc_2_block golden has structural zeros on the diagonal of B
    # (covered by test_cycle_reduction_reference_goldens); here: an explicit anti-diagonal B
    n = 6
    Bm = np.fliplr(np.eye(n))[None] * np.arange(1, n + 1)[None, :, None]
    Am = 0.1 * rng.standard_normal((1, n, n))
    Dm = rng.standard_normal((1, n, 2))
    Tb, Rb = batched.backward_direct_batched(Am, Bm, Dm)
    assert_allclose(Tb[0], np.linalg.solve(-Bm[0], Am[0]), atol=1e-13)
    assert_allclose(Rb[0], -np.linalg.solve(Bm[0], Dm[0]), atol=1e-13)


# ------------------------------------------------------------------------------------------------
# gensys on the device (complex QZ + reordering + Jacobi SVDs)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("key", ["one_block", "rbc_2_block", "full_nk"])
def test_gensys_reference_goldens(ref_goldens, key):
    g = ref_goldens
    A, B, C, D = (g[f"{key}_{x}"][None] for x in "ABCD")
    out = batched.gensys_batched(A, B, C, D, tol=1e-8)
    assert list(out["eu"][0]) == list(g[f"{key}_ref_gensys_eu"]) and out["success"][0]
    assert_allclose(out["T"][0], g[f"{key}_ref_gensys_T"], atol=1e-9, rtol=0)
    assert_allclose(out["T"][0], g[f"{key}_ref_cr_T"], atol=1e-8, rtol=1e-8)  # tests/model/test_perturbation.py:205-206
    assert_allclose(out["R"][0], g[f"{key}_ref_gensys_R"], atol=1e-8, rtol=1e-8)


def test_gensys_rbc_and_sw(rbc_golden, sw_golden):
    th = {k[6:]: rbc_golden[k] for k in rbc_golden.files if k.startswith("theta_")}
    A, B, C, D = wl.rbc_linearized_jacobians(**th)
    out = batched.gensys_batched(A, B, C, D, tol=1e-8)
    assert np.all(out["success"]) and np.all(out["eu"] == np.array([1, 1, 0]))
    assert_allclose(out["T"], rbc_golden["ref_gensys_T"], atol=1e-9)
    assert_allclose(out["R"], rbc_golden["ref_gensys_R"], atol=1e-8, rtol=1e-8)
    nb = int(sw_golden["n_draws"])
    b = wl.sw_shaped_batch(nb)
    out = batched.gensys_batched(b["A"], b["B"], b["C"], b["D"], tol=1e-8)
    assert np.all(out["success"])
    assert np.array_equal(out["eu"], sw_golden["ref_gensys_eu"].astype(np.int32))
    assert_allclose(out["T"], sw_golden["ref_gensys_T"], atol=1e-9)
    assert_allclose(out["T"], b["T_star"], atol=1e-9)


def test_gensys_failure_codes(failure_golden):
    g = failure_golden
    names = ["ok", "nonunique", "noexist", "coincident"]
    A, B, C, D = _stack(g, names)
    out = batched.gensys_batched(A, B, C, D, tol=1e-8)
    for i, name in enumerate(names):
        assert list(out["eu"][i]) == list(g[f"{name}_ref_gensys_eu"]), name
        assert out["success"][i] == (name == "ok")
    assert_allclose(out["T"][0], g["ok_ref_gensys_T"], atol=1e-9)
    assert_allclose(out["T"][2], g["noexist_ref_gensys_T"], atol=1e-8)  # the reference still returns G1 there
    assert np.all(out["T"][3] == 0)  # coincident zeros: zero-filled (gensys.py:255-265)
    # fused path with solver="gensys"
    om = wl.sw_shaped_observation_model()
    q = np.full(7, 1e-4)
    f = batched.solve_kalman_logp_batched(A, B, C, D, q, om["Z"], om["y"], Hdiag=om["Hdiag"], solver="gensys", tol=1e-8)
    assert np.isfinite(f["logp"][0]) and np.all(f["logp"][1:] == -np.inf)
    ref = oracle.solve_kalman_logp(A[0], B[0], C[0], D[0], np.diag(q), om["Z"], om["y"], H=np.diag(om["Hdiag"]), solver="gensys")
    assert_allclose(f["logp"][0], ref["logp"], rtol=1e-8)  # north_star tolerance; gensys T differs from CR T at 1e-13


def test_gensys_random_structures_vs_oracle():
    rng = np.random.default_rng(5)
    systems, expect = [], []
    for trial in range(40):
        n = int(rng.integers(3, 9))
        ns_ = int(rng.integers(1, n))
        nl = int(rng.integers(1, n))
        A, B, C, D, _ = wl.sw_shaped_system(1000 + trial, n=n, n_state=ns_, n_lead=nl, k=1)
        if trial % 3 == 0:
            A[0] = 0
            C[0] = 0
        To, ok, euo = oracle.gensys_T_success(A, B, C, D, 1e-8)
        out = batched.gensys_batched(A[None], B[None], C[None], D[None], tol=1e-8)
        assert list(out["eu"][0]) == list(euo), (trial, out["eu"][0], euo)
        if ok:
            assert_allclose(out["T"][0], To, atol=1e-8)


def test_gensys_capacity_flag():
    """A draw whose pencil exceeds the capacity implied by a (too small) lead hint is flagged, not
    silently wrong."""
    b = wl.sw_shaped_batch(2)
    # n_lead_hint is the caller's promise of an upper bound: the same answer from the window path with a fresh capacity record,
    # with a cached one (second call) and from the single-launch kernel
    for split, cache in ((1, 1), (1, 1), (1, 0), (0, 1)):
        _set_option("gensys_split", split)
        try:
            out = batched.gensys_batched(b["A"], b["B"], b["C"], b["D"], tol=1e-8, n_lead_hint=5,
                                         options={"gensys_split": split, "gensys_shape_cache": cache})
        finally:
            _set_option("gensys_split", 1)
        assert np.all(out["status"] & _lib.ST_GENSYS_TOO_BIG) and np.all(out["eu"][:, 0] == -3) and np.all(out["T"] == 0)
    ok = batched.gensys_batched(b["A"], b["B"], b["C"], b["D"], tol=1e-8, n_lead_hint=12)
    assert np.all(ok["status"] == 0)


@pytest.mark.parametrize("n,k", [(9, 2), (24, 4), (40, 7)])
def test_policy_norms(n, k):
    """deterministic_norm / stochastic_norm diagnostics (gEconpy/model/statespace.py:1181-1204)."""
    nb = 5
    ns = max(1, n // 2)
    sysm = [wl.sw_shaped_system(300 + i, n=n, n_state=ns, n_lead=max(1, n // 3), k=k) for i in range(nb)]
    A, B, C, D, Tst = (np.stack([s[j] for s in sysm]) for j in range(5))
    rng = np.random.default_rng(n)
    T = Tst + 1e-3 * rng.standard_normal(Tst.shape)  # a perturbed policy so that the norms are not ~0
    R = np.stack([oracle.compute_selection_matrix(B[i], C[i], D[i], Tst[i]) for i in range(nb)])
    mask = np.zeros(n, dtype=bool)
    mask[:ns] = True
    mask[rng.integers(ns, n)] = True
    det, sto = batched.policy_norms_batched(A, B, C, D, T, R, mask)
    for i in range(nb):
        P = T[i][mask][:, mask]
        Q = R[i][mask]
        Rp = T[i][:, mask]
        det_ref = np.linalg.norm(A[i][:, mask] + B[i] @ Rp + C[i] @ Rp @ P)
        sto_ref = np.linalg.norm(B[i] @ R[i] + C[i] @ Rp @ Q + D[i])
        assert_allclose(det[i], det_ref, rtol=1e-10)
        assert_allclose(sto[i], sto_ref, rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize("m,k,p,ns", [(12, 2, 3, 5), (24, 4, 4, 9), (40, 7, 7, 18), (40, 7, 8, 40), (64, 6, 5, 23)])
def test_kalman_dense_z_fast_path(m, k, p, ns):
    """Dense design matrix (observation equations, statespace.py:297-332) on the compact-state fast path."""
    nb, T_len = 6, 30
    T, R, q, Z, d, H, y = _kalman_inputs(nb, m, k, p, T_len, ns, seed=100 + m + p, selector=False)
    logp, st = batched.kalman_logp_batched(T, R, q, Z, y, d=d, Hdiag=H, q_mode="diag_batched")
    assert np.all(st == 0)
    Zb = np.stack([Z * (1.0 + 0.1 * i) for i in range(nb)])  # per-draw design matrices
    logp_b, st_b = batched.kalman_logp_batched(T, R, q, Zb, y, d=d, Hdiag=H, q_mode="diag_batched")
    assert np.all(st_b == 0)
    for i in range(nb):
        ref = oracle.kalman_filter_logp(y, T[i], R[i], np.diag(q[i]), Z, H=np.diag(H), d=d)
        assert_allclose(logp[i], ref, rtol=LOGP_RTOL)
        ref_b = oracle.kalman_filter_logp(y, T[i], R[i], np.diag(q[i]), Zb[i], H=np.diag(H), d=d)
        assert_allclose(logp_b[i], ref_b, rtol=LOGP_RTOL)


# ------------------------------------------------------------------------------------------------
# edge cases (empty / degenerate / maximum sizes), the way the reference's tests probe its boundaries
# ------------------------------------------------------------------------------------------------
def test_empty_batch_and_empty_series():
    z3 = np.zeros((0, 5, 5))
    T, status, n_iter = batched.cycle_reduction_batched(z3, z3, z3)
    assert T.shape == (0, 5, 5) and status.shape == (0,)
    out = batched.gensys_batched(z3, z3, z3, np.zeros((0, 5, 1)))
    assert out["T"].shape == (0, 5, 5)
    # T_len = 0: the log-likelihood of no data is 0
    b = wl.sw_shaped_batch(2)
    om = wl.sw_shaped_observation_model()
    r = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], np.zeros((0, 7)),
                                          Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000)
    assert np.all(r["logp"] == 0.0) and np.all(r["status"] == 0)


def test_scalar_model_n1():
    # a y_{t-1} + b y_t + c E y_{t+1} + d e = 0  ->  c T^2 + b T + a = 0, stable root
    a, bb, c, dd = -0.5, 1.0, -0.2, -1.0
    A, B, C, D = (np.full((3, 1, 1), v) for v in (a, bb, c, dd))
    root = (-bb + np.sqrt(bb * bb - 4 * a * c)) / (2 * c)
    root2 = (-bb - np.sqrt(bb * bb - 4 * a * c)) / (2 * c)
    stable = root if abs(root) < 1 else root2
    T, status, _ = batched.cycle_reduction_batched(A, B, C, max_iter=200, tol=1e-12)
    assert np.all(status == 0)
    assert_allclose(T[:, 0, 0], stable, atol=1e-12)
    g = batched.gensys_batched(A, B, C, D, tol=1e-8)
    assert np.all(g["success"])
    assert_allclose(g["T"][:, 0, 0], stable, atol=1e-12)
    y = np.random.default_rng(0).standard_normal((12, 1))
    out = batched.solve_kalman_logp_batched(A, B, C, D, np.array([0.3]), np.ones((1, 1)), y, Hdiag=np.array([0.1]),
                                            tol=1e-12, max_iter=200, return_policy=True)
    ref = oracle.solve_kalman_logp(A[0], B[0], C[0], D[0], np.array([[0.3]]), np.ones((1, 1)), y, H=np.array([[0.1]]),
                                   tol=1e-12, max_iter=200)
    assert_allclose(out["logp"], ref["logp"], rtol=LOGP_RTOL)


def test_all_missing_and_missing_column():
    b = wl.sw_shaped_batch(3)
    om = wl.sw_shaped_observation_model()
    q = b["sigma"] ** 2
    y_all = np.full((20, 7), np.nan)
    r = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], y_all, Hdiag=om["Hdiag"], tol=1e-8,
                                          max_iter=1000)
    assert np.all(r["logp"] == 0.0)  # every step fully missing -> every ll_t = 0
    y_col = om["y"][:40].copy()
    y_col[:, 3] = oracle.MISSING_FILL  # one series never observed
    r = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], y_col, Hdiag=om["Hdiag"], tol=1e-8,
                                          max_iter=1000)
    for i in range(3):
        ref = oracle.solve_kalman_logp(b["A"][i], b["B"][i], b["C"][i], b["D"][i], np.diag(q[i]), om["Z"], y_col,
                                       H=np.diag(om["Hdiag"]))
        assert_allclose(r["logp"][i], ref["logp"], rtol=LOGP_RTOL)


def test_explosive_transition_is_flagged():
    """rho(T) > 1: the stationary covariance does not exist; the draw must be flagged, not crash/hang."""
    rng = np.random.default_rng(1)
    m, k, p = 10, 2, 2
    T = rng.standard_normal((2, m, m))
    T[0] *= 0.6 / np.max(np.abs(np.linalg.eigvals(T[0])))
    T[1] *= 1.3 / np.max(np.abs(np.linalg.eigvals(T[1])))
    R = rng.standard_normal((2, m, k))
    Z = np.zeros((p, m))
    Z[0, 0] = Z[1, 3] = 1.0
    y = rng.standard_normal((15, p))
    logp, st = batched.kalman_logp_batched(T, R, np.ones(k), Z, y, Hdiag=np.full(p, 0.1))
    assert st[0] == 0 and np.isfinite(logp[0])
    assert st[1] & _lib.ST_LYAP_FAIL and logp[1] == -np.inf


def test_full_size_batches_properties():
    """BASELINE sizes: 4096 SW-shaped draws (configs[2]) and 65536 RBC draws -- checked through
    size-independent properties (a sample against the oracle, status all-clear, repeatability)."""
    nb = 4096
    base = wl.sw_shaped_batch(64)
    rep = nb // 64
    A, B, C, D = (np.tile(base[x], (rep, 1, 1)) for x in "ABCD")
    q = np.tile(base["sigma"] ** 2, (rep, 1))
    om = wl.sw_shaped_observation_model()
    r = batched.solve_kalman_logp_batched(A, B, C, D, q, om["Z"], om["y"], Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000)
    assert np.all(r["status"] == 0)
    assert np.array_equal(r["logp"][:64], r["logp"][-64:])  # same draw, same bits, wherever it sits
    for i in (0, 17, 63):
        ref = oracle.solve_kalman_logp(base["A"][i], base["B"][i], base["C"][i], base["D"][i], np.diag(q[i]), om["Z"],
                                       om["y"], H=np.diag(om["Hdiag"]))
        assert_allclose(r["logp"][i], ref["logp"], rtol=LOGP_RTOL)
    # RBC, 65536 draws (the multi-GPU config's total, on one device)
    nb = 65536
    th = wl.rbc_prior_draws(nb, seed=2)
    A, B, C, D = wl.rbc_linearized_jacobians(**th)
    Z = np.zeros((1, 8))
    Z[0, 7] = 1.0
    y = np.random.default_rng(0).normal(0, 0.05, (200, 1))
    r = batched.solve_kalman_logp_batched(A, B, C, D, (th["sigma_A"] ** 2)[:, None], Z, y, tol=1e-8, max_iter=1000,
                                          q_mode="diag_batched")
    assert np.all(r["status"] == 0) and np.all(np.isfinite(r["logp"]))
    for i in (0, 12345, 65535):
        ref = oracle.solve_kalman_logp(A[i], B[i], C[i], D[i], np.array([[th["sigma_A"][i] ** 2]]), Z, y)
        assert_allclose(r["logp"][i], ref["logp"], rtol=LOGP_RTOL)


def test_solvability_check_batched(failure_golden):
    """The draw-batch driver (perturbation_diagnostics.py:105-161): labels and norms, in input order."""
    from geconpy_amd.diagnostics import solvability_check_batched

    g = failure_golden
    names = ["ok", "nonunique", "noexist", "coincident"]
    A, B, C, D = _stack(g, names)
    good = wl.sw_shaped_batch(3)
    A = np.concatenate([good["A"], A]); B = np.concatenate([good["B"], B])
    C = np.concatenate([good["C"], C]); D = np.concatenate([good["D"], D])
    for solver in ("cycle_reduction", "gensys"):
        out = solvability_check_batched(A, B, C, D, solver=solver, tol=1e-8)
        assert list(out["failure_step"][:4]) == [None] * 4
        assert all(f == "perturbation" for f in out["failure_step"][4:])
        assert np.all(out["norm_deterministic"][:4] < 1e-8) and np.all(out["norm_stochastic"][:4] < 1e-8)
        assert np.all(np.isnan(out["norm_deterministic"][4:]))
        # the reference's own formulas on the reference's partition
        for i in range(4):
            T, R = out["T"][i], out["R"][i]
            mask = np.abs(T).max(axis=0) >= 1e-8
            PP = np.where(np.abs(T) < 1e-8, 0, T); QQ = np.where(np.abs(R) < 1e-8, 0, R)
            det_ref = np.linalg.norm(A[i][:, mask] + B[i] @ PP[:, mask] + C[i] @ PP[:, mask] @ PP[mask][:, mask])
            sto_ref = np.linalg.norm(B[i] @ QQ + C[i] @ PP[:, mask] @ QQ[mask] + D[i])
            assert_allclose(out["norm_deterministic"][i], det_ref, atol=1e-13)
            assert_allclose(out["norm_stochastic"][i], sto_ref, atol=1e-13)
    # a wrong policy is caught by the norm gate
    out = solvability_check_batched(A[:2], B[:2], C[:2], D[:2], norm_tol=1e-30)
    assert all(f in ("deterministic_norm", "stochastic_norm") for f in out["failure_step"])
    up = solvability_check_batched(A[:2], B[:2], C[:2], D[:2], upstream_failed=[True, False])
    assert up["failure_step"][0] == "steady_state" and up["failure_step"][1] is None


@pytest.mark.parametrize("key", ["one_block", "rbc_2_block", "full_nk"])
def test_policy_adjoints_vs_kronecker_oracle(ref_goldens, key):
    """Device doubling solve of the adjoint Stein equation vs the reference's n^2 x n^2 Kronecker solve
    (gEconpy/solvers/shared.py:53-71, restated in oracle/shared.py)."""
    g = ref_goldens
    A, B, C, D = (g[f"{key}_{x}"] for x in "ABCD")
    T = g[f"{key}_ref_cr_T"]
    rng = np.random.default_rng(1)
    nb = 3
    Tbar = rng.standard_normal((nb,) + T.shape)
    Ab, Bb, Cb, st = batched.policy_adjoints_batched(np.tile(B, (nb, 1, 1)), np.tile(C, (nb, 1, 1)), np.tile(T, (nb, 1, 1)), Tbar)
    assert np.all(st == 0)
    for i in range(nb):
        Ar, Br, Cr = oracle.policy_function_adjoints(A, B, C, T, Tbar[i])
        scale = np.abs(Ar).max()
        assert_allclose(Ab[i], Ar, atol=1e-9 * scale)
        assert_allclose(Bb[i], Br, atol=1e-9 * scale)
        assert_allclose(Cb[i], Cr, atol=1e-9 * scale)


def test_policy_adjoints_sw_shaped():
    b = wl.sw_shaped_batch(4)
    rng = np.random.default_rng(2)
    Tbar = rng.standard_normal(b["A"].shape)
    Ab, Bb, Cb, st = batched.policy_adjoints_batched(b["B"], b["C"], b["T_star"], Tbar)
    assert np.all(st == 0)
    for i in range(4):
        Ar, Br, Cr = oracle.policy_function_adjoints(b["A"][i], b["B"][i], b["C"][i], b["T_star"][i], Tbar[i])
        scale = np.abs(Ar).max()
        assert_allclose(Ab[i], Ar, atol=1e-9 * scale)
        assert_allclose(Cb[i], Cr, atol=1e-9 * scale)


def _steady_steps(fn, nb):
    """Run fn() with the steady-step recorder armed; returns (fn's result, int32[nb] first steady step)."""
    import torch

    from geconpy_amd.engine import LogpEngine

    eng = LogpEngine(0)
    buf = torch.full((nb,), -7, dtype=torch.int32, device=eng.device)
    eng.record_steady_steps(buf)
    try:
        out = fn()
        torch.cuda.synchronize()
    finally:
        eng.record_steady_steps(None)
    return out, buf.cpu().numpy()


def test_kalman_steady_state_switch_matches_full_recursion():
    """The steady-state switch (default tol 1e-14) must actually engage on the SW-shaped workload and
    leave logp unchanged at the 1e-12 level relative to the step-for-step recursion (tol = 0) and
    within LOGP_RTOL of the oracle, which never switches."""
    nb = 64
    b = wl.sw_shaped_batch(nb)
    om = wl.sw_shaped_observation_model()
    q = b["sigma"] ** 2

    def run():
        return batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], om["y"],
                                                 Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000)

    assert _lib.make_options().kalman_steady_tol == 1e-14
    r_ss, at = _steady_steps(run, nb)
    assert np.all(r_ss["status"] == 0)
    assert np.all(at > 0) and np.all(at < 150), at  # every draw reaches its fixed point well before T_len
    _set_option("kalman_steady_tol", 0.0)
    try:
        r_full, at0 = _steady_steps(run, nb)
    finally:
        _set_option("kalman_steady_tol", 1e-14)
    assert np.all(at0 == -1)
    assert_allclose(r_ss["logp"], r_full["logp"], rtol=1e-12)
    for i in (0, 10, 63):
        ref = oracle.solve_kalman_logp(b["A"][i], b["B"][i], b["C"][i], b["D"][i], np.diag(q[i]), om["Z"], om["y"],
                                       H=np.diag(om["Hdiag"]))
        assert_allclose(r_ss["logp"][i], ref["logp"], rtol=LOGP_RTOL)


@pytest.mark.parametrize("selector", [True, False])
def test_kalman_steady_state_resumes_on_mask_change(selector):
    """Missing entries AFTER the switch: the step with a different mask must leave steady mode, run the
    full update from the current covariance, and (the mask being constant again) re-enter it."""
    nb, m, k, p, T_len, ns = 5, 24, 4, 4, 160, 9
    T, R, q, Z, d, H, y = _kalman_inputs(nb, m, k, p, T_len, ns, seed=77, selector=selector)
    T *= 0.5  # spectral radius <= 0.475: the covariance recursion settles within a few dozen steps
    y[90, 1] = np.nan
    y[91, 1] = np.nan
    y[120, :] = oracle.MISSING_FILL
    y[140:, 2] = np.nan  # a series that stops being observed: second fixed point with another mask

    def run():
        return batched.kalman_logp_batched(T, R, q, Z, y, d=d, Hdiag=H, q_mode="diag_batched")

    (logp, st), at = _steady_steps(run, nb)
    assert np.all(st == 0)
    assert np.all((at > 9) & (at < 90)), at
    for i in range(nb):
        ref = oracle.kalman_filter_logp(y, T[i], R[i], np.diag(q[i]), Z, H=np.diag(H), d=d)
        assert_allclose(logp[i], ref, rtol=LOGP_RTOL)


def test_kalman_steady_tol_validation():
    for bad in (-1.0, 1e-3):
        with pytest.raises(_lib.DsgeHipError):
            with _lib.options_scope({"kalman_steady_tol": bad}):
                pass
    assert _lib.make_options().kalman_steady_tol == 1e-14


def test_cycle_reduction_compact_equals_dense(sw_golden, ref_goldens, rbc_golden):
    """The column-compact kernel drops only exactly-zero columns of A and C: T, status and iteration
    counts must be IDENTICAL (bit for bit) to the dense kernel's, on the SW-shaped systems, the
    reference goldens and the RBC draws; a system with dense A and C must fall through to the dense
    kernel and still be solved."""
    lib = _lib.load()
    b = wl.sw_shaped_batch(int(sw_golden["n_draws"]))
    th = {k[6:]: rbc_golden[k] for k in rbc_golden.files if k.startswith("theta_")}
    sets = [tuple(b[x] for x in "ABC"), wl.rbc_linearized_jacobians(**th)[:3]]
    for key in ("one_block", "rbc_2_block", "full_nk"):
        sets.append(tuple(ref_goldens[f"{key}_{x}"][None] for x in "ABC"))
    for A, B, C in sets:
        assert (np.abs(A).sum(axis=-2) == 0).sum() > 0  # the structure the compact kernel exploits is there
        T1, st1, it1 = batched.cycle_reduction_batched(A, B, C, max_iter=1000, tol=1e-9)
        _set_option("cr_compact", 0)
        try:
            T0, st0, it0 = batched.cycle_reduction_batched(A, B, C, max_iter=1000, tol=1e-9)
        finally:
            _set_option("cr_compact", 1)
        assert np.array_equal(st0, st1) and np.array_equal(it0, it1) and np.all(st1 == 0)
        assert np.array_equal(T0, T1)
    # dense A and C: |S| + |L| = 2n > 8*ceil(n/8) -> dense kernel
    rng = np.random.default_rng(5)
    n = 12
    Tstar = rng.standard_normal((3, n, n))
    Tstar *= 0.5 / np.abs(np.linalg.eigvals(Tstar)).max(axis=1)[:, None, None]
    G = rng.standard_normal((3, n, n))
    G *= 0.4 / np.abs(np.linalg.eigvals(G)).max(axis=1)[:, None, None]
    M = np.eye(n) + 0.2 * rng.standard_normal((3, n, n))
    C = M @ G
    B = M - C @ Tstar
    A = -M @ Tstar
    T, st, _ = batched.cycle_reduction_batched(A, B, C, max_iter=1000, tol=1e-12)
    assert np.all(st == 0)
    assert_allclose(T, Tstar, atol=1e-9)


# ------------------------------------------------------------------------------------------------
# scan_cycle_reduction semantics and the gensys numpy wrappers
# ------------------------------------------------------------------------------------------------
def test_scan_cycle_reduction_semantics(ref_goldens, sw_golden):
    """_scan_cycle_reduction (cycle_reduction.py:246-294): A0-norm-only stopping rule, fixed trip count,
    T from whatever iterate was reached.  Steps taken must equal the oracle's exactly."""
    b = wl.sw_shaped_batch(int(sw_golden["n_draws"]))
    sets = [tuple(b[x] for x in "ABC")]
    for key in ("one_block", "rbc_2_block", "full_nk"):
        sets.append(tuple(ref_goldens[f"{key}_{x}"][None] for x in "ABC"))
    for A, B, C in sets:
        for max_iter, tol in ((50, 1e-7), (3, 1e-7), (50, 1e-2)):  # converged / trip count exhausted / loose
            T, st, n_steps = batched.scan_cycle_reduction_batched(A, B, C, max_iter=max_iter, tol=tol)
            assert np.all(st == 0)
            for i in range(A.shape[0]):
                T_ref, n_ref = oracle.scan_cycle_reduction(A[i], B[i], C[i], max_iter=max_iter, tol=tol)
                assert n_steps[i] == n_ref
                assert_allclose(T[i], T_ref, atol=1e-9 * max(1.0, np.abs(T_ref).max()), rtol=0)
    # the njit variant needs ||A2|| < tol as well: on these systems it takes at least as many steps
    A, B, C = sets[0]
    _, _, it_njit = batched.cycle_reduction_batched(A, B, C, max_iter=50, tol=1e-7)
    _, _, it_scan = batched.scan_cycle_reduction_batched(A, B, C, max_iter=50, tol=1e-7)
    assert np.all(it_scan <= it_njit)
    # NaN input: flagged, zero T, no hang
    An = A[:2].copy()
    An[0, 0, 0] = np.nan
    T, st, _ = batched.scan_cycle_reduction_batched(An, B[:2], C[:2], max_iter=50, tol=1e-7)
    assert st[0] & _lib.ST_NAN and np.all(T[0] == 0) and st[1] == 0


def test_fused_pipeline_with_scan_cycle_reduction():
    b = wl.sw_shaped_batch(6)
    om = wl.sw_shaped_observation_model()
    q = b["sigma"] ** 2
    r = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], om["y"][:60], Hdiag=om["Hdiag"],
                                          solver="scan_cycle_reduction", tol=1e-7, max_iter=50, return_policy=True)
    assert np.all(r["status"] == 0)
    for i in range(6):
        ref = oracle.solve_kalman_logp(b["A"][i], b["B"][i], b["C"][i], b["D"][i], np.diag(q[i]), om["Z"], om["y"][:60],
                                       H=np.diag(om["Hdiag"]), solver="scan_cycle_reduction", tol=1e-7, max_iter=50)
        assert r["n_iter"][i] == ref["n_iter"]
        assert_allclose(r["logp"][i], ref["logp"], rtol=LOGP_RTOL)


def test_numpy_gensys_wrapper_large_pencil_fallback():
    """a model whose pencil n + #lead exceeds what the raw-pencil kernel holds in LDS (~52) still solves through
    solve_policy_function_with_gensys: T and R come from the window-path kernels, embedded in the reference's (N, N) / (N, k)
    return shapes (gensys.py:617-631; callers read G_1[:n, :n] and impact[:n], :657-666)."""
    from geconpy_amd import solvers

    n, ns, nl, k = 48, 20, 14, 5  # N = 62
    A, B, C, D, T_star = wl.sw_shaped_system(991, n=n, n_state=ns, n_lead=nl, k=k)
    G_1, constant, impact, f_mat, f_wt, y_wt, gev, eu, loose = solvers.solve_policy_function_with_gensys(A, B, C, D, 1e-8)
    N = n + nl
    assert eu == [1, 1, 0] and G_1.shape == (N, N) and impact.shape == (N, k) and constant.shape == (N, 1)
    Tg, ok = oracle.gensys_T_success(A, B, C, D)[:2]
    assert ok
    assert_allclose(G_1[:n, :n], Tg, atol=1e-9)
    assert_allclose(G_1[:n, :n], T_star, atol=1e-9)
    assert_allclose(impact[:n], oracle.compute_selection_matrix(B, C, D, Tg), atol=1e-9)
    G1s, eus = solvers.solve_policy_function_with_gensys(A, B, C, D, 1e-8, return_all_matrices=False)
    assert eus == eu and np.array_equal(G1s, G_1)
    # the contract must not depend on the pencil size.  An INDETERMINATE draw of the same size (the generator's G
    # rescaled to spectral radius 1.5: fewer than #lead unstable roots, eu[1] = 0 -- SURVEY 8c iv) still returns G_1 as an
    # (N, N) array on this route, as the reference does (its consumer slices G_1[:n, :n] before reading eu, gensys.py:657-666)
    Minv = np.linalg.inv(B + C @ T_star)          # M = B + C T*  (B = M - C T*), G = M^-1 C
    G = Minv @ C
    G2 = G * (1.5 / np.max(np.abs(np.linalg.eigvals(G))))
    M = B + C @ T_star
    C2 = M @ G2
    B2 = M - C2 @ T_star
    eu_ref = oracle.gensys_T_success(A, B2, C2, D)[2]
    assert eu_ref[0] == 1 and eu_ref[1] == 0
    G1f, euf = solvers.solve_policy_function_with_gensys(A, B2, C2, D, 1e-8, return_all_matrices=False)
    assert isinstance(G1f, np.ndarray) and G1f.shape == (N, N) and list(euf[:2]) == [1, 0] and euf[2] == eu_ref[2]
    full = solvers.solve_policy_function_with_gensys(A, B2, C2, D, 1e-8)
    assert len(full) == 9 and full[0].shape == (N, N) and full[2].shape == (N, k) and full[7] == euf
    assert np.all(np.isfinite(full[0])) and full[3] is None and full[8] is None
    # coincident zeros (one equation zeroed): the 9-tuple of Nones also with return_all_matrices=False, like gensys()
    A0, B0, C0 = A.copy(), B.copy(), C.copy()
    A0[3] = B0[3] = C0[3] = 0.0
    zz = solvers.solve_policy_function_with_gensys(A0, B0, C0, D, 1e-8, return_all_matrices=False)
    assert len(zz) == 9 and zz[0] is None and zz[7][:2] == [-2, -2]


def test_numpy_gensys_wrappers(ref_goldens, failure_golden):
    """solve_policy_function_with_gensys (gensys.py:617-631) and the raw-pencil gensys() (:398-521) against the outputs of
    the reference's own _gensys_setup + _gensys_core (tests/golden/reference_goldens.npz): full-size G_1, impact from the
    QZ formula, generalized eigenvalues, eu."""
    from geconpy_amd import solvers

    g = ref_goldens
    for key in ("one_block", "rbc_2_block", "full_nk"):
        A, B, C, D = (g[f"{key}_{x}"] for x in "ABCD")
        n = A.shape[0]
        G_1, constant, impact, f_mat, f_wt, y_wt, gev, eu, loose = solvers.solve_policy_function_with_gensys(A, B, C, D, 1e-8)
        N = g[f"{key}_ref_gensys_G1"].shape[0]
        assert eu == [1, 1, 0] and G_1.shape == (N, N) and impact.shape == (N, D.shape[1]) and constant.shape == (N, 1)
        nu_ref = int(np.sum(~(np.abs(g[f"{key}_ref_gensys_gev"][:, 1]) < np.abs(g[f"{key}_ref_gensys_gev"][:, 0]))))
        assert f_mat.shape == (nu_ref, nu_ref) and f_wt.shape == (nu_ref, D.shape[1]) and y_wt.shape == (N, nu_ref)
        assert loose.shape == (N, N - n) and np.iscomplexobj(f_mat) and np.iscomplexobj(y_wt) and not np.iscomplexobj(loose)
        assert_allclose(G_1, g[f"{key}_ref_gensys_G1"], atol=1e-9)
        assert_allclose(G_1[:n, :n], g[f"{key}_ref_gensys_T"], atol=1e-9)
        assert_allclose(impact[:n, :], g[f"{key}_ref_gensys_R"], atol=1e-9)
        assert np.all(constant == 0.0)
        # generalized eigenvalues: same multiset of beta / alpha moduli (the order inside the stable / unstable groups is
        # LAPACK's business), stable group first
        ref_gev = g[f"{key}_ref_gensys_gev"]
        assert gev.shape == ref_gev.shape == (N, 2)
        with np.errstate(divide="ignore", invalid="ignore"):
            mod = np.abs(gev[:, 1]) / np.abs(gev[:, 0])
            mod_ref = np.abs(ref_gev[:, 1]) / np.abs(ref_gev[:, 0])
        fin, fin_ref = (mod > 1e-6) & (mod < 1e6), (mod_ref > 1e-6) & (mod_ref < 1e6)  # zero / infinite roots: counted
        assert_allclose(np.sort(mod[fin]), np.sort(mod_ref[fin_ref]), rtol=1e-7)
        assert np.sum(mod <= 1e-6) == np.sum(mod_ref <= 1e-6) and np.sum(~(mod < 1e6)) == np.sum(~(mod_ref < 1e6))
        ns = int(np.sum(mod_ref < 1.0))
        assert np.all(mod[:ns] < 1.0) and np.all(~(mod[ns:] < 1.0))
        assert np.all(gev[:, 1].imag == 0.0) and np.all(gev[:, 1].real >= 0.0)  # LAPACK's normalisation of beta
        G_1s, eu_s = solvers.solve_policy_function_with_gensys(A, B, C, D, 1e-8, return_all_matrices=False)
        assert eu_s == eu and np.array_equal(G_1s, G_1)
    f = failure_golden
    A, B, C, D = (f[f"coincident_{x}"] for x in "ABCD")
    out = solvers.solve_policy_function_with_gensys(A, B, C, D, 1e-8)
    assert out[7] == [-2, -2, 0] and all(m is None for m in out[:7])
    for name in ("nonunique", "noexist"):
        A, B, C, D = (f[f"{name}_{x}"] for x in "ABCD")
        out = solvers.solve_policy_function_with_gensys(A, B, C, D, 1e-8)
        assert out[7] == [int(v) for v in f[f"{name}_ref_gensys_eu"]]
        assert_allclose(out[0][:40, :40], f[f"{name}_ref_gensys_T"], atol=1e-8)  # the reference still returns G1 there


@pytest.mark.parametrize("key", ["one_block", "rbc_2_block", "full_nk", "nonunique", "arbitrary", "arbitrary_nonunique"])
def test_gensys_forward_outputs_vs_reference(key):
    """f_mat, f_wt, y_wt, loose of gensys' 9-tuple (gensys.py:367-393), formed on the device, against the outputs of the
    reference's own _gensys_core (tests/golden/gensys_forward.npz, make_gensys_forward_golden.py).  The three complex
    matrices are defined up to the unitary basis of the unstable block of the ordered Schur form, so the comparison is on what
    the forward solution uses: the Markov parameters y_wt f_mat^s f_wt, the spectrum of f_mat, the column space of y_wt --
    plus the defining equations  B22 f_mat = A22  in invariant form (f_mat similar to the reference's) and loose itself (real
    and basis independent; non-zero only for the non-unique systems, incl. one with a Pi that is not orthonormal)."""
    from geconpy_amd import solvers

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "gensys_forward.npz"))
    g0, g1, c, psi, pi = (g[f"{key}_{x}"] for x in ("g0", "g1", "c", "psi", "pi"))
    G_1, constant, impact, f_mat, f_wt, y_wt, gev, eu, loose = solvers.gensys(g0, g1, c, psi, pi, tol=float(g[f"{key}_tol"]))
    N, nu = g0.shape[0], g[f"{key}_ref_fmat"].shape[0]
    assert eu == [int(v) for v in g[f"{key}_ref_eu"]]
    assert f_mat.shape == (nu, nu) and f_wt.shape == (nu, psi.shape[1]) and y_wt.shape == (N, nu) and loose.shape == pi.shape
    assert_allclose(G_1, g[f"{key}_ref_G1"], atol=1e-8)
    assert_allclose(impact, g[f"{key}_ref_impact"], atol=1e-8)
    scale = max(1.0, float(np.abs(g[f"{key}_ref_markov"]).max()))
    P = f_wt
    for s_ in range(4):
        assert_allclose(y_wt @ P, g[f"{key}_ref_markov"][s_], atol=1e-8 * scale)
        P = f_mat @ P
    ev = np.linalg.eigvals(f_mat)
    ev = ev[np.lexsort((ev.imag, ev.real))]
    ref_ev = g[f"{key}_ref_fmat_eig"]
    # (infinite roots give a multiple, defective zero eigenvalue of f_mat: that one moves by eps^(1/multiplicity))
    assert_allclose(np.sort(np.abs(ev)), np.sort(np.abs(ref_ev)), rtol=1e-7, atol=1e-5)
    # every eigenvalue of f_mat = B22^-1 A22 is the reciprocal of an UNSTABLE root beta / alpha: modulus <= 1 (+ rounding)
    assert np.all(np.abs(ev) <= 1.0 + 1e-8)
    assert_allclose(y_wt @ np.linalg.pinv(y_wt), g[f"{key}_ref_ywt_proj"], atol=1e-7)
    assert_allclose(loose, g[f"{key}_ref_loose"], atol=1e-8 * max(1.0, float(np.abs(g[f"{key}_ref_loose"]).max())))
    if key.endswith("nonunique"):
        assert np.abs(loose).max() > 0.1  # the case the output exists for


def test_raw_pencil_gensys_arbitrary_inputs(ref_goldens):
    """gensys(g0, g1, c, psi, pi) on pencils that do NOT come from _gensys_setup, against the oracle's restatement of
    _gensys_core: a non-zero constant, a Pi whose columns are neither orthogonal nor normalised (only its column space
    matters), a row-scaled / row-mixed pencil."""
    from geconpy_amd import solvers

    g = ref_goldens
    rng = np.random.default_rng(11)
    for key in ("one_block", "rbc_2_block", "full_nk"):
        A, B, C, D = (g[f"{key}_{x}"] for x in "ABCD")
        g0, g1, c, psi, pi = oracle.gensys_setup(A, B, C, D, 1e-8)
        s0, s1, s2, s3, s4 = solvers.gensys_setup(A, B, C, D, 1e-8)
        for u, v in zip((g0, g1, c, psi, pi), (s0, s1, s2, s3, s4)):
            assert np.array_equal(u, v)  # index arithmetic: bit-identical pencil
        N = g0.shape[0]
        M = np.eye(N) + 0.3 * rng.standard_normal((N, N))  # mixes the equations: same solution
        Rm = np.eye(pi.shape[1]) + 0.5 * rng.standard_normal((pi.shape[1], pi.shape[1]))
        c2 = 0.1 * rng.standard_normal((N, 1))
        args = (M @ g0, M @ g1, M @ c2, M @ psi, M @ pi @ Rm)
        ref = oracle.gensys(*args, tol=1e-8)
        out = solvers.gensys(*args, tol=1e-8)
        assert out[7] == [int(v) for v in ref[7]] == [1, 1, 0]
        assert_allclose(out[0], ref[0], atol=1e-8)
        assert_allclose(out[2], ref[2], atol=1e-8)
        assert_allclose(out[0], g[f"{key}_ref_gensys_G1"], atol=1e-8)  # and unchanged by the mixing
        # the constant: Sims' formula (the reference's omits G0^-1 on the stable block and is basis dependent for c != 0,
        # include/dsge_hip.h).  Its defining property: the fixed point y* = (I - G1)^-1 C of y_t = G1 y_{t-1} + C solves
        # the original system without shocks and expectational errors, (g0 - g1) y* = c.
        g0m, g1m, cm = args[0], args[1], args[2]
        ystar = np.linalg.solve(np.eye(N) - out[0], out[1])
        assert_allclose((g0m - g1m) @ ystar, cm, atol=1e-7 * max(1.0, np.abs(ystar).max()))
        assert np.all(solvers.gensys(args[0], args[1], 0 * cm, args[3], args[4])[1] == 0.0)


    # batched driver, all four solvers give the same policy on a healthy system
    b = wl.sw_shaped_batch(4)
    outs = {s: solvers.solve_policy_functions_batched(b["A"], b["B"], b["C"], b["D"], solver=s, tol=1e-9)
            for s in ("cycle_reduction", "scan_cycle_reduction", "gensys")}
    for s, o in outs.items():
        assert np.all(o["success"]), s
        assert_allclose(o["T"], b["T_star"], atol=1e-9)
        assert np.all(o["resid"] < 1e-16)
    assert np.array_equal(outs["gensys"]["eu"], np.tile([1, 1, 0], (4, 1)))


@pytest.mark.parametrize("m,k,p", [(9, 2, 3), (24, 4, 4), (40, 7, 7)])
def test_autocorrelation_matrices(m, k, p):
    """sample_autocorrelation_matrices (statespace.py:1262-1300) / _compute_autocovariance_matrix
    (covariance.py:133-161) for a batch of draws in one launch."""
    nb = 5
    T, R, q, Z, _d, H, _y = _kalman_inputs(nb, m, k, p, 10, max(2, m // 2), seed=100 + m)
    for kw in (dict(n_lags=6, lag_step=1, correlation=True), dict(n_lags=3, lag_step=4, correlation=True),
               dict(n_lags=5, lag_step=1, correlation=False)):
        acf, st, sig = batched.autocorrelation_matrices_batched(T, R, q, q_mode="diag_batched", return_sigma=True, **kw)
        assert np.all(st == 0) and acf.shape == (nb, kw["n_lags"] + 1, m, m)
        obs, st2 = batched.autocorrelation_matrices_batched(T, R, q, q_mode="diag_batched", Z=Z, Hdiag=H, **kw)
        assert np.all(st2 == 0) and obs.shape == (nb, kw["n_lags"] + 1, p, p)
        for i in range(nb):
            ref = oracle.autocorrelation_matrices(T[i], R[i], np.diag(q[i]), **kw)
            scale = np.abs(ref).max()
            assert_allclose(acf[i], ref, atol=1e-11 * scale, rtol=1e-9)
            assert_allclose(sig[i], oracle.solve_discrete_lyapunov(T[i], R[i] @ np.diag(q[i]) @ R[i].T), rtol=1e-10,
                            atol=1e-12 * np.abs(sig[i]).max())
            ref_o = oracle.autocorrelation_matrices(T[i], R[i], np.diag(q[i]), Z=Z, H=np.diag(H), **kw)
            assert_allclose(obs[i], ref_o, atol=1e-11 * np.abs(ref_o).max(), rtol=1e-9)
            if kw["correlation"]:
                assert_allclose(np.diag(acf[i, 0]), 1.0, rtol=1e-13)
    # no stationary distribution: flagged, NaN-filled
    Tx = T[:2].copy()
    Tx[1] *= 1.5 / np.max(np.abs(np.linalg.eigvals(Tx[1])))
    acf, st = batched.autocorrelation_matrices_batched(Tx, R[:2], q[:2], n_lags=2, q_mode="diag_batched")
    assert st[0] == 0 and np.all(np.isfinite(acf[0])) and (st[1] & _lib.ST_LYAP_FAIL) and np.all(np.isnan(acf[1]))


def _structured_system(rng, n, s_cols, l_cols, rho_t=0.9, rho_g=0.7):
    """A + B T + C T^2 = 0 with a known stable solvent T* whose non-zero columns are ``s_cols`` and a C whose
    non-zero columns are ``l_cols`` (SURVEY 8d recipe, arbitrary supports)."""
    s, l = len(s_cols), len(l_cols)
    Tstar = np.zeros((n, n))
    if s:
        M = rng.standard_normal((n, s))
        M *= rho_t * rng.uniform(0.5, 1.0) / max(np.max(np.abs(np.linalg.eigvals(M[s_cols]))), 1e-12)
        Tstar[:, s_cols] = M
    G = np.zeros((n, n))
    if l:
        Gm = rng.standard_normal((n, l))
        Gm *= rho_g * rng.uniform(0.4, 1.0) / max(np.max(np.abs(np.linalg.eigvals(Gm[l_cols]))), 1e-12)
        G[:, l_cols] = Gm
    Mx = np.eye(n) + 0.2 * rng.standard_normal((n, n))
    C = Mx @ G
    B = Mx - C @ Tstar
    A = -Mx @ Tstar
    return A, B, C, Tstar


def test_cycle_reduction_structure_fuzz():
    """Random sizes and column supports (overlapping, empty, too wide for the compact tile): the column-compact
    kernel, the dense kernel and the oracle must agree; compact vs dense bit for bit."""
    rng = np.random.default_rng(2026)
    lib = _lib.load()
    cases = []
    for n in (5, 8, 13, 16, 17, 24, 31, 40, 48):
        for _ in range(3):
            s = int(rng.integers(0, n + 1))
            l = int(rng.integers(0, n + 1))
            cases.append((n, np.sort(rng.choice(n, s, replace=False)), np.sort(rng.choice(n, l, replace=False))))
    cases.append((12, np.arange(12), np.arange(12)))      # dense A and C: s + l = 2n > tile -> dense kernel
    cases.append((10, np.array([], dtype=int), np.arange(3)))  # purely forward-looking: T = 0
    cases.append((10, np.arange(4), np.array([], dtype=int)))  # no leads: one step
    for n, s_cols, l_cols in cases:
        A, B, C, Tstar = _structured_system(rng, n, s_cols, l_cols)
        A3, B3, C3 = A[None], B[None], C[None]
        T1, st1, it1 = batched.cycle_reduction_batched(A3, B3, C3, max_iter=200, tol=1e-10)
        _set_option("cr_compact", 0)
        try:
            T0, st0, it0 = batched.cycle_reduction_batched(A3, B3, C3, max_iter=200, tol=1e-10)
        finally:
            _set_option("cr_compact", 1)
        assert st1[0] == 0 and st0[0] == 0, (n, len(s_cols), len(l_cols))
        assert it1[0] == it0[0] and np.array_equal(T1, T0), (n, len(s_cols), len(l_cols))
        T_or, ok, it_or = oracle.cycle_reduction_core(A, B, C, 200, 1e-10)
        assert ok and it_or == it1[0]
        assert_allclose(T1[0], Tstar, atol=1e-8)
        assert_allclose(T1[0], T_or, atol=1e-9)
        zero_cols = np.setdiff1d(np.arange(n), s_cols)
        assert np.all(T1[0][:, zero_cols] == 0.0)  # exactly zero, as the Kalman fast path relies on


def test_kalman_fuzz_against_oracle():
    """Random sizes, state supports, selector positions (observed states and observed non-states), weights,
    measurement errors (including none), intercepts and missing-data patterns, long enough for the steady-state
    switch to engage: logp vs the oracle filter, fast path vs general kernel."""
    rng = np.random.default_rng(77)
    for trial in range(14):
        m = int(rng.integers(3, 49))
        k = int(rng.integers(1, min(m, 8) + 1))
        p = int(rng.integers(1, min(k, 8) + 1))  # p <= k: no stochastic singularity without measurement error
        ns = int(rng.integers(1, m + 1))
        nb, T_len = 3, int(rng.integers(20, 120))
        T, R, q, Z, d, H, y = _kalman_inputs(nb, m, k, p, T_len, ns, seed=1000 + trial)
        T *= rng.uniform(0.3, 1.0)
        if trial % 3 == 0:
            H = None  # no measurement error: F = Z P Z' + jitter
        if trial % 4 == 1:
            y[rng.random(y.shape) < 0.15] = np.nan
        if trial % 5 == 2:
            y[T_len // 2:, 0] = oracle.MISSING_FILL
        scale = 10.0 ** rng.uniform(-3, 1)
        q = q * scale
        logp, st = batched.kalman_logp_batched(T, R, q, Z, y, d=d, Hdiag=H, q_mode="diag_batched")
        logp_gen, st2 = batched.kalman_logp_batched(T, R, q, Z, y, d=d, Hdiag=H, q_mode="diag_batched", n_state_hint=0,
                                                    z_selector_hint=0)
        assert np.all(st == 0) and np.all(st2 == 0), (trial, m, k, p, ns)
        for i in range(nb):
            ref = oracle.kalman_filter_logp(y, T[i], R[i], np.diag(q[i]), Z, H=None if H is None else np.diag(H), d=d)
            assert_allclose(logp[i], ref, rtol=LOGP_RTOL, err_msg=str((trial, m, k, p, ns)))
            assert_allclose(logp_gen[i], ref, rtol=LOGP_RTOL, err_msg=str((trial, m, k, p, ns)))


def test_kalman_tiny_kernel_matches_wave_kernels():
    """Small models (reduced filter <= 6 variables, p <= 3) take the thread-per-draw kernel: same logp as the
    wave-per-draw kernels (switch off -> on) and as the oracle; observed non-states, weights, missing data, d, H."""
    lib = _lib.load()
    rng = np.random.default_rng(314)
    cases = [(8, 1, 1, 2), (8, 2, 2, 3), (9, 3, 3, 3), (12, 2, 2, 4), (6, 1, 1, 5), (16, 3, 2, 2)]
    for m, k, p, ns in cases:
        nb, T_len = 70, 90  # more than one 64-thread block
        T, R, q, Z, d, H, y = _kalman_inputs(nb, m, k, p, T_len, ns, seed=500 + m + p)
        T *= 0.7
        y[40, 0] = np.nan
        logp1, st1 = batched.kalman_logp_batched(T, R, q, Z, y, d=d, Hdiag=H, q_mode="diag_batched")
        _set_option("kalman_tiny", 0)
        try:
            logp0, st0 = batched.kalman_logp_batched(T, R, q, Z, y, d=d, Hdiag=H, q_mode="diag_batched")
        finally:
            _set_option("kalman_tiny", 1)
        assert np.all(st1 == 0) and np.all(st0 == 0), (m, k, p, ns)
        assert_allclose(logp1, logp0, rtol=1e-11, err_msg=str((m, k, p, ns)))
        for i in (0, 33, 69):
            ref = oracle.kalman_filter_logp(y, T[i], R[i], np.diag(q[i]), Z, H=np.diag(H), d=d)
            assert_allclose(logp1[i], ref, rtol=LOGP_RTOL)
    # explosive draw inside a tiny batch: flagged, not crashed
    T, R, q, Z, d, H, y = _kalman_inputs(3, 8, 1, 1, 30, 2, seed=9)
    cols = np.flatnonzero(np.abs(T[1]).sum(axis=0))
    T[1] *= 1.4 / np.max(np.abs(np.linalg.eigvals(T[1][np.ix_(cols, cols)])))
    logp, st = batched.kalman_logp_batched(T, R, q, Z, y, d=d, Hdiag=H, q_mode="diag_batched")
    assert st[0] == 0 and st[2] == 0 and (st[1] & _lib.ST_LYAP_FAIL) and logp[1] == -np.inf


def test_cr_fused_selection_matches_explicit_formula():
    """Fused pipeline, cycle reduction: R taken from the final elimination (-A1_hat^-1 D) vs the explicit
    R = -(C T + B)^-1 D of the assemble kernel: same logp to 1e-11 on healthy draws, same statuses on failing ones,
    including draws that take the dense fallback kernel."""
    lib = _lib.load()
    om = wl.sw_shaped_observation_model()
    b = wl.sw_shaped_batch(24)
    q = b["sigma"] ** 2
    A = b["A"].copy()
    A[5, 0, 0] = np.nan                                 # a failing draw
    sets = [(A, b["B"], b["C"], b["D"], q, om["Z"], om["y"][:60], om["Hdiag"])]
    rng = np.random.default_rng(12)
    n = 12                                              # dense A and C: the compact kernel hands over to the dense one
    sysd = [_structured_system(rng, n, np.arange(n), np.arange(n), rho_t=0.8, rho_g=0.5) for _ in range(4)]
    Ad, Bd, Cd = (np.stack([s_[i] for s_ in sysd]) for i in range(3))
    Dd = rng.standard_normal((4, n, 2))
    Zd = np.zeros((2, n))
    Zd[0, 1] = Zd[1, 4] = 1.0
    sets.append((Ad, Bd, Cd, Dd, np.full((4, 2), 0.3), Zd, rng.standard_normal((40, 2)), np.array([0.1, 0.2])))
    for A_, B_, C_, D_, q_, Z_, y_, H_ in sets:
        kw = dict(Hdiag=H_, tol=1e-10, max_iter=200, q_mode="diag_batched")
        r1 = batched.solve_kalman_logp_batched(A_, B_, C_, D_, q_, Z_, y_, **kw)
        _set_option("cr_fused_selection", 0)
        try:
            r0 = batched.solve_kalman_logp_batched(A_, B_, C_, D_, q_, Z_, y_, **kw)
        finally:
            _set_option("cr_fused_selection", 1)
        assert np.array_equal(r1["status"], r0["status"])
        good = r0["status"] == 0
        assert good.sum() >= len(good) - 1
        assert_allclose(r1["logp"][good], r0["logp"][good], rtol=1e-11)
        assert np.all(r1["logp"][~good] == -np.inf)


def test_kalman_matches_statsmodels_and_arbitrary_precision():
    """Device filter (every kernel variant the hints select) against (i) statsmodels' log-likelihood at jitter = 0
    (independent third-party implementation; itself only 1.2e-8 accurate on the ill-conditioned RBC case) and (ii) the
    40-digit mpmath evaluation of the recursion at jitter 0 and 1e-8 (tests/golden/mp_kalman.npz)."""
    import os

    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(gold, "statsmodels_kalman.npz"))
    mpg = np.load(os.path.join(gold, "mp_kalman.npz"))
    for name in g["names"]:
        c = {key: g[f"{name}_{key}"] for key in ("T", "R", "Q", "Z", "H", "d", "y")}
        T3, R3 = c["T"][None], c["R"][None]
        for hints in (dict(), dict(n_state_hint=0, z_selector_hint=0)):
            for label, jit in (("", 0.0), ("_jitter", 1e-8)):
                logp, st = batched.kalman_logp_batched(T3, R3, c["Q"], c["Z"], c["y"], d=c["d"], Hdiag=np.diag(c["H"]).copy(),
                                                       q_mode="full", jitter=jit, **hints)
                assert st[0] == 0
                if jit == 0.0:
                    assert_allclose(logp[0], float(g[f"{name}_loglike"]), rtol=5e-8, err_msg=f"{name} {hints}")
                key = f"{name}_loglike_mp{label}"
                if key in mpg.files:
                    assert_allclose(logp[0], float(mpg[key]), rtol=1e-10, err_msg=f"{name}{label} {hints}")


def test_kalman_mfma_products_match_valu():
    """The FP64-MFMA prediction products (16 x 16 core tile + VALU fringe) against the VALU register-block products:
    same logp to 1e-11 for reduced sizes below, at and above one tile (u = 9, 16, 18, 24), and within LOGP_RTOL of the
    oracle."""
    lib = _lib.load()
    for m, k, p, ns in [(12, 2, 2, 9), (20, 4, 4, 16), (40, 7, 7, 18), (30, 6, 6, 24), (16, 3, 3, 10)]:
        nb, T_len = 5, 60
        T, R, q, Z, d, H, y = _kalman_inputs(nb, m, k, p, T_len, ns, seed=4000 + m)
        logp0, st0 = batched.kalman_logp_batched(T, R, q, Z, y, d=d, Hdiag=H, q_mode="diag_batched")
        _set_option("kalman_mfma", 1)  # experimental path, off by default
        try:
            logp1, st1 = batched.kalman_logp_batched(T, R, q, Z, y, d=d, Hdiag=H, q_mode="diag_batched")
        finally:
            _set_option("kalman_mfma", 2)  # (the default since ABI 9)
        assert np.all(st1 == 0) and np.all(st0 == 0), (m, ns)
        assert_allclose(logp1, logp0, rtol=1e-11, err_msg=str((m, k, p, ns)))
        for i in (0, 4):
            ref = oracle.kalman_filter_logp(y, T[i], R[i], np.diag(q[i]), Z, H=np.diag(H), d=d)
            assert_allclose(logp1[i], ref, rtol=LOGP_RTOL)


@pytest.mark.parametrize("m,k,p,ns,states_only", [(40, 7, 7, 18, True), (40, 7, 7, 18, False), (40, 7, 3, 17, False),
                                                   (30, 5, 8, 20, True), (24, 4, 2, 19, False), (28, 6, 6, 17, True),
                                                   (40, 7, 7, 16, True), (40, 7, 7, 21, True),
                                                   # KT = 3 and 4 (9 .. 16 state variables), with and without observed non-states
                                                   (24, 4, 3, 9, True), (24, 4, 3, 9, False), (30, 5, 4, 12, False),
                                                   (30, 5, 6, 14, True), (36, 6, 5, 16, False), (20, 3, 8, 13, False)])
def test_kalman_mf_kernel_matches_valu_kernels(m, k, p, ns, states_only):
    """kalman_mf_kernel<KT, TM> (round 6: the covariance in the tile layout of the 4 x 4 x 4 FP64 matrix instruction; default for 9 ..
    20 state variables: KT = 3, 4, 5 tiles, TM = KT or KT + 2) against the VALU kernels (dsge_options.kalman_mfma = 0) and the oracle: selector values other than 1,
    d != 0, NaN and fill-marker missing data, full recursion and steady-state switch, observed non-states.  With Z on arbitrary
    variables the retained set exceeds 20 for some sizes: those draws are handed on to the VALU cascade by the kernel itself, and
    the hint sizes 16 / 21 never reach it -- the results must not care."""
    nb, T_len = 6, 60
    T, R, q, Z, d, H, y = _kalman_inputs(nb, m, k, p, T_len, ns, seed=7000 + 31 * m + ns + p)
    if states_only:
        cols = np.flatnonzero(np.any(T[0] != 0.0, axis=0))
        rng = np.random.default_rng(7 + m + p)
        Z = np.zeros((p, m))
        Z[np.arange(p), rng.choice(cols, p, replace=False)] = rng.choice([1.0, 0.25, -2.0], p)
    for stol in (None, 0.0):
        o_mf = {} if stol is None else {"kalman_steady_tol": stol}
        o_va = {**o_mf, "kalman_mfma": 0}
        lp_mf, st_mf = batched.kalman_logp_batched(T, R, q, Z, y, d=d, Hdiag=H, q_mode="diag_batched", options=o_mf)
        lp_va, st_va = batched.kalman_logp_batched(T, R, q, Z, y, d=d, Hdiag=H, q_mode="diag_batched", options=o_va)
        assert np.all(st_mf == 0) and np.all(st_va == 0), (st_mf, st_va)
        assert_allclose(lp_mf, lp_va, rtol=1e-11)
        for i in (0, nb - 1):
            ref = oracle.kalman_filter_logp(y, T[i], R[i], np.diag(q[i]), Z, H=np.diag(H), d=d)
            assert_allclose(lp_mf[i], ref, rtol=LOGP_RTOL)
    # a stationary covariance handed in (P0 given: no doubling in the kernel) -- the standalone filter entry point computes it itself;
    # the fused evaluation with both solvers runs the kernel behind the solver (R folded in): tests/test_gpu_conventions.py


def test_gensys_window_path_matches_single_launch(ref_goldens, failure_golden):
    """gensys as three launches on the active window (dsge_gensys_win.hpp) vs the single-launch kernel: same eu and status
    everywhere, T to 1e-10 on the goldens, the failure systems (non-unique / no solution / coincident zeros: the
    reference still returns G1 there) and SW-shaped draws; both against the oracle."""
    lib = _lib.load()
    sets = []
    for key in ("one_block", "rbc_2_block", "full_nk"):
        sets.append(tuple(ref_goldens[f"{key}_{x}"][None] for x in "ABCD"))
    names = ["ok", "nonunique", "noexist", "coincident"]
    sets.append(tuple(np.stack([failure_golden[f"{nm}_{x}"] for nm in names]) for x in "ABCD"))
    b = wl.sw_shaped_batch(96)
    sets.append(tuple(b[x] for x in "ABCD"))
    # a mixed batch: structures differ per draw (one draw with a dense A => z = 0 for the caps)
    rng = np.random.default_rng(5)
    A2, B2, C2, D2 = (b[x][:6].copy() for x in "ABCD")
    A2[3] += 1e-3 * rng.standard_normal(A2[3].shape)
    sets.append((A2, B2, C2, D2))
    for A, B, C, D in sets:
        _set_option("gensys_split", 2)  # also for the small goldens (auto mode keeps them on one launch)
        try:
            out1 = batched.gensys_batched(A, B, C, D, tol=1e-8)
        finally:
            _set_option("gensys_split", 1)
        _set_option("gensys_split", 0)
        try:
            out0 = batched.gensys_batched(A, B, C, D, tol=1e-8)
        finally:
            _set_option("gensys_split", 1)
        assert np.array_equal(out1["eu"], out0["eu"]), (out1["eu"], out0["eu"])
        assert np.array_equal(out1["status"], out0["status"])
        scale = max(1.0, np.abs(out0["T"]).max())
        assert_allclose(out1["T"], out0["T"], atol=1e-10 * scale)
        for i in range(min(A.shape[0], 4)):
            T_ref, ok, _eu = oracle.gensys_T_success(A[i], B[i], C[i], D[i], tol=1e-8)
            if out1["eu"][i][0] > -2:
                assert_allclose(out1["T"][i], T_ref, atol=1e-8)
            assert bool(ok) == bool(out1["status"][i] == 0)


def test_bk_eigenvalues_batched():
    """dsge_bk_eigenvalues_batched (reduce + QZ launches of the window path) vs the reference's compute_bk_eigenvalues
    (golden, tests/golden/make_bk_golden.py) and the oracle: moduli to 1e-7 relative (finite roots; the 1/tol-sized
    'infinite' ones only in magnitude), counts exact, BK verdict = the eu verdict of gensys on the same systems."""
    import os

    from geconpy_amd.diagnostics import check_bk_condition_batched

    here = os.path.dirname(os.path.abspath(__file__))
    g = np.load(os.path.join(here, "golden", "bk_eigenvalues.npz"))
    rg = np.load(os.path.join(here, "golden", "reference_goldens.npz"))
    fg = np.load(os.path.join(here, "golden", "failure_cases.npz"))
    groups = [[(k, tuple(rg[f"{k}_{x}"] for x in "ABCD"))] for k in ("one_block", "rbc_2_block", "full_nk")]
    b = wl.sw_shaped_batch(2)
    groups.append([(k, tuple(fg[f"{k}_{x}"] for x in "ABCD")) for k in ("ok", "nonunique", "noexist")]
                  + [(f"sw{i}", tuple(b[x][i] for x in "ABCD")) for i in range(2)])
    for grp in groups:
        A, B, C, D = (np.stack([c[1][j] for c in grp]) for j in range(4))
        out = batched.bk_eigenvalues_batched(A, B, C, tol=1e-8)
        assert np.all(out["status"] == 0)
        for i, (name, _) in enumerate(grp):
            m = int(out["n_eig"][i])
            ref_mod = np.hypot(g[f"{name}_real"], g[f"{name}_imag"])
            assert m == ref_mod.size and int(out["n_forward"][i]) == int(g[f"{name}_n_forward"])
            mod = np.hypot(out["real"][i, :m], out["imag"][i, :m])
            assert np.all(np.diff(mod) >= 0)
            finite = ref_mod < 1e4
            assert_allclose(mod[finite], ref_mod[finite], rtol=1e-7, atol=1e-10)
            # infinite roots (alpha = 0): beta / tol depends on the scaling of the Schur form, only the size is meaningful
            assert np.all(mod[~finite] > 1e6)
            assert int(out["n_unstable"][i]) == int((ref_mod > 1).sum())
            # complex eigenvalues of a real pencil come in conjugate pairs: compare as multisets of (re, |im|)
            ref_pairs = np.sort_complex(g[f"{name}_real"][finite] + 1j * np.abs(g[f"{name}_imag"][finite]))
            dev_pairs = np.sort_complex(out["real"][i, :m][finite] + 1j * np.abs(out["imag"][i, :m][finite]))
            assert_allclose(dev_pairs, ref_pairs, rtol=1e-6, atol=1e-8)
        ok = check_bk_condition_batched(A, B, C, D, return_value="bool")
        eu = batched.gensys_batched(A, B, C, D, tol=1e-8)["eu"]
        assert np.array_equal(ok, (eu[:, 0] == 1) & (eu[:, 1] == 1))
        frames = check_bk_condition_batched(A, B, C, D, return_value="dataframe")
        assert list(frames[0].columns) == ["Modulus", "Real", "Imaginary"] and len(frames[0]) == int(out["n_eig"][0])


@pytest.mark.parametrize("solver", ["cycle_reduction", "gensys"])
def test_full_nk_perturbed_workload(solver):
    """SURVEY 8d sanity configuration: the reference's full_nk golden system (N = 38) under seeded 1e-3 relative
    perturbations; fused device evaluation vs the oracle on every draw, both solvers."""
    shard, om = wl.full_nk_batch(12, first_draw=5, T_len=80)
    q = shard["sigma"] ** 2
    out = batched.solve_kalman_logp_batched(shard["A"], shard["B"], shard["C"], shard["D"], q, om["Z"], om["y"],
                                            Hdiag=om["Hdiag"], solver=solver, tol=1e-8, max_iter=1000)
    n_ok = 0
    for i in range(12):
        r = oracle.solve_kalman_logp(shard["A"][i], shard["B"][i], shard["C"][i], shard["D"][i], np.diag(q[i]), om["Z"],
                                     om["y"], H=np.diag(om["Hdiag"]), solver=solver, tol=1e-8, max_iter=1000)
        assert bool(r["success"]) == bool(out["status"][i] == 0)
        if r["success"]:
            assert_allclose(out["logp"][i], r["logp"], rtol=LOGP_RTOL)
            n_ok += 1
        else:
            assert out["logp"][i] == -np.inf
    assert n_ok >= 10


def test_fused_host_path_gensys_chunks_on_two_streams():
    """The host-pointer fused call stages the batch in chunks on two streams; with solver="gensys" each stream must own its
    window workspace (a shared one would be overwritten by the next chunk's reduce launch).  1024 draws (2 chunks in
    flight) must reproduce four independent 256-draw calls (one chunk each) bit for bit."""
    om = wl.sw_shaped_observation_model()
    b = wl.sw_shaped_batch(1024)
    q = b["sigma"] ** 2
    y = om["y"][:30]
    kw = dict(Hdiag=om["Hdiag"], solver="gensys", tol=1e-8)
    big = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], y, **kw)
    for c in range(4):
        sl = slice(256 * c, 256 * (c + 1))
        part = batched.solve_kalman_logp_batched(b["A"][sl], b["B"][sl], b["C"][sl], b["D"][sl], q[sl], om["Z"], y, **kw)
        assert np.array_equal(part["status"], big["status"][sl])
        assert np.array_equal(part["logp"], big["logp"][sl])
    assert np.all(big["status"] == 0)


def test_gensys_window_path_fuzz():
    """Window path over random sizes and structures (number of state variables, of forward-looking variables, of
    non-state variables; state columns scattered instead of leading; explosive variants with too few / too many stable
    roots): eu and the success flag exact against the oracle (LAPACK), T to 1e-8 wherever the reference defines it; the
    single-launch kernel must agree where it fits."""
    lib = _lib.load()
    rng = np.random.default_rng(2026)
    shapes = [(6, 2, 1), (9, 4, 3), (13, 5, 2), (20, 9, 7), (27, 12, 6), (33, 14, 11), (40, 18, 12), (47, 20, 14), (31, 31, 9)]
    for n, ns, nl in shapes:
        k = min(3, n // 2)
        sysl = []
        for j in range(4):
            A, B, C, D, _ = wl.sw_shaped_system(7000 + 13 * n + j, n=n, n_state=ns, n_lead=nl, k=k)
            if j == 1:  # scatter the variables: a random symmetric permutation of variables (rows stay equations)
                perm = rng.permutation(n)
                A, B, C = A[:, perm], B[:, perm], C[:, perm]
            if j == 2:  # more unstable roots than forward-looking variables (no stable solution)
                A = 2.5 * A
            if j == 3:  # fewer unstable roots than forward-looking variables (indeterminacy: eu = [1, 0, k] on most sizes)
                C = 6.0 * C
            sysl.append((A, B, C, D))
        A, B, C, D = (np.stack([s_[i] for s_ in sysl]) for i in range(4))
        _set_option("gensys_split", 2)
        try:
            out = batched.gensys_batched(A, B, C, D, tol=1e-8)
        finally:
            _set_option("gensys_split", 1)
        single = None
        _set_option("gensys_split", 0)
        try:
            single = batched.gensys_batched(A, B, C, D, tol=1e-8)
        except _lib.DsgeHipError:
            single = None  # the single-launch kernel does not fit this pencil into 160 KB of LDS
        finally:
            _set_option("gensys_split", 1)
        for i in range(4):
            T_ref, ok, eu_ref = oracle.gensys_T_success(A[i], B[i], C[i], D[i], tol=1e-8)
            assert list(out["eu"][i]) == [int(x) for x in eu_ref], (n, ns, nl, i, out["eu"][i], eu_ref)
            assert bool(out["status"][i] == 0) == bool(ok)
            scale = max(1.0, np.abs(T_ref).max())
            if eu_ref[0] > -2:
                assert_allclose(out["T"][i], T_ref, atol=1e-8 * scale, err_msg=str((n, ns, nl, i)))
            if single is not None:
                assert np.array_equal(single["eu"][i], out["eu"][i])
                assert_allclose(single["T"][i], out["T"][i], atol=1e-8 * scale)


@pytest.mark.parametrize("split", [2, 0])
def test_gensys_nan_inf_inputs_are_flagged_not_hung(split):
    """NaN / Inf entries in one draw's Jacobians: the draw is flagged (success False, logp = -inf in the fused call), the
    launch terminates (every loop of the QZ / Jacobi kernels is bounded) and the neighbouring draws are untouched --
    window path and single-launch kernel."""
    lib = _lib.load()
    b = wl.sw_shaped_batch(6)
    A, B, C, D = (b[x].copy() for x in "ABCD")
    A[1, 3, 2] = np.nan
    B[3, 0, 0] = np.inf
    C[4, 5, 39] = np.nan
    clean = batched.gensys_batched(b["A"], b["B"], b["C"], b["D"], tol=1e-8)
    _set_option("gensys_split", split)
    try:
        out = batched.gensys_batched(A, B, C, D, tol=1e-8)
        om = wl.sw_shaped_observation_model()
        fused = batched.solve_kalman_logp_batched(A, B, C, D, b["sigma"] ** 2, om["Z"], om["y"][:40], Hdiag=om["Hdiag"],
                                                  solver="gensys", tol=1e-8)
    finally:
        _set_option("gensys_split", 1)
    for i in (1, 3):
        assert out["status"][i] != 0 and not out["success"][i] and np.all(out["T"][i] == 0)
    # a NaN in a column of C: |C|.sum(0) > tol is False for that column, so it is not a forward-looking variable for
    # _gensys_setup (gensys.py:580-589) and never enters the pencil -- the reference solves the reduced system and so
    # does the device (same eu, same T); the fused evaluation then fails at R = -(C T + B)^-1 D
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        T_ref, ok_ref, eu_ref = oracle.gensys_T_success(A[4], B[4], C[4], D[4], tol=1e-8)
    assert list(out["eu"][4]) == [int(x) for x in eu_ref] and bool(out["success"][4]) == bool(ok_ref)
    assert_allclose(out["T"][4], T_ref, atol=1e-8)
    for i in (1, 3, 4):
        assert fused["status"][i] != 0 and fused["logp"][i] == -np.inf
    for i in (0, 2, 5):
        assert out["status"][i] == 0
        assert_allclose(out["T"][i], clean["T"][i], atol=1e-10)
        assert np.isfinite(fused["logp"][i]) and fused["status"][i] == 0


def test_kalman_block_steady_matches_step_by_step():
    """Tail hand-off (kalman_tail_kernel: steady-state steps two at a time, one matrix-vector product per pair, R'R = F^-1,
    rows in registers; experimental, off by default) vs the step-by-step register loop: same status, logp to 1e-12; with missing-data patterns that interrupt the pairs (a mask
    change at an odd and at an even offset, a fully missing step, a series that stops), an odd remaining step count, and
    against the oracle."""
    lib = _lib.load()
    om = wl.sw_shaped_observation_model()
    b = wl.sw_shaped_batch(24)
    q = b["sigma"] ** 2
    ys = []
    y0 = om["y"].copy()
    ys.append(y0)
    y1 = om["y"][:151].copy()            # odd length
    y1[90, 1] = np.nan                   # single missing entry at an even offset
    y1[121, 3] = np.nan                  # ... and at an odd one
    y1[130, :] = np.nan                  # fully missing step
    ys.append(y1)
    y2 = om["y"].copy()
    y2[100:, 2] = np.nan                 # a series that stops: second steady segment with another mask
    y2[60:64, 0] = oracle.MISSING_FILL   # fill-value run
    ys.append(y2)
    for y in ys:
        out0 = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], y, Hdiag=om["Hdiag"], tol=1e-8,
                                                 max_iter=1000)
        _set_option("kalman_block", 1)
        try:
            out1 = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], y, Hdiag=om["Hdiag"],
                                                     tol=1e-8, max_iter=1000)
        finally:
            _set_option("kalman_block", 0)
        assert np.array_equal(out1["status"], out0["status"]) and np.all(out1["status"] == 0)
        assert_allclose(out1["logp"], out0["logp"], rtol=1e-12)
        for i in (0, 7, 23):
            r = oracle.solve_kalman_logp(b["A"][i], b["B"][i], b["C"][i], b["D"][i], np.diag(q[i]), om["Z"], y,
                                         H=np.diag(om["Hdiag"]), tol=1e-8, max_iter=1000)
            assert_allclose(out1["logp"][i], r["logp"], rtol=LOGP_RTOL)


def test_dispatch_order_and_chunks_do_not_change_results():
    """Scheduling switches of the fused device call -- Kalman workgroups dispatched in descending order of the cycle-
    reduction iteration counts (default on), chunks on several streams (default off) -- must leave logp / status of every
    draw bit-identical (each draw writes its own outputs), including failed draws and a batch that is not a multiple of
    the chunk size."""
    import torch

    from geconpy_amd.engine import LogpEngine

    lib = _lib.load()
    nb = 1500
    b = wl.sw_shaped_batch(nb)
    om = wl.sw_shaped_observation_model()
    A = b["A"].copy()
    A[77, 0, 0] = np.nan
    eng = LogpEngine(torch.device("cuda", 0))
    dA, dB, dC, dD = (eng.to_device(x) for x in (A, b["B"], b["C"], b["D"]))
    dq = eng.to_device(b["sigma"] ** 2)
    dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"][:50]), eng.to_device(om["Hdiag"])
    hints = eng.structure_hints(dA, dZ)

    def run():
        lp, st = eng.solve_kalman_logp(dA, dB, dC, dD, dq, dZ, dy, Hdiag=dH, tol=1e-8, max_iter=1000,
                                       n_state_hint=hints[0], z_selector_hint=hints[1])
        torch.cuda.synchronize()
        return lp.cpu().numpy(), st.cpu().numpy()

    ref_lp, ref_st = run()
    assert ref_st[77] != 0 and ref_lp[77] == -np.inf and np.count_nonzero(ref_st) == 1
    try:
        for order, chunks in ((0, 0), (1, 3), (0, 4), (2, 2), (2, 0)):
            _set_option("kalman_order", order)
            _set_option("pipeline_chunks", chunks)
            lp, st = run()
            assert np.array_equal(st, ref_st) and np.array_equal(lp, ref_lp), (order, chunks)
    finally:
        _set_option("kalman_order", 1)
        _set_option("pipeline_chunks", 0)


def test_fused_calls_on_two_streams_do_not_share_scratch():
    """The library is re-entrant per stream: fused evaluations enqueued back to back on two torch streams --
    different batches, cycle reduction on one and gensys on the other -- keep their intermediates (T, R, RQR, P0, the gensys
    window workspace, the dispatch order) in per-stream arenas and reproduce the results of running them one after the other."""
    import torch

    from geconpy_amd.engine import LogpEngine

    om = wl.sw_shaped_observation_model()
    eng = LogpEngine(torch.device("cuda", 0))
    dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"][:60]), eng.to_device(om["Hdiag"])
    jobs = []
    for first, nb, solver in ((0, 1536, "cycle_reduction"), (3000, 1024, "gensys"), (5000, 768, "cycle_reduction")):
        b = wl.sw_shaped_batch(nb, first_draw=first)
        t = tuple(eng.to_device(b[x]) for x in "ABCD") + (eng.to_device(b["sigma"] ** 2),)
        jobs.append((t, solver))

    def run(job):
        (dA, dB, dC, dD, dq), solver = job
        hints = eng.structure_hints(dA, dZ)
        return eng.solve_kalman_logp(dA, dB, dC, dD, dq, dZ, dy, Hdiag=dH, solver=solver, tol=1e-8, max_iter=1000,
                                     n_state_hint=hints[0], z_selector_hint=hints[1])

    ref = []
    for job in jobs:
        lp, st = run(job)
        torch.cuda.synchronize()
        ref.append((lp.cpu().numpy().copy(), st.cpu().numpy().copy()))
    streams = [torch.cuda.Stream() for _ in jobs]
    outs = [None] * len(jobs)
    for rep in range(3):  # enqueue everything before anything is waited for
        for i, job in enumerate(jobs):
            with torch.cuda.stream(streams[i]):
                outs[i] = run(job)
    torch.cuda.synchronize()
    for (lp, st), (rlp, rst) in zip(outs, ref):
        assert np.array_equal(st.cpu().numpy(), rst) and np.array_equal(lp.cpu().numpy(), rlp)


@pytest.mark.parametrize("n,ns,nl", [(50, 22, 15), (56, 25, 16)])
def test_fused_pipeline_seven_wide_tile(n, ns, nl):
    """n = 49..56 runs on the 7 x 7 register-block instances (56-wide tile) of every kernel of the fused call instead of the
    spilling 64-wide ones: logp against the oracle, T against the construction."""
    k = p = 7
    nb = 4
    sysm = [wl.sw_shaped_system(4200 + 7 * n + i, n=n, n_state=ns, n_lead=nl, k=k) for i in range(nb)]
    A, B, C, D, Tst = (np.stack([s_[j] for s_ in sysm]) for j in range(5))
    q = np.full((nb, k), 1e-4)
    Z = np.zeros((p, n))
    Z[np.arange(p), np.arange(p)] = 1.0
    y = np.random.default_rng(n).normal(0, 0.02, (40, p))
    H = np.full(p, 1e-4)
    out = batched.solve_kalman_logp_batched(A, B, C, D, q, Z, y, Hdiag=H, tol=1e-9, max_iter=1000, return_policy=True)
    assert np.all(out["status"] == 0)
    assert_allclose(out["T"], Tst, atol=1e-8)
    for i in range(nb):
        r = oracle.solve_kalman_logp(A[i], B[i], C[i], D[i], np.diag(q[i]), Z, y, H=np.diag(H), tol=1e-9, max_iter=1000)
        assert_allclose(out["logp"][i], r["logp"], rtol=LOGP_RTOL)


def test_fused_call_is_independent_of_batch_size():
    """logp / status of a draw must not depend on how many other draws share the launch: batch sizes around the thresholds
    of the dispatch order (512) and of the workgroup loops (1, 63, 64, 65), cut from one 600-draw set, against the
    full-set call -- bit for bit, both solvers."""
    import torch

    from geconpy_amd.engine import LogpEngine

    nb = 600
    b = wl.sw_shaped_batch(nb, first_draw=7000)
    om = wl.sw_shaped_observation_model()
    eng = LogpEngine(torch.device("cuda", 0))
    dev = {x: eng.to_device(b[x]) for x in "ABCD"}
    dq = eng.to_device(b["sigma"] ** 2)
    dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"][:48]), eng.to_device(om["Hdiag"])
    hints = eng.structure_hints(dev["A"], dZ)
    for solver in ("cycle_reduction", "gensys"):
        def run(lo, hi):
            lp, st = eng.solve_kalman_logp(dev["A"][lo:hi], dev["B"][lo:hi], dev["C"][lo:hi], dev["D"][lo:hi], dq[lo:hi], dZ, dy,
                                           Hdiag=dH, solver=solver, tol=1e-8, max_iter=1000, n_state_hint=hints[0],
                                           z_selector_hint=hints[1])
            torch.cuda.synchronize()
            return lp.cpu().numpy(), st.cpu().numpy()

        full_lp, full_st = run(0, nb)
        assert np.all(full_st == 0)
        for lo, hi in ((0, 1), (5, 68), (100, 164), (17, 82), (0, 511), (40, 552), (87, 600)):
            lp, st = run(lo, hi)
            assert np.array_equal(st, full_st[lo:hi]) and np.array_equal(lp, full_lp[lo:hi]), (solver, lo, hi)


def _fused_policy(eng, dev, dq, dZ, dy, dH, hints, **kw):
    import torch

    nb, n = dev["A"].shape[:2]
    T = torch.full((nb, n, n), np.nan, dtype=torch.float64, device=eng.device)
    R = torch.full((nb, n, dev["D"].shape[2]), np.nan, dtype=torch.float64, device=eng.device)
    lp, st = eng.solve_kalman_logp(dev["A"], dev["B"], dev["C"], dev["D"], dq, dZ, dy, Hdiag=dH, tol=1e-9, max_iter=1000,
                                   n_state_hint=hints[0], z_selector_hint=hints[1], T_out=T, R_out=R, **kw)
    torch.cuda.synchronize()
    return lp.cpu().numpy(), st.cpu().numpy(), T.cpu().numpy(), R.cpu().numpy()


def test_cr_static_deflation_matches_full_system():
    """Cycle reduction behind the static-variable deflation (default on; 30 of 40 variables on the SW-shaped draws) against
    the full-size iteration: T, R to 1e-10, logp to the parity tolerance, status identical -- with a failed draw (NaN), a
    draw with FEWER static variables than the measured bound (a lag coefficient on a static variable: solved by the
    full-size kernels) and one with more (an extra zero column: only the first h are deflated)."""
    import torch

    from geconpy_amd.engine import LogpEngine

    lib = _lib.load()
    nb = 700
    b = wl.sw_shaped_batch(nb, first_draw=9100)
    om = wl.sw_shaped_observation_model()
    A, B, C, D = (b[x].copy() for x in "ABCD")
    static = np.where(~(A[0] != 0).any(0) & ~(C[0] != 0).any(0))[0]
    assert len(static) == 10
    A[5, 0, 0] = np.nan
    A[9, 3, static[2]] = 1e-3   # static variable turned (weakly) predetermined
    A[300, 7, static[-1]] = -2e-3
    states = np.where((A[0] != 0).any(0) & ~(C[0] != 0).any(0))[0]
    A[11, :, states[0]] = 0.0   # one more static variable than the bound (a state that stops feeding back)
    A[12, :, states[-1]] = 0.0
    eng = LogpEngine(torch.device("cuda", 0))
    dev = {x: eng.to_device(v) for x, v in zip("ABCD", (A, B, C, D))}
    dq = eng.to_device(b["sigma"] ** 2)
    dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"][:40]), eng.to_device(om["Hdiag"])
    hints = eng.structure_hints(dev["A"], dZ)
    try:
        _set_cr_deflation(0)
        lp0, st0, T0, R0 = _fused_policy(eng, dev, dq, dZ, dy, dH, hints)
        _set_cr_deflation(1)
        # the bound is measured on the first batch of a model size: give it the clean one, so that draws 9 and 300 of
        # the second call are the violators
        clean = {x: eng.to_device(b[x]) for x in "ABCD"}
        _fused_policy(eng, clean, dq, dZ, dy, dH, hints)
        lp1, st1, T1, R1 = _fused_policy(eng, dev, dq, dZ, dy, dH, hints)
        lp2, st2, T2, R2 = _fused_policy(eng, dev, dq, dZ, dy, dH, hints)
    finally:
        _set_cr_deflation(1)
    assert np.array_equal(st0, st1) and st0[5] != 0 and np.count_nonzero(st0[[0, 9, 300, 699]]) == 0
    ok = st0 == 0
    assert ok.sum() >= nb - 3
    assert_allclose(T1[ok], T0[ok], atol=1e-10)
    assert_allclose(R1[ok], R0[ok], atol=1e-10)
    assert_allclose(lp1[ok], lp0[ok], rtol=LOGP_RTOL)
    assert lp1[5] == -np.inf
    # the violators went through the full-size kernels: bit-identical to the run without deflation
    for i in (9, 300):
        assert np.array_equal(T1[i], T0[i]) and np.array_equal(R1[i], R0[i]) and lp1[i] == lp0[i]
    assert np.array_equal(lp1, lp2) and np.array_equal(T1, T2)
    # and against the oracle
    for i in (0, 9, 300, 699):
        r = oracle.solve_kalman_logp(A[i], B[i], C[i], D[i], np.diag(b["sigma"][i] ** 2), om["Z"], om["y"][:40],
                                     H=np.diag(om["Hdiag"]), tol=1e-9, max_iter=1000)
        assert_allclose(lp1[i], r["logp"], rtol=LOGP_RTOL)
        assert_allclose(T1[i], r["T"], atol=1e-9)


@pytest.mark.parametrize("n,ns,nl", [(40, 18, 12), (24, 10, 7), (32, 14, 10), (44, 20, 13), (24, 8, 6), (32, 16, 9), (44, 16, 12),
                                     (48, 23, 15), (50, 22, 15), (56, 25, 16), (64, 28, 20)])
def test_cr_fused_deflation_equals_three_launches(n, ns, nl):
    """The one-launch form of the deflated cycle reduction (default; dsge_cr_fused.hpp: the reduced system goes from the QR
    to the iteration to the back-substitution through LDS) runs the arithmetic of the three launches: T, R, logp, status
    bit for bit -- SW-shaped systems of 40 -> 30 variables (tiles 5 / 4), 24 -> 17 (3 / 3), 32 -> 24 (4 / 3), 44 -> 33
    (6 / 5), 24 -> 14 (3 / 2), 32 -> 25 (4 / 4) and 44 -> 28 (6 / 4, the cap of 16 deflated variables), and with three columns per lane in the QR 48 -> 38 (6 / 5),
    50 -> 37 (7 / 5), 56 -> 41 (7 / 6), 64 -> 48 (8 / 6): every instance of the kernel, each batch with a NaN draw and a draw with fewer static variables than the bound."""
    import torch

    from geconpy_amd.engine import LogpEngine

    k = p = 7
    nb = 200
    sysm = [wl.sw_shaped_system(8800 + 11 * n + i, n=n, n_state=ns, n_lead=nl, k=k) for i in range(nb)]
    A, B, C, D, Tst = (np.stack([s_[j] for s_ in sysm]) for j in range(5))
    static = np.where(~(A[0] != 0).any(0) & ~(C[0] != 0).any(0))[0]
    assert len(static) == n - ns - nl
    A[7, 0, 0] = np.nan
    A[21, 2, static[1]] = 1e-3  # one static variable less than the bound: handed to the full-size kernel
    q = np.full((nb, k), 1e-4)
    Z = np.zeros((p, n))
    Z[np.arange(p), np.arange(p)] = 1.0
    y = np.random.default_rng(n).normal(0, 0.02, (30, p))
    H = np.full(p, 1e-4)
    eng = LogpEngine(torch.device("cuda", 0))
    dev = {x: eng.to_device(v) for x, v in zip("ABCD", (A, B, C, D))}
    dq, dZ, dy, dH = eng.to_device(q), eng.to_device(Z), eng.to_device(y), eng.to_device(H)
    hints = eng.structure_hints(dev["A"], dZ)
    res = {}
    for fused in (0, 1):
        with _lib.options_scope({"cr_fused_deflation": fused, "n_static_hint": len(static)}):
            res[fused] = _fused_policy(eng, dev, dq, dZ, dy, dH, hints)
    for a, c in zip(res[0], res[1]):
        assert np.array_equal(a, c, equal_nan=True)
    lp, st, T, R = res[1]
    assert st[7] != 0 and np.count_nonzero(st) == 1
    ok = st == 0
    ok[21] = False  # (its A was changed: the generator's T is no longer the solution)
    assert_allclose(T[ok], Tst[ok], atol=1e-8)
    r = oracle.solve_kalman_logp(A[3], B[3], C[3], D[3], np.diag(q[3]), Z, y, H=np.diag(H), tol=1e-9, max_iter=1000)
    assert_allclose(lp[3], r["logp"], rtol=LOGP_RTOL)
    assert_allclose(T[3], r["T"], atol=1e-9)


def test_cr_fused_deflation_hands_over_what_does_not_fit():
    """16 shocks on the 40-variable system: the right-hand side [A_dy[:,S] | D_red] of the final solve (18 + 16 columns) does
    not fit the 32-wide reduced tile, so the one-launch kernel flags every draw and the full-size dense kernel solves them --
    same T, R, logp as the three-launch path (which eliminates D separately) to 1e-10, status clear."""
    import torch

    from geconpy_amd.engine import LogpEngine

    n, ns, nl, k, p, nb = 40, 18, 12, 16, 7, 24
    sysm = [wl.sw_shaped_system(9900 + i, n=n, n_state=ns, n_lead=nl, k=k) for i in range(nb)]
    A, B, C, D, Tst = (np.stack([s_[j] for s_ in sysm]) for j in range(5))
    q = np.full((nb, k), 1e-4)
    Z = np.zeros((p, n))
    Z[np.arange(p), np.arange(p)] = 1.0
    y = np.random.default_rng(3).normal(0, 0.02, (30, p))
    H = np.full(p, 1e-4)
    eng = LogpEngine(torch.device("cuda", 0))
    dev = {x: eng.to_device(v) for x, v in zip("ABCD", (A, B, C, D))}
    dq, dZ, dy, dH = eng.to_device(q), eng.to_device(Z), eng.to_device(y), eng.to_device(H)
    hints = eng.structure_hints(dev["A"], dZ)
    res = {}
    for fused in (0, 1):
        with _lib.options_scope({"cr_fused_deflation": fused, "n_static_hint": n - ns - nl}):
            res[fused] = _fused_policy(eng, dev, dq, dZ, dy, dH, hints)
    lp0, st0, T0, R0 = res[0]
    lp1, st1, T1, R1 = res[1]
    assert np.all(st0 == 0) and np.all(st1 == 0)
    assert_allclose(T1, Tst, atol=1e-8)
    assert_allclose(T1, T0, atol=1e-10)
    assert_allclose(R1, R0, atol=1e-10)
    assert_allclose(lp1, lp0, rtol=LOGP_RTOL)


@pytest.mark.parametrize("n,ns,nl,k,p", [(15, 1, 3, 11, 8), (14, 4, 2, 10, 5), (24, 3, 2, 17, 4)])
def test_more_shocks_than_the_reduced_tile_is_wide(n, ns, nl, k, p):
    """Mostly static systems with many shocks (found by tools/fuzz_fused.py): the deflation would leave fewer dynamic
    variables than there are shocks, and D_red no longer fits one column group of the reduced tile -- the launcher must
    fall back to the full-size solve (round 1 silently dropped the columns of D beyond the tile: wrong R, wrong logp)."""
    nb = 3
    sysm = [wl.sw_shaped_system(3100 + 7 * n + i, n=n, n_state=ns, n_lead=nl, k=k) for i in range(nb)]
    A, B, C, D, Tst = (np.stack([s_[j] for s_ in sysm]) for j in range(5))
    q = np.full((nb, k), 1e-4)
    Z = np.zeros((p, n))
    Z[np.arange(p), np.arange(p)] = 1.0
    y = np.random.default_rng(n).normal(0, 0.02, (12, p))
    H = np.full(p, 1e-4)
    out = batched.solve_kalman_logp_batched(A, B, C, D, q, Z, y, Hdiag=H, tol=1e-10, max_iter=1000, q_mode="diag_batched")
    assert np.all(out["status"] == 0)
    for i in range(nb):
        r = oracle.solve_kalman_logp(A[i], B[i], C[i], D[i], np.diag(q[i]), Z, y, H=np.diag(H), tol=1e-10, max_iter=1000)
        assert_allclose(out["logp"][i], r["logp"], rtol=LOGP_RTOL)


@pytest.mark.parametrize("key", ["one_block", "rbc_2_block", "full_nk"])
def test_cr_static_deflation_on_reference_goldens(ref_goldens, key):
    """The reference's own models (3 of 9, 6 of 12, 4 of 24 static variables): fused call with the deflation against the
    reference's cycle-reduction T and the oracle's logp."""
    A, B, C, D = (ref_goldens[f"{key}_{x}"] for x in "ABCD")
    n, k = A.shape[0], D.shape[1]
    nb = 3
    rng = np.random.default_rng(11)
    Ab, Bb, Cb, Db = (np.repeat(x[None], nb, 0).copy() for x in (A, B, C, D))
    Bb[1:] *= 1.0 + 1e-3 * rng.standard_normal((nb - 1, 1, 1))
    p = min(k, 2)
    Z = np.zeros((p, n))
    Z[np.arange(p), np.arange(p) + 1] = 1.0
    y = rng.normal(0, 0.05, (30, p))
    q = np.full((nb, k), 0.01)
    H = np.full(p, 1e-3)
    lib = _lib.load()
    outs = []
    try:
        for on in (0, 1):
            _set_cr_deflation(on)
            outs.append(batched.solve_kalman_logp_batched(Ab, Bb, Cb, Db, q, Z, y, Hdiag=H, tol=1e-10, max_iter=1000))
    finally:
        _set_cr_deflation(1)
    assert np.all(outs[0]["status"] == 0) and np.all(outs[1]["status"] == 0)
    assert_allclose(outs[1]["logp"], outs[0]["logp"], rtol=LOGP_RTOL)
    for i in range(nb):
        r = oracle.solve_kalman_logp(Ab[i], Bb[i], Cb[i], Db[i], np.diag(q[i]), Z, y, H=np.diag(H), tol=1e-10, max_iter=1000)
        assert_allclose(outs[1]["logp"][i], r["logp"], rtol=LOGP_RTOL)


@pytest.mark.parametrize("n,ns,nl", [(16, 6, 5), (24, 10, 7), (50, 22, 15), (56, 25, 16), (64, 28, 20)])
def test_cr_static_deflation_other_sizes(n, ns, nl):
    """Deflation across tile sizes: systems with more than 128 columns in [B_st | B_dy | A_dy | C_dy | D] (n >= 46) take the
    second column chunk of the deflation kernel (reflectors read back from LDS); n = 64 has 16 static variables, the cap."""
    import torch

    from geconpy_amd.engine import LogpEngine

    k = p = 7
    nb = 6
    sysm = [wl.sw_shaped_system(5200 + 7 * n + i, n=n, n_state=ns, n_lead=nl, k=k) for i in range(nb)]
    A, B, C, D, Tst = (np.stack([s_[j] for s_ in sysm]) for j in range(5))
    q = np.full((nb, k), 1e-4)
    Z = np.zeros((p, n))
    Z[np.arange(p), np.arange(p)] = 1.0
    y = np.random.default_rng(n).normal(0, 0.02, (30, p))
    H = np.full(p, 1e-4)
    lib = _lib.load()
    eng = LogpEngine(torch.device("cuda", 0))
    dev = {x: eng.to_device(v) for x, v in zip("ABCD", (A, B, C, D))}
    dq, dZ, dy, dH = eng.to_device(q), eng.to_device(Z), eng.to_device(y), eng.to_device(H)
    hints = eng.structure_hints(dev["A"], dZ)
    try:
        _set_cr_deflation(0)
        lp0, st0, T0, R0 = _fused_policy(eng, dev, dq, dZ, dy, dH, hints)
        _set_cr_deflation(1)
        lp1, st1, T1, R1 = _fused_policy(eng, dev, dq, dZ, dy, dH, hints)
    finally:
        _set_cr_deflation(1)
    assert np.all(st0 == 0) and np.all(st1 == 0)
    assert_allclose(T1, Tst, atol=1e-8)
    assert_allclose(T1, T0, atol=1e-9)
    assert_allclose(R1, R0, atol=1e-9)
    assert_allclose(lp1, lp0, rtol=LOGP_RTOL)
    r = oracle.solve_kalman_logp(A[0], B[0], C[0], D[0], np.diag(q[0]), Z, y, H=np.diag(H), tol=1e-9, max_iter=1000)
    assert_allclose(lp1[0], r["logp"], rtol=LOGP_RTOL)


def test_cr_static_deflation_bound_corrects_itself():
    """The bound h is measured on the first batch of a model size.  If later batches have FEWER static variables every draw
    is flagged and solved at full size (bit-identical to the run without deflation); the launcher measures again every 256th
    call and keeps the minimum, after which the later batches are deflated again (same T to 1e-10, no longer bit-identical)."""
    import torch

    from geconpy_amd.engine import LogpEngine

    lib = _lib.load()
    nb = 64
    b = wl.sw_shaped_batch(nb, first_draw=12000)
    om = wl.sw_shaped_observation_model()
    static = np.where(~(b["A"][0] != 0).any(0) & ~(b["C"][0] != 0).any(0))[0]
    A2 = b["A"].copy()
    A2[:, 4, static[0]] = 1e-3  # nine static variables in every draw
    eng = LogpEngine(torch.device("cuda", 0))
    clean = {x: eng.to_device(b[x]) for x in "ABCD"}
    fewer = dict(clean, A=eng.to_device(A2))
    dq = eng.to_device(b["sigma"] ** 2)
    dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"][:6]), eng.to_device(om["Hdiag"])
    hints = eng.structure_hints(fewer["A"], dZ)
    try:
        _set_cr_deflation(0)
        lp0, st0, T0, R0 = _fused_policy(eng, fewer, dq, dZ, dy, dH, hints)
        _set_cr_deflation(1)
        _fused_policy(eng, clean, dq, dZ, dy, dH, hints)  # h = 10
        lp1, st1, T1, R1 = _fused_policy(eng, fewer, dq, dZ, dy, dH, hints)
        assert np.all(st0 == 0) and np.array_equal(T1, T0) and np.array_equal(lp1, lp0)  # all flagged: full-size kernels
        for _ in range(256):
            eng.solve_kalman_logp(fewer["A"], fewer["B"], fewer["C"], fewer["D"], dq, dZ, dy, Hdiag=dH, tol=1e-9, max_iter=1000,
                                  n_state_hint=hints[0], z_selector_hint=hints[1])
        lp2, st2, T2, R2 = _fused_policy(eng, fewer, dq, dZ, dy, dH, hints)
    finally:
        _set_cr_deflation(1)
    assert np.all(st2 == 0) and not np.array_equal(T2, T0)  # h = 9 now: deflated again
    assert_allclose(T2, T0, atol=1e-10)
    assert_allclose(lp2, lp0, rtol=LOGP_RTOL)


@pytest.mark.parametrize("n,ns,nl", [(50, 22, 15), (56, 25, 16), (64, 28, 20)])
def test_cr_four_wave_kernel_matches_one_wave_kernel(n, ns, nl):
    """49..64 variables: cycle reduction on four wavefronts per draw (cr_wide_kernel, default) against the one-wavefront
    compact kernel: same iteration counts and status, T and R to 1e-11 (the norms of the stopping rule are summed in a
    different order; the rest is the same arithmetic), logp to the parity tolerance -- with a NaN draw, without the
    deflation (which would shrink the systems below 49 variables)."""
    import torch

    from geconpy_amd.engine import LogpEngine

    k = p = 7
    nb = 40
    sysm = [wl.sw_shaped_system(4300 + 13 * n + i, n=n, n_state=ns, n_lead=nl, k=k) for i in range(nb)]
    A, B, C, D, Tst = (np.stack([s_[j] for s_ in sysm]) for j in range(5))
    A[6, 1, 1] = np.nan
    T4, st4, it4 = batched.cycle_reduction_batched(A, B, C, max_iter=1000, tol=1e-9, options={"cr_four_waves": 1})
    T1, st1, it1 = batched.cycle_reduction_batched(A, B, C, max_iter=1000, tol=1e-9, options={"cr_four_waves": 0})
    assert np.array_equal(st4, st1) and st4[6] != 0 and np.count_nonzero(st4) == 1
    ok = st4 == 0
    assert np.array_equal(it4[ok], it1[ok])
    assert_allclose(T4[ok], T1[ok], atol=1e-11)
    assert_allclose(T4[ok], Tst[ok], atol=1e-8)
    q = np.full((nb, k), 1e-4)
    Z = np.zeros((p, n))
    Z[np.arange(p), np.arange(p)] = 1.0
    y = np.random.default_rng(n).normal(0, 0.02, (30, p))
    H = np.full(p, 1e-4)
    eng = LogpEngine(torch.device("cuda", 0))
    dev = {x: eng.to_device(v) for x, v in zip("ABCD", (A, B, C, D))}
    dq, dZ, dy, dH = eng.to_device(q), eng.to_device(Z), eng.to_device(y), eng.to_device(H)
    hints = eng.structure_hints(dev["A"], dZ)
    res = {}
    for four in (0, 1):
        with _lib.options_scope({"cr_four_waves": four, "cr_deflation": 0}):
            res[four] = _fused_policy(eng, dev, dq, dZ, dy, dH, hints)
    assert np.array_equal(res[0][1], res[1][1])
    assert_allclose(res[1][0][ok], res[0][0][ok], rtol=LOGP_RTOL)
    assert_allclose(res[1][3][ok], res[0][3][ok], atol=1e-10)  # R out of the final elimination
    r = oracle.solve_kalman_logp(A[3], B[3], C[3], D[3], np.diag(q[3]), Z, y, H=np.diag(H), tol=1e-9, max_iter=1000)
    assert_allclose(res[1][0][3], r["logp"], rtol=LOGP_RTOL)


def test_cr_two_wave_instance_is_bit_identical():
    """The 32-wide compact cycle-reduction kernel built for two waves per SIMD (default) runs the same arithmetic as the
    one-wave instance: T, R, logp bit for bit, on the deflated SW-shaped system (30 variables) and on a 32-variable one."""
    import torch

    from geconpy_amd.engine import LogpEngine

    lib = _lib.load()
    eng = LogpEngine(torch.device("cuda", 0))
    om = wl.sw_shaped_observation_model()
    b = wl.sw_shaped_batch(300, first_draw=15000)
    sysm = [wl.sw_shaped_system(6100 + i, n=32, n_state=14, n_lead=9, k=7) for i in range(40)]
    b32 = {x: np.stack([s_[j] for s_ in sysm]) for j, x in enumerate("ABCD")}
    for batch, Z, nb in ((b, om["Z"], 300), (b32, np.eye(7, 32), 40)):
        dev = {x: eng.to_device(batch[x]) for x in "ABCD"}
        dq = eng.to_device(np.full((nb, 7), 1e-4))
        dZ, dy, dH = eng.to_device(Z), eng.to_device(om["y"][:30]), eng.to_device(om["Hdiag"])
        hints = eng.structure_hints(dev["A"], dZ)
        try:
            _set_option("cr_two_waves", 0)
            r0 = _fused_policy(eng, dev, dq, dZ, dy, dH, hints)
            _set_option("cr_two_waves", 1)
            r1 = _fused_policy(eng, dev, dq, dZ, dy, dH, hints)
        finally:
            _set_option("cr_two_waves", 1)
        assert np.all(r0[1] == 0)
        for a0, a1 in zip(r0, r1):
            assert np.array_equal(a0, a1)


def test_cr_static_deflation_fuzz():
    """Deflation on / off over random sizes, shock counts and variable orders (the static columns scattered by a random
    permutation of the variables, so the index tables, the two-column chunks and the scatter of the inflate kernel see
    arbitrary masks): T, R to 1e-9, logp to the parity tolerance, status identical."""
    import torch

    from geconpy_amd.engine import LogpEngine

    lib = _lib.load()
    eng = LogpEngine(torch.device("cuda", 0))
    rng = np.random.default_rng(2024)
    tried = 0
    for trial in range(28):
        n = int(rng.integers(9, 49))
        ns = max(2, int(0.4 * n) + int(rng.integers(-1, 2)))
        nl = max(1, int(0.25 * n) + int(rng.integers(-1, 2)))
        k = int(rng.choice([1, 2, 3, 7, min(n // 3, 10)]))
        p = min(k, 4)
        nb = 5
        try:
            sysm = [wl.sw_shaped_system(9000 + 31 * trial + i, n=n, n_state=ns, n_lead=nl, k=k) for i in range(nb)]
        except Exception:
            continue
        perm = rng.permutation(n)
        A, B, C, D = (np.stack([s_[j] for s_ in sysm]) for j in range(4))
        A, B, C = (np.ascontiguousarray(M[:, :, perm]) for M in (A, B, C))
        n_static = int(np.count_nonzero(~((A[0] != 0).any(0) | (C[0] != 0).any(0))))
        Z = np.zeros((p, n))
        Z[np.arange(p), rng.choice(n, p, replace=False)] = 1.0
        y = rng.normal(0, 0.02, (12, p))
        dev = {x: eng.to_device(v) for x, v in zip("ABCD", (A, B, C, D))}
        dq, dZ, dy, dH = eng.to_device(np.full((nb, k), 1e-4)), eng.to_device(Z), eng.to_device(y), eng.to_device(np.full(p, 1e-4))
        hints = eng.structure_hints(dev["A"], dZ)
        try:
            _set_cr_deflation(0)
            lp0, st0, T0, R0 = _fused_policy(eng, dev, dq, dZ, dy, dH, hints)
            _set_cr_deflation(1)
            lp1, st1, T1, R1 = _fused_policy(eng, dev, dq, dZ, dy, dH, hints)
        finally:
            _set_cr_deflation(1)
        assert np.array_equal(st0, st1), (n, k, n_static)
        ok = st0 == 0
        assert ok.any(), (n, k, n_static)
        assert_allclose(T1[ok], T0[ok], atol=1e-9, err_msg=str((n, k, n_static)))
        assert_allclose(R1[ok], R0[ok], atol=1e-9, err_msg=str((n, k, n_static)))
        assert_allclose(lp1[ok], lp0[ok], rtol=LOGP_RTOL, err_msg=str((n, k, n_static)))
        tried += 1
    assert tried >= 20


@pytest.mark.parametrize("n,ns,nl,k", [(40, 18, 12, 7), (24, 10, 7, 3), (12, 5, 3, 1), (50, 22, 15, 16)])
def test_rqr_kernel_matches_assemble_kernel(n, ns, nl, k):
    """Fused call with a diagonal Q (`rqr_kernel`: sym(R diag(q) R') with one result column per lane) against the same Q
    handed over as full per-draw matrices (`assemble_kernel`): logp to 1e-12, status identical; a failed draw gives -inf
    on both routes.  Shared and per-draw q."""
    import torch

    from geconpy_amd.engine import LogpEngine

    eng = LogpEngine(torch.device("cuda", 0))
    rng = np.random.default_rng(n + k)
    nb = 9
    sysm = [wl.sw_shaped_system(7700 + 13 * n + i, n=n, n_state=ns, n_lead=nl, k=k) for i in range(nb)]
    A, B, C, D = (np.stack([s_[j] for s_ in sysm]) for j in range(4))
    A[4, 0, 0] = np.nan
    p = min(k, 5)
    Z = np.zeros((p, n))
    Z[np.arange(p), np.arange(p)] = 1.0
    y = rng.normal(0, 0.02, (25, p))
    dev = [eng.to_device(x) for x in (A, B, C, D)]
    dZ, dy, dH = eng.to_device(Z), eng.to_device(y), eng.to_device(np.full(p, 1e-4))
    hints = eng.structure_hints(dev[0], dZ)
    for q in (rng.uniform(0.5e-4, 2e-4, k), rng.uniform(0.5e-4, 2e-4, (nb, k))):
        qb = np.broadcast_to(q, (nb, k))
        Qfull = np.stack([np.diag(qb[i]) for i in range(nb)])
        outs = []
        for Q, mode in ((q, 0 if q.ndim == 1 else 1), (Qfull, 3)):
            lp, st = eng.solve_kalman_logp(*dev, eng.to_device(np.ascontiguousarray(Q)), dZ, dy, Hdiag=dH, q_mode=mode, tol=1e-9,
                                           max_iter=1000, n_state_hint=hints[0], z_selector_hint=hints[1])
            torch.cuda.synchronize()
            outs.append((lp.cpu().numpy(), st.cpu().numpy()))
        (lp_d, st_d), (lp_f, st_f) = outs
        assert np.array_equal(st_d, st_f) and st_d[4] != 0 and lp_d[4] == -np.inf and lp_f[4] == -np.inf
        ok = st_d == 0
        assert ok.sum() == nb - 1
        assert_allclose(lp_d[ok], lp_f[ok], rtol=1e-12)


def test_ill_conditioned_solves_are_refined():
    """A draw whose A1 has condition 1.2e8 in the second iteration (found by tools/fuzz_cr.py, seed 11): the blocked
    Gauss-Jordan alone ends 1.5e-7 from the oracle's T (numpy's LAPACK LU is 1e-10 from an extended-precision solution);
    with the step of iterative refinement the pivot-ratio test triggers, the one-wavefront kernels and the dense fallback
    are at LAPACK's level; the four-wavefront kernel (default above 48 variables) hands a flagged draw to its out-of-line
    refined instance (crw_solve_refined, dsge_cr_wide.hpp).  1e-9 for every kernel."""
    A, B, C = (x[None] for x in wl.sw_shaped_system(386903548, n=62, n_state=9, n_lead=3, k=1)[:3])
    Tc, conv, itc = oracle.cycle_reduction_core(A[0], B[0], C[0], 200, 1e-9)
    assert conv
    for opts, bar in (({"cr_four_waves": 0}, 1e-9), ({"cr_compact": 0}, 1e-9), ({}, 1e-9)):
        T, st, it = batched.cycle_reduction_batched(A, B, C, max_iter=200, tol=1e-9, options=opts)
        assert st[0] == 0 and it[0] == itc
        assert np.abs(T[0] - Tc).max() <= bar, (opts, np.abs(T[0] - Tc).max())


def test_system_on_which_float64_cycle_reductions_cannot_agree():
    """tests/golden/cr_ill_conditioned_54.npz (found by tools/fuzz_cr.py, seed 3101): 54 variables, cond(A1) = 2..5e6 in
    every iteration, |C| = 3e5.  The reference's float64 path (LAPACK LU, cycle_reduction.py:150-160) is 1.8e-8 from the
    40-digit T stored with the system -- 0.8..6e-8 over rounding-level perturbations of it, tools/cr_accuracy_study.py --, so
    no float64 implementation can be within 1e-9 of it by more than luck.  The refinement step of the device forms its
    residual in twice the working precision (mm_residual_dot2) and lands 7 times closer to the exact T than the reference:
    asserted here as at most HALF the reference's own distance, for the four-wavefront, one-wavefront and dense kernels."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cr_ill_conditioned_54.npz"))
    A, B, C, tol = g["A"][None], g["B"][None], g["C"][None], float(g["tol"])
    Tc, conv, itc = oracle.cycle_reduction_core(A[0], B[0], C[0], 200, tol)
    assert conv
    e_ref = np.abs(Tc - g["T_exact"]).max()
    assert 5e-9 < e_ref < 1e-7  # the reference's own float64 result is this far from exact arithmetic
    for opts in ({}, {"cr_four_waves": 0}, {"cr_compact": 0}):
        T, st, it = batched.cycle_reduction_batched(A, B, C, max_iter=200, tol=tol, options=opts)
        assert st[0] == 0 and it[0] == itc
        e_dev = np.abs(T[0] - g["T_exact"]).max()
        assert e_dev <= 0.5 * e_ref, (opts, e_dev, e_ref)


def test_kalman_filter_outputs_per_step():
    """save_kalman_filter_outputs_in_idata (statespace.py:1145, 1151-1157): per-step log-likelihood, predicted / filtered
    states and covariances from the device against the oracle's recursion, on SW-shaped draws with missing observations
    (one partial row, one empty row), dense-Z and selector-Z; the per-step ll sums to the fast kernels' logp, and the stored
    per-step ll of tests/golden/sw_shaped.npz (oracle_ll) is reproduced."""
    nb = 4
    b = wl.sw_shaped_batch(nb)
    om = wl.sw_shaped_observation_model()
    T = b["T_star"]
    R = np.stack([oracle.compute_selection_matrix(b["B"][i], b["C"][i], b["D"][i], T[i]) for i in range(nb)])
    q = b["sigma"] ** 2
    rng = np.random.default_rng(5)
    for variant in ("selector", "dense"):
        y = om["y"][:60].copy()
        y[5, 2] = np.nan
        y[11] = np.nan
        Z = om["Z"].copy()
        d = None
        if variant == "dense":
            Z = Z + 0.05 * rng.standard_normal(Z.shape) * (rng.random(Z.shape) < 0.2)
            d = rng.normal(0, 0.01, Z.shape[0])
        out = batched.kalman_filter_outputs_batched(T, R, q, Z, y, d=d, Hdiag=om["Hdiag"], full_covariances=True)
        diag = batched.kalman_filter_outputs_batched(T, R, q, Z, y, d=d, Hdiag=om["Hdiag"])
        lp, st = batched.kalman_logp_batched(T, R, q, Z, y, d=d, Hdiag=om["Hdiag"])
        assert (out["status"] == 0).all() and (st == 0).all()
        assert_allclose(out["ll"].sum(axis=1), lp, rtol=1e-10)
        for i in range(nb):
            tot, ll, stt = oracle.kalman_filter_logp(y, T[i], R[i], np.diag(q[i]), Z, H=np.diag(om["Hdiag"]), d=d, return_states=True)
            assert_allclose(out["ll"][i], ll, rtol=1e-8, atol=1e-9)
            assert ll[11] == 0.0 and out["ll"][i, 11] == 0.0
            sc = max(1.0, np.abs(stt["a_filt"]).max())
            assert_allclose(out["predicted_states"][i], stt["a_pred"], atol=1e-9 * sc)
            assert_allclose(out["filtered_states"][i], stt["a_filt"], atol=1e-9 * sc)
            pc = np.abs(stt["P_pred"]).max()
            assert_allclose(out["predicted_covs"][i], stt["P_pred"], atol=1e-9 * pc)
            assert_allclose(out["filtered_covs"][i], stt["P_filt"], atol=1e-9 * pc)
            assert_allclose(diag["filtered_covs"][i], np.diagonal(stt["P_filt"], axis1=1, axis2=2), atol=1e-9 * pc)
            assert_allclose(diag["predicted_covs"][i], np.diagonal(stt["P_pred"], axis1=1, axis2=2), atol=1e-9 * pc)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sw_shaped.npz"))
    if "oracle_ll" in g.files:
        n_g = g["oracle_ll"].shape[0]
        bb = wl.sw_shaped_batch(n_g)
        Rg = np.stack([oracle.compute_selection_matrix(bb["B"][i], bb["C"][i], bb["D"][i], bb["T_star"][i]) for i in range(n_g)])
        o2 = batched.kalman_filter_outputs_batched(bb["T_star"], Rg, bb["sigma"] ** 2, om["Z"], om["y"], Hdiag=om["Hdiag"])
        assert_allclose(o2["ll"], g["oracle_ll"], rtol=1e-7, atol=1e-8)


def test_gensys_real_stage_matches_complex_only():
    """gensys window path: the real double-shift sweeps in front of the complex single-shift iteration (dsge_options.
    gensys_real_stage, default on; device model: tests/device_models/gensys_qz_model.py::real_double_shift_stage) are an
    accelerator -- same eu, same status, T equal to 1e-10 with the stage off -- on SW-shaped draws, the reference's golden
    systems (under perturbation) and systems without a unique stable solution."""
    b = wl.sw_shaped_batch(96)
    cases = [(b["A"], b["B"], b["C"], b["D"])]
    fk = wl.full_nk_batch(48)[0]
    cases.append((fk["A"], fk["B"], fk["C"], fk["D"]))
    # explosive / indeterminate variants: scale the lead block (more / fewer unstable roots than forward-looking variables)
    cases.append((b["A"][:32], b["B"][:32], 3.0 * b["C"][:32], b["D"][:32]))
    cases.append((b["A"][:32] * 1.6, b["B"][:32], b["C"][:32] * 0.2, b["D"][:32]))
    for A, B, C, D in cases:
        on = batched.gensys_batched(A, B, C, D, tol=1e-8, options={"gensys_split": 2, "gensys_real_stage": 1})
        off = batched.gensys_batched(A, B, C, D, tol=1e-8, options={"gensys_split": 2, "gensys_real_stage": 0})
        assert np.array_equal(on["eu"], off["eu"])
        assert np.array_equal(on["status"], off["status"]) if "status" in on else True
        good = (off["eu"][:, 0] == 1) & (off["eu"][:, 1] == 1)
        sc = np.maximum(1.0, np.abs(off["T"]).max(axis=(1, 2)))
        err = np.abs(on["T"] - off["T"]).max(axis=(1, 2)) / sc
        assert (err[good] <= 1e-10).all(), err[good].max()
    # and against the oracle (LAPACK's ordered QZ) on the first draws
    on = batched.gensys_batched(b["A"][:8], b["B"][:8], b["C"][:8], b["D"][:8], tol=1e-8, options={"gensys_split": 2})
    for i in range(8):
        Tref, ok = oracle.gensys_T_success(b["A"][i], b["B"][i], b["C"][i], b["D"][i], tol=1e-8)[:2]
        assert ok and np.abs(on["T"][i] - Tref).max() <= 1e-9


def test_gensys_two_draws_per_wavefront_match_one():
    """gensys window path, round 4: the real double-shift sweeps with two draws per wavefront (dsge_options.gensys_pairs,
    dsge_gensys_pair.hpp; default on when the window and #lead are <= 32) against the one-draw kernel -- same eu, same status,
    T equal to 1e-10 -- on an ODD number of SW-shaped draws (the last wavefront has one idle half), on full_nk, on batches whose
    halves finish their sweeps at very different times (a well-behaved draw next to one without a unique stable solution) and
    with a shape of its own in every other draw (two window sizes inside one batch)."""
    b = wl.sw_shaped_batch(67)
    cases = [(b["A"], b["B"], b["C"], b["D"])]
    fk = wl.full_nk_batch(33)[0]
    cases.append((fk["A"], fk["B"], fk["C"], fk["D"]))
    A, B, C, D = (b[x][:32].copy() for x in "ABCD")
    C[1::2] *= 3.0  # odd draws: more unstable roots than forward-looking variables
    A[2::4] *= 1.6
    cases.append((A, B, C, D))
    b2 = wl.sw_shaped_batch(16, n_state=14)  # window 26 next to window 30
    mix = [np.stack([b[x][i] if i % 2 == 0 else b2[x][i] for i in range(16)]) for x in "ABCD"]
    cases.append(tuple(mix))
    for A, B, C, D in cases:
        off = batched.gensys_batched(A, B, C, D, tol=1e-8, options={"gensys_split": 2, "gensys_pairs": 0})
        for mode in (1, 2):  # 1: the sweeps on two draws per wavefront (default); 2: also the Hessenberg-triangular launch
            on = batched.gensys_batched(A, B, C, D, tol=1e-8, options={"gensys_split": 2, "gensys_pairs": mode})
            assert np.array_equal(on["eu"], off["eu"])
            assert np.array_equal(on["status"], off["status"])
            good = (off["eu"][:, 0] == 1) & (off["eu"][:, 1] == 1)
            sc = np.maximum(1.0, np.abs(off["T"]).max(axis=(1, 2)))
            err = np.abs(on["T"] - off["T"]).max(axis=(1, 2)) / sc
            assert (err[good] <= 1e-10).all(), (mode, err[good].max())
    on = batched.gensys_batched(b["A"][:8], b["B"][:8], b["C"][:8], b["D"][:8], tol=1e-8, options={"gensys_split": 2})
    for i in range(8):
        Tref, ok = oracle.gensys_T_success(b["A"][i], b["B"][i], b["C"][i], b["D"][i], tol=1e-8)[:2]
        assert ok and np.abs(on["T"][i] - Tref).max() <= 1e-9


def test_gensys_direct_blocks_match_iteration():
    """gensys window path, round 4: the isolated 2 x 2 blocks the real double-shift stage leaves (complex pairs, well separated
    real pairs) are triangularised in closed form before the zhgeqz-style iteration (dsge_options.gensys_direct_blocks, default
    on; qz_direct_blocks in dsge_gensys.hpp).  With the option off every block goes through the iteration: same eu, same
    status, T equal to 1e-10 -- on SW-shaped draws, on full_nk, on draws without a unique stable solution and with the
    one-draw-per-wavefront sweeps as well."""
    b = wl.sw_shaped_batch(96, seed0=4400)
    cases = [(b["A"], b["B"], b["C"], b["D"])]
    fk = wl.full_nk_batch(33)[0]
    cases.append((fk["A"], fk["B"], fk["C"], fk["D"]))
    A, B, C, D = (b[x][:32].copy() for x in "ABCD")
    C[1::2] *= 3.0
    A[2::4] *= 1.6
    cases.append((A, B, C, D))
    for A, B, C, D in cases:
        for pairs in (1, 0):
            off = batched.gensys_batched(A, B, C, D, tol=1e-8,
                                         options={"gensys_split": 2, "gensys_pairs": pairs, "gensys_direct_blocks": 0})
            on = batched.gensys_batched(A, B, C, D, tol=1e-8,
                                        options={"gensys_split": 2, "gensys_pairs": pairs, "gensys_direct_blocks": 1})
            assert np.array_equal(on["eu"], off["eu"])
            assert np.array_equal(on["status"], off["status"])
            good = (off["eu"][:, 0] == 1) & (off["eu"][:, 1] == 1)
            sc = np.maximum(1.0, np.abs(off["T"]).max(axis=(1, 2)))
            err = np.abs(on["T"] - off["T"]).max(axis=(1, 2)) / sc
            assert (err[good] <= 1e-10).all(), (pairs, err[good].max())


def test_gensys_cached_capacity_record_is_rescued_and_renewed():
    """dsge_options.gensys_shape_cache (default on): the window path measures its capacity record (max #lead, window, deflated
    roots) on the first call of a (model size, lead hint) and launches later calls without the measuring launch and its stream
    synchronisation.  A later batch with a LARGER window (fewer zero columns of A: more state variables) does not fit the cached
    launches: its draws are solved in the same call by the rescue pass (single-launch kernel), and the call after that has a
    fresh record.  All three calls must return what the uncached path returns."""
    n, nl, k = 36, 10, 4
    small = wl.sw_shaped_batch(24, n=n, n_state=12, n_lead=nl, k=k, seed0=91000)
    large = wl.sw_shaped_batch(24, n=n, n_state=20, n_lead=nl, k=k, seed0=92000)  # window 30 instead of 22
    ref_small = batched.gensys_batched(*(small[x] for x in "ABCD"), tol=1e-8, n_lead_hint=nl,
                                       options={"gensys_split": 2, "gensys_shape_cache": 0})
    ref_large = batched.gensys_batched(*(large[x] for x in "ABCD"), tol=1e-8, n_lead_hint=nl,
                                       options={"gensys_split": 2, "gensys_shape_cache": 0})
    assert ref_small["success"].all() and ref_large["success"].all()
    opts = {"gensys_split": 2, "gensys_shape_cache": 1}
    runs = [(small, ref_small), (small, ref_small), (large, ref_large), (large, ref_large), (small, ref_small), (large, ref_large)]
    for bb, ref in runs:  # measure / cached / cached + rescue / measured again / cached (record larger than needed) / cached
        out = batched.gensys_batched(*(bb[x] for x in "ABCD"), tol=1e-8, n_lead_hint=nl, options=opts)
        assert np.array_equal(out["eu"], ref["eu"]) and np.array_equal(out["status"], ref["status"])
        assert np.abs(out["T"] - ref["T"]).max() <= 1e-9
        assert np.abs(out["T"] - bb["T_star"]).max() <= 1e-8
        assert np.abs(out["R"] - ref["R"]).max() <= 1e-8


def test_zero_T_on_solver_failure_matches_the_reference_default_graph():
    """add_solver_success_check=False is the reference's default (statespace.py:1148, 1210-1215): a failed cycle reduction hands
    on T = 0 (cycle_reduction.py:181) and the model sees the FINITE log-likelihood of that system.  With
    DSGE_SOLVER_FLAG_ZERO_T_ON_FAILURE the fused call does the same (status still reports the failure); without it (default)
    the draw gets -inf.  Converged draws are unaffected by the flag (1e-10: R comes from the explicit solve instead of the
    solver's final elimination)."""
    nb = 12
    b = wl.sw_shaped_batch(nb)
    om = wl.sw_shaped_observation_model()
    q = b["sigma"] ** 2
    y = om["y"][:80]
    kw = dict(Hdiag=om["Hdiag"], tol=1e-8, q_mode=1)
    # max_iter = 6: some of these draws need 7..9 iterations
    strict = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], y, max_iter=6, return_policy=True, **kw)
    loose = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], y, max_iter=6, return_policy=True,
                                              add_solver_success_check=False, **kw)
    failed = strict["status"] != 0
    assert failed.any() and (~failed).any()
    assert np.array_equal(loose["status"], strict["status"])
    assert np.all(strict["logp"][failed] == -np.inf) and np.all(np.isfinite(loose["logp"][failed]))
    assert np.all(loose["T"][failed] == 0.0)
    ok = ~failed
    assert np.abs(loose["logp"][ok] - strict["logp"][ok]).max() <= 1e-10 * np.abs(strict["logp"][ok]).max()
    for i in range(nb):
        ref = oracle.solve_kalman_logp(b["A"][i], b["B"][i], b["C"][i], b["D"][i], np.diag(q[i]), om["Z"], y, H=np.diag(om["Hdiag"]),
                                       tol=1e-8, max_iter=6, add_solver_success_check=False)
        assert ref["success"] == (not failed[i])
        assert abs(loose["logp"][i] - ref["logp"]) <= 1e-8 * abs(ref["logp"]), (i, loose["logp"][i], ref["logp"])
        if failed[i]:
            assert np.abs(loose["R"][i] - ref["R"]).max() <= 1e-9 * max(1.0, np.abs(ref["R"]).max())


def test_round3_entry_points_reject_malformed_calls_and_accept_empty_batches():
    """Call-level errors of the entry points added in round 3 are return codes with a message (never a crash, never a wrong
    number): the dense-Z gradient beyond n + p = 56, a full covariance of the wrong shape, second order with p > 8 or a
    retained set that does not list the states first, filter outputs with p > DSGE_MAX_P; an empty batch is a no-op."""
    b = wl.sw_shaped_batch(2)
    om = wl.sw_shaped_observation_model()
    q = b["sigma"] ** 2
    y = om["y"][:20]
    # n + p > 56 on the dense-Z gradient route
    sysm = [wl.sw_shaped_system(9100 + i, n=52, n_state=20, n_lead=14, k=7) for i in range(2)]
    A52, B52, C52, D52 = (np.stack([s_[j] for s_ in sysm]) for j in range(4))
    Zd = np.random.default_rng(0).standard_normal((7, 52))
    with pytest.raises(_lib.DsgeHipError, match="n \\+ p"):
        batched.solve_kalman_logp_grad_batched(A52, B52, C52, D52, np.full((2, 7), 1e-4), Zd, y, dense_z=True)
    with pytest.raises(ValueError):
        batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], None, om["Z"], y, Q=np.eye(3))
    with pytest.raises(ValueError):
        batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], y, Q=np.eye(7))
    # second order: retained variables must list the states first
    s2 = wl.sw_second_order_batch(2)
    S, Lc, U = batched.second_order_structure(s2["A"], s2["C"], om["Z"])
    bad_U = np.ascontiguousarray(U[::-1])
    with pytest.raises(_lib.DsgeHipError, match="states first"):
        batched.second_order_logp_batched(s2["A"], s2["B"], s2["C"], s2["D"], s2["hess_idx"], s2["hess_val"], s2["sigma"] ** 2,
                                          om["Z"], y, Hdiag=om["Hdiag"], structure=(S, Lc, bad_U))
    # empty batches
    e3 = np.empty((0, 40, 40))
    out = batched.solve_kalman_logp_grad_batched(e3, e3, e3, np.empty((0, 40, 7)), np.empty((0, 7)), om["Z"], y, dense_z=True,
                                                 return_Z_bar=True)
    assert out["logp"].shape == (0,) and out["Z_bar"].shape == (0, 7, 40)
    out = batched.kalman_filter_outputs_batched(np.empty((0, 40, 40)), np.empty((0, 40, 7)), np.empty((0, 7)), om["Z"], y)
    assert out["ll"].shape == (0, 20)
