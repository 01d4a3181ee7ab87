"""CPU: executable models of the DEVICE algorithms vs the oracle.

The device kernels do not run the reference's LAPACK calls; they run (a) a complex single-shift
QZ + adjacent-swap reordering + Jacobi SVDs for gensys and (b) a downdate-form Kalman update.
These models restate exactly that arithmetic in numpy so that the algorithms themselves are
validated here, on the CPU, against the oracle and the golden vectors."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import oracle
from geconpy_amd import workloads as wl
from tests.device_models.gensys_qz_model import gensys_device_model, jacobi_svd, lartg
from tests.device_models.adjoint_compact_model import (compact_doubling, full_doubling, fused_pullback, kronecker_solve,
                                                       two_step_pullback)
from tests.device_models.kalman_model import kalman_downdate_logp
from tests.device_models.kalman_tile_model import kalman_tile_logp, retained_variables


def test_lartg():
    rng = np.random.default_rng(0)
    for _ in range(50):
        f, g = rng.standard_normal(2) + 1j * rng.standard_normal(2)
        if rng.random() < 0.2:
            f = 0j
        if rng.random() < 0.2:
            g = 0j
        c, s, r = lartg(f, g)
        assert_allclose(c * f + s * g, r, atol=1e-14)
        assert_allclose(-np.conj(s) * f + c * g, 0, atol=1e-14)
        assert_allclose(c * c + abs(s) ** 2, 1.0, atol=1e-14)


def test_jacobi_svd():
    rng = np.random.default_rng(1)
    for r, c in [(12, 12), (40, 12), (6, 12), (3, 1)]:
        M = rng.standard_normal((r, c)) + 1j * rng.standard_normal((r, c))
        if r == 12:
            M[:, 3] = M[:, 2]  # rank deficient
        G, V, s = jacobi_svd(M)
        assert_allclose(G @ V.conj().T, M, atol=1e-12)
        assert_allclose(V.conj().T @ V, np.eye(c), atol=1e-12)
        assert_allclose(np.sort(s)[::-1][: min(r, c)], np.linalg.svd(M, compute_uv=False), atol=1e-10)


@pytest.mark.parametrize("key", ["one_block", "rbc_2_block", "full_nk"])
def test_gensys_model_reference_goldens(ref_goldens, key):
    g = ref_goldens
    A, B, C, D = (g[f"{key}_{x}"] for x in "ABCD")
    T, eu, info = gensys_device_model(A, B, C, D, 1e-8)
    assert list(eu) == list(g[f"{key}_ref_gensys_eu"])
    assert_allclose(T, g[f"{key}_ref_gensys_T"], atol=1e-10, rtol=0)
    # same generalized eigenvalue moduli as LAPACK
    lam = np.sort(np.abs(info["beta"]) / np.maximum(np.abs(info["alpha"]), 1e-300))
    gr = g[f"{key}_ref_gensys_gev"]
    lam_r = np.sort(np.abs(gr[:, 1]) / np.maximum(np.abs(gr[:, 0]), 1e-300))
    fin = lam_r < 1e6
    assert_allclose(lam[fin], lam_r[fin], rtol=1e-6, atol=1e-9)


def test_gensys_model_rbc_and_sw(rbc_golden, sw_golden):
    th = {k[6:]: rbc_golden[k] for k in rbc_golden.files if k.startswith("theta_")}
    A, B, C, D = wl.rbc_linearized_jacobians(**th)
    for i in range(0, 64, 4):
        T, eu, _ = gensys_device_model(A[i], B[i], C[i], D[i], 1e-8)
        assert list(eu) == [1, 1, 0]
        assert_allclose(T, rbc_golden["ref_gensys_T"][i], atol=1e-10)
    b = wl.sw_shaped_batch(4)
    for i in range(4):
        T, eu, _ = gensys_device_model(b["A"][i], b["B"][i], b["C"][i], b["D"][i], 1e-8)
        assert list(eu) == list(sw_golden["ref_gensys_eu"][i])
        assert_allclose(T, sw_golden["ref_gensys_T"][i], atol=1e-10)


@pytest.mark.parametrize("name", ["ok", "nonunique", "noexist", "coincident"])
def test_gensys_model_failure_codes(failure_golden, name):
    g = failure_golden
    A, B, C, D = (g[f"{name}_{x}"] for x in "ABCD")
    T, eu, _ = gensys_device_model(A, B, C, D, 1e-8)
    assert list(eu) == list(g[f"{name}_ref_gensys_eu"])
    if name in ("ok", "noexist"):
        assert_allclose(T, g[f"{name}_ref_gensys_T"], atol=1e-9)
    if name == "coincident":
        assert np.all(T == 0)


def test_gensys_model_random_structures():
    """Small random systems with singular A / C blocks (zero and infinite roots, repeated roots)."""
    rng = np.random.default_rng(5)
    n_checked = 0
    for trial in range(40):
        n = int(rng.integers(3, 9))
        ns = int(rng.integers(1, n))
        nl = int(rng.integers(1, n))
        A, B, C, D, Tst = wl.sw_shaped_system(1000 + trial, n=n, n_state=ns, n_lead=nl, k=1)
        if trial % 3 == 0:  # static equation: a row that involves only time-t variables
            A[0] = 0
            C[0] = 0
        To, ok, euo = oracle.gensys_T_success(A, B, C, D, 1e-8)
        T, eu, _ = gensys_device_model(A, B, C, D, 1e-8)
        assert list(eu) == list(euo), (trial, eu, euo)
        if ok:
            assert_allclose(T, To, atol=1e-8)
            n_checked += 1
    assert n_checked > 10


def test_kalman_downdate_model_equals_joseph_oracle():
    b = wl.sw_shaped_batch(3)
    om = wl.sw_shaped_observation_model()
    y = om["y"].copy()
    y[5, 2] = np.nan
    y[17, :] = np.nan
    y[40, 0] = oracle.MISSING_FILL
    for i in range(3):
        Q = np.diag(b["sigma"][i] ** 2)
        r = oracle.solve_kalman_logp(b["A"][i], b["B"][i], b["C"][i], b["D"][i], Q, om["Z"], y, H=np.diag(om["Hdiag"]))
        lp = kalman_downdate_logp(y, r["T"], r["R"], Q, om["Z"], om["Hdiag"], np.zeros(7), r["P0"])
        assert_allclose(lp, r["logp"], rtol=1e-12)


@pytest.mark.parametrize("observed", [None, wl.SW_OBSERVED_JUMPS], ids=["observe_states", "observe_jumps"])
def test_kalman_tile_model_equals_joseph_oracle(observed):
    """kalman_mf_kernel's recursion (upper triangle only, downdate as a product, prediction through the state block, one Newton
    step on the reciprocals, steady-state switch) against the reference's Joseph-form filter, with missing data."""
    b = wl.sw_shaped_batch(3)
    om = wl.sw_shaped_observation_model(observed=observed)
    y = om["y"].copy()
    y[5, 2] = np.nan
    y[17, :] = np.nan
    y[40, 0] = oracle.MISSING_FILL
    y[150, 3] = np.nan  # (after the switch: the full update resumes for this step)
    for i in range(3):
        Q = np.diag(b["sigma"][i] ** 2)
        r = oracle.solve_kalman_logp(b["A"][i], b["B"][i], b["C"][i], b["D"][i], Q, om["Z"], y, H=np.diag(om["Hdiag"]))
        RQR = r["R"] @ Q @ r["R"].T
        perm, s = retained_variables(r["T"], om["Z"])
        assert s < r["T"].shape[0] and len(perm) == (18 if observed is None else 25)  # (retained variables of the 40)
        lp, at = kalman_tile_logp(y, r["T"], RQR, om["Z"], om["Hdiag"], np.zeros(7), r["P0"], return_steady_step=True)
        assert_allclose(lp, r["logp"], rtol=1e-12)
        assert 0 < at < y.shape[0]
        lp_full = kalman_tile_logp(y, r["T"], RQR, om["Z"], om["Hdiag"], np.zeros(7), r["P0"], steady_tol=0.0)
        assert_allclose(lp_full, r["logp"], rtol=1e-12)
        assert_allclose(lp, lp_full, rtol=1e-12)
        # the reciprocal: hardware seed + one Newton step against the correctly rounded quotient
        lp_exact = kalman_tile_logp(y, r["T"], RQR, om["Z"], om["Hdiag"], np.zeros(7), r["P0"], seed_error=0.0)
        assert_allclose(lp, lp_exact, rtol=1e-13)
        # and the register-block kernels' recursion (symmetrised prediction) agrees with the mirrored upper triangle
        lp_dd = kalman_downdate_logp(y, r["T"], r["R"], Q, om["Z"], om["Hdiag"], np.zeros(7), r["P0"])
        assert_allclose(lp, lp_dd, rtol=1e-12)


def test_kalman_tile_model_long_sample_stays_symmetric_positive():
    """Carrying one triangle for 2000 steps (no symmetrisation, no Joseph form): same log-likelihood as the oracle's recursion."""
    b = wl.sw_shaped_batch(1, first_draw=11)
    om = wl.sw_shaped_observation_model(T_len=2000)
    Q = np.diag(b["sigma"][0] ** 2)
    r = oracle.solve_kalman_logp(b["A"][0], b["B"][0], b["C"][0], b["D"][0], Q, om["Z"], om["y"], H=np.diag(om["Hdiag"]))
    lp = kalman_tile_logp(om["y"], r["T"], r["R"] @ Q @ r["R"].T, om["Z"], om["Hdiag"], np.zeros(7), r["P0"], steady_tol=0.0)
    assert_allclose(lp, r["logp"], rtol=1e-12)


def test_window_gensys_algebra(ref_goldens, failure_golden):
    """Post-processing identity of the three-launch gensys path (dsge_gensys_win.hpp) against the reference outputs:
    goldens, SW-shaped draws, and the non-unique / no-solution systems (T is still G1[:n,:n] there)."""
    from tests.device_models.gensys_window_model import window_gensys
    from geconpy_amd import workloads as wl

    for key in ("one_block", "rbc_2_block", "full_nk"):
        A, B, C = (ref_goldens[f"{key}_{x}"] for x in "ABC")
        T, eu = window_gensys(A, B, C)
        assert eu == [1, 1, 0]
        assert_allclose(T, ref_goldens[f"{key}_ref_gensys_T"], atol=1e-10)
    for name in ("ok", "nonunique", "noexist"):
        A, B, C = (failure_golden[f"{name}_{x}"] for x in "ABC")
        T, eu = window_gensys(A, B, C)
        assert eu == list(failure_golden[f"{name}_ref_gensys_eu"])
        assert_allclose(T, failure_golden[f"{name}_ref_gensys_T"], atol=1e-10)
    b = wl.sw_shaped_batch(3)
    for i in range(3):
        T, eu = window_gensys(b["A"][i], b["B"][i], b["C"][i])
        T_ref, ok, _ = oracle.gensys_T_success(b["A"][i], b["B"][i], b["C"][i], b["D"][i], tol=1e-8)
        assert ok and eu == [1, 1, 0]
        assert_allclose(T, T_ref, atol=1e-11)


def test_cr_static_deflation_algebra(ref_goldens):
    """Static-variable deflation in front of cycle reduction (dsge_cr_deflate.hpp): same T and R as the reference's
    cycle reduction on the full system, on the goldens and on SW-shaped draws; also with fewer static variables
    deflated than the system has (the device's lower-bound hint)."""
    from tests.device_models.cr_deflation_model import deflated_cycle_reduction

    for key, h_expect in (("one_block", 3), ("rbc_2_block", 6), ("full_nk", 4)):
        A, B, C, D = (ref_goldens[f"{key}_{x}"] for x in "ABCD")
        T_ref = ref_goldens[f"{key}_ref_cr_T"]
        R_ref = oracle.compute_selection_matrix(B, C, D, T_ref)
        for h in (None, 1):
            T, R, h_used, _ = deflated_cycle_reduction(A, B, C, D, h=h)
            assert h_used == (h_expect if h is None else h)
            assert_allclose(T, T_ref, atol=1e-10)
            assert_allclose(R, R_ref, atol=1e-10)
    b = wl.sw_shaped_batch(3)
    for i in range(3):
        A, B, C, D = (b[x][i] for x in "ABCD")
        T, R, h_used, it = deflated_cycle_reduction(A, B, C, D)
        full = oracle.cycle_reduction_core(A, B, C, 1000, 1e-9)
        assert h_used == 10 and it == full[2]
        assert_allclose(T, b["T_star"][i], atol=1e-11)
        assert_allclose(R, oracle.compute_selection_matrix(B, C, D, b["T_star"][i]), atol=1e-11)


def test_gensys_model_real_double_shift_stage():
    """Device-algorithm model of round 3's real double-shift accelerator (gensys_qz_model.real_double_shift_stage, the model of
    gw_realqz_sweeps in gensys_hesstri_kernel): with and without it the model returns the same eu and T (1e-10), both equal to the oracle's
    LAPACK-based gensys, and the complex single-shift iteration is left with a handful of rotations."""
    from geconpy_amd import workloads as wl

    b = wl.sw_shaped_batch(3)
    for i in range(3):
        A, B, C, D = (b[x][i] for x in "ABCD")
        T0, eu0, i0 = gensys_device_model(A, B, C, D, real_stage=False)
        T1, eu1, i1 = gensys_device_model(A, B, C, D, real_stage=True)
        assert list(eu0) == list(eu1) == [1, 1, 0]
        assert np.abs(T0 - T1).max() <= 1e-10
        Tref, ok, _ = oracle.gensys_T_success(A, B, C, D, tol=1e-8)
        assert ok and np.abs(T1 - Tref).max() <= 1e-10
        steps, sweeps = i1["real_steps"]
        assert steps > 0 and i1["rot_qz"] < 0.05 * i0["rot_qz"], (steps, i1["rot_qz"], i0["rot_qz"])


def test_refinement_with_an_extended_residual_on_the_ill_conditioned_fixture():
    """tests/golden/cr_ill_conditioned_54.npz: the stored oracle T is what the oracle computes, it is ~2e-8 from the stored
    40-digit T, and cycle reduction whose solves take ONE step of iterative refinement with the residual in extended
    precision (numpy longdouble standing in for the device's Dot2 residual, mm_residual_dot2 in dsge_device.hpp) ends several
    times closer to the exact T -- while a float64 residual, one step or two, stays at the reference's level."""
    import os

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cr_ill_conditioned_54.npz"))
    A, B, C, Tx, tol = g["A"], g["B"], g["C"], g["T_exact"], float(g["tol"])
    Tc, conv, itc = oracle.cycle_reduction_core(A, B, C, 200, tol)
    assert conv and np.abs(Tc - g["T_oracle"]).max() <= 1e-12
    e_ref = np.abs(Tc - Tx).max()
    assert 5e-9 < e_ref < 1e-7
    n = A.shape[0]

    def cr(extended):
        A0, A1, A2, Ah = A.copy(), B.copy(), C.copy(), B.copy()
        for it in range(50):
            R = np.hstack([A0, A2])
            X = np.linalg.solve(A1, R)
            if extended:
                r = (R.astype(np.longdouble) - A1.astype(np.longdouble) @ X.astype(np.longdouble)).astype(np.float64)
            else:
                r = R - A1 @ X
            X = X + np.linalg.solve(A1, r)
            X0, X2 = X[:, :n], X[:, n:]
            m00, m02, m20, m22 = A0 @ X0, A0 @ X2, A2 @ X0, A2 @ X2
            A1, Ah, A0, A2 = A1 - m02 - m20, Ah - m20, -m00, -m22
            if np.abs(A0).sum(0).max() < tol and np.abs(A2).sum(0).max() < tol:
                break
        return -np.linalg.solve(Ah, A), it + 1

    T_ext, it_ext = cr(True)
    T_f64, it_f64 = cr(False)
    assert it_ext == itc and it_f64 == itc
    if np.finfo(np.longdouble).eps < 1e-18:  # (platforms whose longdouble is float64 have nothing to show here)
        assert np.abs(T_ext - Tx).max() <= 0.3 * e_ref
    assert np.abs(T_f64 - Tx).max() >= 0.3 * e_ref


# ---- gensys by spectral division (round 5): the certificate's decision rule against the oracle's gensys -------------------------
def _spectral_division_cases(seed, trials):
    from tests.device_models.spectral_division_model import gensys_by_spectral_division

    rng = np.random.default_rng(seed)
    out = []
    for _ in range(trials):
        n = int(rng.integers(4, 30))
        ns = int(rng.integers(1, max(2, n // 2)))
        nl = int(rng.integers(1, max(2, n // 3)))
        k = int(rng.integers(1, min(n, 5) + 1))
        A, B, C, D, Tst = wl.sw_shaped_system(int(rng.integers(1 << 30)), n=n, n_state=ns, n_lead=nl, k=k)
        kind = "regular"
        u = rng.random()
        M = B + C @ Tst
        if u < 0.15:
            A = A * rng.uniform(5, 40)
            kind = "explosive"
        elif u < 0.30:
            G = np.linalg.solve(M, C)
            G = G * (rng.uniform(1.05, 3.0) / np.max(np.abs(np.linalg.eigvals(G))))
            C = M @ G
            B = M - C @ Tst
            kind = "indeterminate"
        elif u < 0.45:
            T2 = Tst.copy()
            T2[:, :ns] *= (1.0 + rng.choice([-1, 1]) * 10.0 ** rng.uniform(-9, -3)) / np.max(np.abs(np.linalg.eigvals(T2[:ns, :ns])))
            A = -M @ T2
            B = M - C @ T2
            kind = "near the unit circle"
        elif u < 0.50:
            C = C.copy()
            C[:, n - 1] *= 1e-12
            kind = "lead column below the tolerance"
        T, cert = gensys_by_spectral_division(A, B, C, 1e-8)
        T_ref, succ, eu = oracle.gensys_T_success(A, B, C, D, 1e-8)
        out.append((kind, cert, bool(succ), [int(e) for e in eu], T, T_ref))
    return out


def test_spectral_division_certificate_never_contradicts_gensys():
    """A certified draw is ALWAYS a draw on which the oracle's gensys says eu = [1, 1, 0], with the same T; the draws gensys
    accepts but the certificate does not take (roots within 2e-4 of the unit circle, a lead column below the tolerance) are the
    ones the device hands to the ordered QZ.  450 random systems, a third of them non-regular by construction."""
    n_cert = n_succ = 0
    kinds = {}
    for seed in (11, 12, 13):
        for kind, cert, succ, eu, T, T_ref in _spectral_division_cases(seed, 150):
            kinds.setdefault(kind, [0, 0, 0])
            kinds[kind][0] += 1
            kinds[kind][1] += cert
            kinds[kind][2] += succ
            if cert:
                assert succ and eu == [1, 1, 0], (kind, eu)
                assert_allclose(T, T_ref, rtol=0, atol=1e-8 * max(1.0, np.abs(T_ref).max()))
            n_cert += cert
            n_succ += succ
    # the certificate takes (nearly) every regular draw: that is what makes it the fast path
    assert kinds["regular"][1] >= 0.97 * kinds["regular"][0], kinds
    # (scaling A does not always destroy the stable solvent: a few "explosive" systems stay regular -- for gensys and for the
    #  certificate alike, which the per-draw assertion above has checked)
    assert kinds["indeterminate"][1] == 0 and kinds["indeterminate"][2] == 0, kinds
    assert kinds["explosive"][1] <= kinds["explosive"][2] <= 0.2 * kinds["explosive"][0], kinds
    assert n_cert <= n_succ


def _q2pi_lapack(A, B, C, D, tol=1e-8):
    """(sorted singular values of Q2 @ pi, smallest diagonal pair) from LAPACK's ordered QZ (gensys.py:227-235, 243, 267-273)."""
    import scipy.linalg as sla

    from oracle.gensys_qz import gensys_setup
    from tests.device_models.scale_cases import lapack_margins

    g0, g1, c, psi, pi = gensys_setup(A, B, C, D, tol)
    _, _, alpha, beta, Qraw, _ = sla.ordqz(g0.astype(complex), g1.astype(complex), sort="ouc", output="complex")
    aa, bb = np.abs(alpha), np.abs(beta)
    stable = ((bb < tol) & (aa >= tol)) | ((bb >= tol) & (aa > bb))
    nu = int(np.sum(~stable))
    Q2 = Qraw.conj().T[len(alpha) - nu :]
    sv = np.sort(sla.svd(Q2 @ pi, compute_uv=False))
    assert abs(lapack_margins(A, B, C, D, tol)[0] - sv[0]) <= 1e-12 * max(1.0, sv[0])
    return sv, float(np.maximum(aa, bb).min())


def test_q2pi_singular_values_closed_form_vs_lapack():
    """The identity behind the certificate's existence guard (round 6): the singular values of gensys's Q2 pi are
    1 / sqrt(1 + sigma_i(N_L)^2), N_L = lead rows of (B + C T)^-1 -- checked against LAPACK's ordered QZ on regular systems and on
    systems with one equation scaled by 1e-3 / 1e-6 (sigma_min follows the scale), 1e-9 relative."""
    from tests.device_models.spectral_division_model import q2pi_singular_values

    rng = np.random.default_rng(61)
    for trial in range(24):
        n = int(rng.integers(6, 24))
        A, B, C, D, Tst = wl.sw_shaped_system(6100 + trial, n=n, n_state=max(1, n // 3), n_lead=max(1, n // 4), k=2)
        e = [1.0, 1e-3, 1e-6][trial % 3]
        r = int(rng.integers(n))
        for X in (A, B, C, D):
            X[r] *= e
        s_ref, _ = _q2pi_lapack(A, B, C, D)
        s_formula = np.sort(q2pi_singular_values(B, C, Tst))
        assert_allclose(s_formula, s_ref, rtol=1e-7, atol=0)


def test_scale_guards_never_certify_what_gensys_rejects():
    """one equation multiplied by <= 1e-9 (or the whole system by ~1e-8, or one variable rescaled) leaves T untouched, so
    the round-5 certificate said eu = [1, 1, 0] where the reference says [-2, -2, 0] / [0, ...].  With the scale guards: 0
    contradictions in 900 such systems -- and the guards are what does it (the round-5 rule contradicts the oracle on the same list)."""
    from tests.device_models.scale_cases import scaled_system
    from tests.device_models.spectral_division_model import gensys_by_spectral_division

    rng = np.random.default_rng(62)
    stats = {k: [0, 0, 0, 0] for k in ("row", "global", "col", "regular")}  # n, oracle ok, certified, certified by the r5 rule
    bad_r5 = 0
    for trial in range(900):
        kind = ("row", "global", "col", "regular")[trial % 4]
        A, B, C, D, e = scaled_system(rng, kind)
        T, cert = gensys_by_spectral_division(A, B, C, 1e-8)
        _, cert_r5 = gensys_by_spectral_division(A, B, C, 1e-8, guards=False)
        _, succ, eu = oracle.gensys_T_success(A, B, C, D, 1e-8)
        regular = [int(x) for x in eu] == [1, 1, 0]
        st = stats[kind]
        st[0] += 1
        st[1] += regular
        st[2] += cert
        st[3] += cert_r5
        assert not cert or regular, (kind, e, eu)
        bad_r5 += cert_r5 and not regular
    assert bad_r5 >= 50, bad_r5  # (what the guards are for)
    assert stats["regular"][2] == stats["regular"][0], stats  # unscaled systems: every one certified
    # the guards are not vacuous either: most scaled systems that the reference still solves keep the fast path
    assert stats["row"][2] >= 0.8 * stats["row"][1] and stats["global"][2] >= 0.5 * stats["global"][1], stats


@pytest.mark.parametrize("factor", [1e-4, 1e-2, 0.3, 0.7, 0.95, 1.05, 1.3, 3.0, 1e2, 1e4])
def test_existence_tolerance_sweep(factor):
    """sigma_min(Q2 pi) swept through realsmall (gensys.py:276-283), tol = 1e-8 ... and two other tolerances.
    The certificate never says [1, 1, 0] where the oracle's existence test fails, and it does take the draw once sigma_min is a
    factor 3 clear of the tolerance (and no diagonal pair of the QZ is below it)."""
    from tests.device_models.scale_cases import existence_sweep_system
    from tests.device_models.spectral_division_model import gensys_by_spectral_division

    for tol in (1e-8, 1e-6, 1e-10):
        for seed in (7001, 7002, 7003):
            A, B, C, D, e = existence_sweep_system(seed, factor * tol)
            smin, min_pair = _q2pi_lapack(A, B, C, D, tol)
            assert abs(smin[0] / (factor * tol) - 1.0) < 1e-3  # (the construction hit its target)
            T, cert = gensys_by_spectral_division(A, B, C, tol)
            _, succ, eu = oracle.gensys_T_success(A, B, C, D, tol)
            regular = [int(x) for x in eu] == [1, 1, 0]
            assert not cert or regular, (tol, seed, factor, eu)
            if factor < 1.0:
                assert not regular and not cert
            if factor >= 3.0 and min_pair > 3.0 * tol:
                assert regular and cert, (tol, seed, factor, eu, min_pair)


@pytest.mark.parametrize("factor", [1e-3, 0.5, 0.999, 1.001, 2.0, 1e3])
def test_lead_column_tolerance_sweep(factor):
    """A lead column of C with sum|C_ij| straddling tol (gensys.py:587): below it gensys drops the column from the pencil although
    the doubling iteration used it -- never certified; above it the column is an ordinary lead column."""
    from tests.device_models.scale_cases import lead_column_sweep_system
    from tests.device_models.spectral_division_model import gensys_by_spectral_division

    for seed in (7101, 7102):
        A, B, C, D = lead_column_sweep_system(seed, factor * 1e-8)
        T, cert = gensys_by_spectral_division(A, B, C, 1e-8)
        _, succ, eu = oracle.gensys_T_success(A, B, C, D, 1e-8)
        assert not cert or [int(x) for x in eu] == [1, 1, 0], (seed, factor, eu)
        if factor <= 1.0:
            assert not cert


def test_spectral_division_on_the_reference_failure_cases():
    """tests/golden/failure_cases.npz (the reference's own solvability cases, incl. the indeterminate system on which its cycle
    reduction CONVERGES): certified iff the reference's gensys returns eu = [1, 1, 0]."""
    import os

    from tests.device_models.spectral_division_model import certify_contraction, gensys_by_spectral_division

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "failure_cases.npz"))
    for name in ("ok", "nonunique", "noexist", "coincident"):
        A, B, C = g[f"{name}_A"], g[f"{name}_B"], g[f"{name}_C"]
        T, cert = gensys_by_spectral_division(A, B, C, 1e-8)
        eu = [int(e) for e in g[f"{name}_ref_gensys_eu"]]
        assert cert == (eu[:2] == [1, 1]), (name, cert, eu)
        if cert:
            assert_allclose(T, g[f"{name}_ref_gensys_T"], atol=1e-9)
    # the reach of the certificate: rho = 1 - 1e-3 is taken, rho = 1 - 1e-5 is not (2^-1/2^12 = 0.99983), rho > 1 never
    assert certify_contraction(np.diag([0.999, 0.5])) and not certify_contraction(np.diag([1.0 - 1e-5, 0.5]))
    assert not certify_contraction(np.diag([1.0 + 1e-9, 0.1])) and not certify_contraction(np.array([[np.nan]]))
    J = np.array([[0.9, 50.0], [0.0, 0.9]])  # non-normal: the norms grow before they decay, the squares still get there
    assert certify_contraction(J)


def test_compact_adjoint_stein_equation_against_the_kronecker_solve():
    """adj_stein_solve_compact's algebra: the nl x ns Stein equation reproduces the reference's Kronecker solve of the policy
    adjoints (shared.py:53-71) -- to 1e-11 on ordinary SW-shaped draws and to the reference's own level on the nearly singular draw
    752 (cond(B + C T) = 3e8), where the n x n doubling of rounds 2-5 overflows; and the powers of the compact Gs do not grow."""
    rng = np.random.default_rng(3)
    worst_full = 0.0
    for first, count in ((0, 12), (752, 1)):
        b = wl.sw_shaped_batch(count, first_draw=first)
        for i in range(count):
            A, B, C, D = (b[k][i] for k in "ABCD")
            T = oracle.solve_policy_function_with_cycle_reduction(A, B, C, D, max_iter=1000, tol=1e-12)[0]
            n = T.shape[0]
            T_bar = rng.standard_normal((n, n))
            T_bar[:, np.all(T == 0, axis=0)] = 0.0  # (the structurally zero columns of T carry no cotangent)
            M = B + C @ T
            S_ref = kronecker_solve(M, C, T, T_bar)
            S_c, growth_c, nl, ns = compact_doubling(M, C, T, T_bar)
            assert (nl, ns) == (wl.SW_SHAPE["n_lead"], wl.SW_SHAPE["n_state"])
            err = np.abs(S_c - S_ref).max() / np.abs(S_ref).max()
            assert growth_c < 1.0
            if first == 752:
                assert np.linalg.cond(M) > 1e8 and err < 1e-6
                S_f, growth_f = full_doubling(M, C, T, T_bar)
                assert not np.isfinite(S_f).all() or growth_f > 1e10  # (what the elimination-based fall-back was for)
            else:
                assert err < 1e-11
                S_f, growth_f = full_doubling(M, C, T, T_bar)
                worst_full = max(worst_full, np.abs(S_f - S_ref).max() / np.abs(S_ref).max())
    assert worst_full < 1e-7


def test_fused_assembly_and_adjoint_algebra():
    """The gradient pipeline's fused launch: the pullback of R = -(B + C T)^-1 D on the elimination the policy adjoints need anyway
    (one factorisation of B + C T instead of two) gives the cotangents of the reference's two steps."""
    rng = np.random.default_rng(5)
    b = wl.sw_shaped_batch(6)
    for i in range(6):
        A, B, C, D = (b[k][i] for k in "ABCD")
        T = oracle.solve_policy_function_with_cycle_reduction(A, B, C, D, max_iter=1000, tol=1e-12)[0]
        R = oracle.compute_selection_matrix(B, C, D, T)
        n = T.shape[0]
        T_bar = rng.standard_normal((n, n))
        T_bar[:, np.all(T == 0, axis=0)] = 0.0
        R_bar = rng.standard_normal(R.shape)
        ref = two_step_pullback(B, C, T, R, R_bar, T_bar)
        fus = fused_pullback(B, C, T, R, R_bar, T_bar)
        for key in ref:
            assert np.abs(fus[key] - ref[key]).max() <= 1e-10 * np.abs(ref[key]).max(), key
