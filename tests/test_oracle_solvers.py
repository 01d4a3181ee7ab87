"""CPU: the oracle restatement vs the reference's own outputs (golden fixtures).

Mirrors the reference's solver tests:
  tests/model/test_perturbation.py:163-206 (gensys == cycle reduction, T structure),
  tests/model/test_model.py:501-529 (failure codes), tests/model/test_model.py:405-421
  (golden A,B,C,D are the inputs).
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import oracle
from geconpy_amd import workloads as wl

KEYS = ["one_block", "rbc_2_block", "full_nk"]


def _abcd(g, key):
    return tuple(g[f"{key}_{x}"] for x in "ABCD")


@pytest.mark.parametrize("key", KEYS)
def test_gensys_matches_reference(ref_goldens, key):
    A, B, C, D = _abcd(ref_goldens, key)
    G1, Cc, impact, fmat, fwt, ywt, gev, eu, loose = oracle.solve_policy_function_with_gensys(A, B, C, D, 1e-8)
    n = A.shape[0]
    assert list(eu) == list(ref_goldens[f"{key}_ref_gensys_eu"])
    assert_allclose(G1, ref_goldens[f"{key}_ref_gensys_G1"], atol=1e-11, rtol=0)
    assert_allclose(G1[:n, :n], ref_goldens[f"{key}_ref_gensys_T"], atol=1e-11, rtol=0)
    assert_allclose(impact[:n], ref_goldens[f"{key}_ref_gensys_R"], atol=1e-11, rtol=0)
    # generalized eigenvalue moduli agree as multisets
    lam = np.sort(np.abs(gev[:, 1]) / np.maximum(np.abs(gev[:, 0]), 1e-300))
    gr = ref_goldens[f"{key}_ref_gensys_gev"]
    lam_r = np.sort(np.abs(gr[:, 1]) / np.maximum(np.abs(gr[:, 0]), 1e-300))
    finite = lam_r < 1e6
    assert_allclose(lam[finite], lam_r[finite], rtol=1e-6)


@pytest.mark.parametrize("key", KEYS)
def test_cycle_reduction_matches_reference(ref_goldens, key):
    A, B, C, D = _abcd(ref_goldens, key)
    T, R, msg, _ = oracle.solve_policy_function_with_cycle_reduction(A, B, C, D, max_iter=1000, tol=1e-8)
    assert msg == "Optimization successful"
    assert_allclose(T, ref_goldens[f"{key}_ref_cr_T"], atol=1e-13, rtol=0)
    assert_allclose(R, ref_goldens[f"{key}_ref_cr_R"], atol=1e-12, rtol=0)
    for tol, it_ref in zip(ref_goldens["cr_tols"], ref_goldens[f"{key}_ref_cr_iters"]):
        Tc, conv, it = oracle.cycle_reduction_core(A, B, C, 1000, float(tol))
        assert conv and it == it_ref
    assert oracle.policy_residual(A, B, C, T) < 1e-20


@pytest.mark.parametrize("key", KEYS)
def test_gensys_equals_cycle_reduction(ref_goldens, key):
    # reference tests/model/test_perturbation.py:163-206, atol = rtol = 1e-8
    A, B, C, D = _abcd(ref_goldens, key)
    Tg, ok, eu = oracle.gensys_T_success(A, B, C, D)
    Tc, conv, _ = oracle.cycle_reduction_core(A, B, C, 1000, 1e-8)
    assert ok and conv
    assert_allclose(Tg, Tc, atol=1e-8, rtol=1e-8)
    Rg = oracle.compute_selection_matrix(B, C, D, Tg)
    Rc = oracle.compute_selection_matrix(B, C, D, Tc)
    assert_allclose(Rg, Rc, atol=1e-8, rtol=1e-8)
    # jumper columns of T are zero, state columns are not
    state = np.abs(A).sum(axis=0) > 0
    assert np.abs(Tg[:, ~state]).max() < 1e-8
    assert np.all(np.abs(Tg[:, state]).sum(axis=0) > 1e-8)


def test_scan_cycle_reduction(ref_goldens):
    A, B, C, D = _abcd(ref_goldens, "full_nk")
    T, n_steps = oracle.scan_cycle_reduction(A, B, C, max_iter=50, tol=1e-7)
    assert_allclose(T, ref_goldens["full_nk_ref_cr_T"], atol=1e-9)
    assert 5 < n_steps < 20


def test_rbc_closed_form(rbc_golden):
    g = rbc_golden
    A, B, C, D = wl.rbc_linearized_jacobians(**wl.RBC_CALIBRATION)
    for x, M in zip("ABCD", (A, B, C, D)):
        assert_allclose(M, g[f"cal_{x}"], atol=0)
    Tg, ok, eu = oracle.gensys_T_success(A, B, C, D)
    assert ok and list(eu) == [1, 1, 0]
    assert_allclose(Tg, g["cal_ref_gensys_T"], atol=1e-12)
    # economic sanity: technology is AR(1) with rho_A, capital is the only other state
    assert_allclose(Tg[0, 0], wl.RBC_CALIBRATION["rho_A"], atol=1e-12)
    assert np.abs(Tg[:, [1, 2, 4, 5, 6, 7]]).max() < 1e-12


def test_rbc_draws(rbc_golden):
    g = rbc_golden
    th = {k[6:]: g[k] for k in g.files if k.startswith("theta_")}
    A, B, C, D = wl.rbc_linearized_jacobians(**th)
    for i in range(A.shape[0]):
        Tg, ok, eu = oracle.gensys_T_success(A[i], B[i], C[i], D[i])
        assert ok
        assert_allclose(Tg, g["ref_gensys_T"][i], atol=1e-11)
        Tc, conv, it = oracle.cycle_reduction_core(A[i], B[i], C[i], 1000, 1e-8)
        assert conv and it == g["ref_cr_iters"][i]
        assert_allclose(Tc, g["ref_cr_T"][i], atol=1e-12)


def test_sw_shaped(sw_golden):
    g = sw_golden
    nb = int(g["n_draws"])
    b = wl.sw_shaped_batch(nb)
    om = wl.sw_shaped_observation_model()
    chk = np.array([np.abs(b[x]).sum() for x in "ABCD"] + [np.abs(om["y"]).sum()])
    assert_allclose(chk, g["input_checksum"], rtol=1e-13)  # RNG stream has not drifted
    for i in range(nb):
        A, B, C, D = (b[x][i] for x in "ABCD")
        Tg, ok, eu = oracle.gensys_T_success(A, B, C, D)
        assert ok and list(eu) == list(g["ref_gensys_eu"][i])
        assert_allclose(Tg, g["ref_gensys_T"][i], atol=1e-11)
        assert_allclose(Tg, b["T_star"][i], atol=1e-11)
        Tc, conv, it = oracle.cycle_reduction_core(A, B, C, 1000, 1e-8)
        assert conv and it == g["ref_cr_iters"][i]
        assert_allclose(Tc, g["ref_cr_T"][i], atol=1e-12)


@pytest.mark.parametrize(
    "name,eu_expected", [("ok", [1, 1, 0]), ("nonunique", None), ("noexist", None), ("coincident", [-2, -2, 0])]
)
def test_failure_codes(failure_golden, name, eu_expected):
    g = failure_golden
    A, B, C, D = (g[f"{name}_{x}"] for x in "ABCD")
    Tg, ok, eu = oracle.gensys_T_success(A, B, C, D)
    assert list(eu) == list(g[f"{name}_ref_gensys_eu"])
    if eu_expected is not None:
        assert list(eu) == eu_expected
    assert ok == (name == "ok")
    if name == "nonunique":  # the class pinned at tests/model/test_model.py:514-529 (eu = [1, 0, k>0])
        assert eu[0] == 1 and eu[1] == 0 and eu[2] > 0
    if name == "noexist":
        assert eu[0] == 0
    if name == "coincident":
        assert np.all(Tg == 0)
        out = oracle.solve_policy_function_with_gensys(A, B, C, D)
        assert out[0] is None and out[7] == [-2, -2, 0]
    Tc, conv, _ = oracle.cycle_reduction_core(A, B, C, 1000, 1e-8)
    assert conv == bool(g[f"{name}_ref_cr_converged"][0])
    if not conv:
        assert np.all(Tc == 0)
        if name == "coincident":
            # exactly singular A1: np.linalg.solve raises, as it does in the reference's
            # numpy variant (cycle_reduction.py:88)
            with pytest.raises(np.linalg.LinAlgError):
                oracle.cycle_reduction_numpy(A, B, C, 100, 1e-8)
        else:
            X, res, msg, _ = oracle.cycle_reduction_numpy(A, B, C, 100, 1e-8)
            assert X is None and msg != "Optimization successful"


def test_backward_direct():
    rng = np.random.default_rng(3)
    n, k = 6, 2
    B = np.eye(n) + 0.1 * rng.standard_normal((n, n))
    A = 0.3 * rng.standard_normal((n, n))
    D = rng.standard_normal((n, k))
    T, R = oracle.solve_policy_function_with_backward_direct(A, B, np.zeros((n, n)), D)
    assert_allclose(A + B @ T, 0, atol=1e-13)
    assert_allclose(B @ R + D, 0, atol=1e-13)


def test_policy_adjoints_match_finite_differences(ref_goldens):
    """Analogue of the reference's verify_grad on the solver adjoints
    (tests/model/test_perturbation.py:209-276): <T_bar, dT> == <A_bar,dA> + <B_bar,dB> + <C_bar,dC>."""
    A, B, C, D = _abcd(ref_goldens, "one_block")
    rng = np.random.default_rng(0)
    T0, conv, _ = oracle.cycle_reduction_core(A, B, C, 1000, 1e-14)
    assert conv
    T_bar = rng.standard_normal(T0.shape)
    A_bar, B_bar, C_bar = oracle.policy_function_adjoints(A, B, C, T0, T_bar)
    for M, M_bar, which in ((A, A_bar, 0), (B, B_bar, 1), (C, C_bar, 2)):
        # directional derivative along a random direction supported on the non-zero pattern
        dM = rng.standard_normal(M.shape) * (M != 0)
        eps = 1e-6
        args_p = [A, B, C]
        args_m = [A, B, C]
        args_p[which] = M + eps * dM
        args_m[which] = M - eps * dM
        Tp, cp, _ = oracle.cycle_reduction_core(*args_p, 1000, 1e-14)
        Tm, cm, _ = oracle.cycle_reduction_core(*args_m, 1000, 1e-14)
        assert cp and cm
        fd = np.sum(T_bar * (Tp - Tm)) / (2 * eps)
        assert_allclose(np.sum(M_bar * dM), fd, rtol=1e-5, atol=1e-8)


def _bk_cases():
    import os

    from geconpy_amd import workloads as wl

    here = os.path.dirname(os.path.abspath(__file__))
    g = np.load(os.path.join(here, "golden", "bk_eigenvalues.npz"))
    rg = np.load(os.path.join(here, "golden", "reference_goldens.npz"))
    fg = np.load(os.path.join(here, "golden", "failure_cases.npz"))
    cases = {k: tuple(rg[f"{k}_{x}"] for x in "ABCD") for k in ("one_block", "rbc_2_block", "full_nk")}
    cases.update({k: tuple(fg[f"{k}_{x}"] for x in "ABCD") for k in ("ok", "nonunique", "noexist")})
    b = wl.sw_shaped_batch(2)
    cases.update({f"sw{i}": tuple(b[x][i] for x in "ABCD") for i in range(2)})
    return g, cases


def test_bk_eigenvalues_match_extracted_reference():
    """oracle.compute_bk_eigenvalues vs the reference's own compute_bk_eigenvalues (perturbation.py:412-445, executed by
    AST extraction in tests/golden/make_bk_golden.py): same LAPACK call, so the values agree to rounding; the BK verdicts
    agree with the eu codes of the same systems."""
    g, cases = _bk_cases()
    for name, (A, B, C, D) in cases.items():
        re, im, nf = oracle.compute_bk_eigenvalues(A, B, C, D, 1e-8)
        assert nf == int(g[f"{name}_n_forward"])
        assert_allclose(np.hypot(re, im), np.hypot(g[f"{name}_real"], g[f"{name}_imag"]), rtol=1e-9, atol=1e-12)
        ok, n_forward, n_unstable = oracle.check_bk_condition(A, B, C, D)
        assert ok == (name not in ("nonunique", "noexist"))
        if name == "nonunique":
            assert n_unstable < n_forward
        if name == "noexist":
            assert n_unstable > n_forward
