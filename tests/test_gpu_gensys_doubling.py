"""dsge_options.gensys_doubling (csrc/dsge_gensys_doubling.hpp): gensys by spectral division -- the doubling iteration computes the
solvent, a per-draw certificate (rho(T[S,S]) < 1 and rho(((B + C T)^-1 C)[L,L]) < 1) stands for eu = [1, 1, 0], the ordered QZ
decides every draw without one.  Outputs must be those of the QZ path and of the reference's gensys: eu EXACT on every system
(regular, indeterminate, explosive, coincident zeros, near-unit roots), T to the cross-solver accuracy."""
import os
import sys

import numpy as np
import pytest
from numpy.testing import assert_allclose

import oracle
from geconpy_amd import _lib, batched
from geconpy_amd import workloads as wl

pytestmark = pytest.mark.gpu
DBL = {"gensys_doubling": 1}  # (the library's default since round 5; explicit here)
QZ = {"gensys_doubling": 0}   # the ordered QZ for every draw


def _stack(g, keys):
    return tuple(np.stack([g[f"{k}_{x}"] for k in keys]) for x in "ABCD")


@pytest.mark.parametrize("key", ["one_block", "rbc_2_block", "full_nk"])
def test_reference_goldens(ref_goldens, key):
    g = ref_goldens
    A, B, C, D = (g[f"{key}_{x}"][None] for x in "ABCD")
    out = batched.gensys_batched(A, B, C, D, tol=1e-8, options=DBL)
    assert list(out["eu"][0]) == [1, 1, 0] and out["success"][0]
    assert_allclose(out["T"][0], g[f"{key}_ref_gensys_T"], atol=1e-10)
    assert_allclose(out["R"][0], g[f"{key}_ref_gensys_R"], atol=1e-9)


def test_failure_codes_are_the_qz_s(failure_golden):
    """ok / non-unique / no solution / coincident zeros of tests/golden/failure_cases.npz (labelled by the reference's own core): the
    certificate must refuse the three non-regular systems -- the non-unique one CONVERGES in cycle reduction (to the minimal solvent)
    -- and the QZ's verdicts come back exactly; the regular one keeps the doubling iteration's T."""
    g = failure_golden
    names = ["ok", "nonunique", "noexist", "coincident"]
    A, B, C, D = _stack(g, names)
    qz = batched.gensys_batched(A, B, C, D, tol=1e-8, options=QZ)
    out = batched.gensys_batched(A, B, C, D, tol=1e-8, options=DBL)
    for i, name in enumerate(names):
        assert list(out["eu"][i]) == list(g[f"{name}_ref_gensys_eu"]), name
        assert out["success"][i] == (name == "ok")
    assert np.array_equal(out["status"], qz["status"])
    assert_allclose(out["T"][0], g["ok_ref_gensys_T"], atol=1e-9)
    assert_allclose(out["T"][1:], qz["T"][1:], atol=1e-12)  # (the same kernel decided them)
    om = wl.sw_shaped_observation_model()
    q = np.full(7, 1e-4)
    f = batched.solve_kalman_logp_batched(A, B, C, D, q, om["Z"], om["y"], Hdiag=om["Hdiag"], solver="gensys", tol=1e-8, options=DBL)
    assert np.isfinite(f["logp"][0]) and np.all(f["logp"][1:] == -np.inf)
    ref = oracle.solve_kalman_logp(A[0], B[0], C[0], D[0], np.diag(q), om["Z"], om["y"], H=np.diag(om["Hdiag"]), solver="gensys")
    assert_allclose(f["logp"][0], ref["logp"], rtol=1e-8)


def test_sw_shaped_draws_fused_and_standalone():
    """512 SW-shaped draws incl. the nearly singular draw 752 region (first_draw = 512): T and eu against the QZ path, logp of the fused
    evaluation against the oracle's gensys; an explosive and an indeterminate draw planted in the batch."""
    nb = 512
    b = wl.sw_shaped_batch(nb, first_draw=512)
    om = wl.sw_shaped_observation_model()
    A, B, C = b["A"].copy(), b["B"].copy(), b["C"].copy()
    A[7] *= 30.0  # explosive: no stable solution
    M = B[11] + C[11] @ b["T_star"][11]
    G = np.linalg.solve(M, C[11])
    G2 = G * (1.5 / np.max(np.abs(np.linalg.eigvals(G))))  # one of the "unstable" roots becomes stable: indeterminacy
    C[11] = M @ G2
    B[11] = M - C[11] @ b["T_star"][11]
    # ... and a REGULAR draw that the certificate cannot take (a stable root at 1 - 1e-5): the QZ solves it, and the fused call must
    # take its R from the explicit selection (the cycle reduction's R belongs to certified draws only)
    T20 = b["T_star"][20].copy()
    T20[:, :18] *= (1.0 - 1e-5) / np.max(np.abs(np.linalg.eigvals(T20[:18, :18])))
    M20 = B[20] + C[20] @ b["T_star"][20]
    A[20] = -M20 @ T20
    B[20] = M20 - C[20] @ T20
    qz = batched.gensys_batched(A, B, C, b["D"], tol=1e-8, options=QZ)
    out = batched.gensys_batched(A, B, C, b["D"], tol=1e-8, options=DBL)
    assert np.array_equal(out["eu"], qz["eu"]) and np.array_equal(out["status"], qz["status"])
    assert not out["success"][7] and not out["success"][11] and out["success"][20] and out["success"].sum() == nb - 2
    assert_allclose(out["T"][20], qz["T"][20], rtol=0, atol=1e-9)  # (the QZ of the rescue pass vs the QZ of the window path)
    ok = out["success"]
    # draw 240 of this batch is draw 752 of the bench batch: cond(B + C T) = 3e8 -- two float64 algorithms agree to 1e-7 there (the
    # reference's own cross-solver test asks 1e-8 on well-conditioned models, tests/model/test_perturbation.py:205-206)
    dT = np.abs(out["T"] - qz["T"]).reshape(nb, -1).max(axis=1)
    dR = np.abs(out["R"] - qz["R"]).reshape(nb, -1).max(axis=1)
    regular = ok & (np.arange(nb) != 240)
    assert dT[regular].max() <= 1e-9 and dR[regular].max() <= 1e-8, (int(dT[regular].argmax()), dT[regular].max())
    assert dT[240] <= 1e-6
    q = b["sigma"] ** 2
    f_qz = batched.solve_kalman_logp_batched(A, B, C, b["D"], q, om["Z"], om["y"], Hdiag=om["Hdiag"], solver="gensys", tol=1e-8, options=QZ)
    f = batched.solve_kalman_logp_batched(A, B, C, b["D"], q, om["Z"], om["y"], Hdiag=om["Hdiag"], solver="gensys", tol=1e-8, options=DBL)
    assert np.array_equal(f["status"] != 0, f_qz["status"] != 0)
    assert_allclose(f["logp"][ok], f_qz["logp"][ok], rtol=5e-9)
    assert np.isfinite(f["logp"][20]) and abs(f["logp"][20] - f_qz["logp"][20]) <= 1e-8 * abs(f_qz["logp"][20])
    for i in (0, 240, 300):  # (240 = draw 752 of the bench batch)
        ref = oracle.solve_kalman_logp(A[i], B[i], C[i], b["D"][i], np.diag(q[i]), om["Z"], om["y"], H=np.diag(om["Hdiag"]), solver="gensys",
                                       tol=1e-8)
        assert_allclose(f["logp"][i], ref["logp"], rtol=1e-8)


def test_roots_near_the_unit_circle_go_to_the_qz():
    """A stable root at 1 - 1e-5 and an 'unstable' one at 1 + 1e-5: gensys's counts are strict comparisons, the certificate gives up
    within its 12 squarings (it certifies rho < 0.99983 only) and the QZ decides -- same eu as the oracle either way."""
    n, ns, nl, k = 12, 5, 4, 3
    sysm = []
    for seed, (rho_s, rho_g) in enumerate([(1.0 - 1e-5, 0.5), (0.6, 1.0 / (1.0 + 1e-5)), (1.0 + 1e-5, 0.5), (0.6, 1.0 + 1e-5)]):
        A, B, C, D, Tst = wl.sw_shaped_system(900 + seed, n=n, n_state=ns, n_lead=nl, k=k)
        M = B + C @ Tst
        S = Tst[:ns, :ns]
        T2 = Tst.copy()
        T2[:, :ns] *= rho_s / np.max(np.abs(np.linalg.eigvals(S)))
        G = np.linalg.solve(M, C)
        G2 = G * (rho_g / np.max(np.abs(np.linalg.eigvals(G))))
        C2 = M @ G2
        sysm.append((-M @ T2, M - C2 @ T2, C2, D))
    A, B, C, D = (np.stack([s_[j] for s_ in sysm]) for j in range(4))
    out = batched.gensys_batched(A, B, C, D, tol=1e-8, options=DBL)
    for i in range(4):
        _, succ, eu = oracle.gensys_T_success(A[i], B[i], C[i], D[i], 1e-8)
        assert list(out["eu"][i][:2]) == [int(eu[0]), int(eu[1])], (i, out["eu"][i], eu)
        assert bool(out["success"][i]) == bool(succ)


def _pad_batch(systems):
    """Stack systems of ONE size (A, B, C, D per entry)."""
    return tuple(np.stack([s_[j] for s_ in systems]) for j in range(4))


@pytest.mark.parametrize("kind", ["row", "global", "col"])
def test_scale_defects_get_the_reference_verdict(kind):
    """ADVICE r5 / VERDICT r5 weak #2: one equation multiplied by 1e-11 .. 1e-5 (or every equation, or one variable's columns) leaves
    the solvent T untouched, yet the reference -- whose zxz and rank tests are ABSOLUTE tolerances, gensys.py:243, 276-283 -- rejects
    the system once the scale is below ~tol.  The certificate's scale guards hand such draws to the ordered QZ.  Asserted:
      * a system the reference rejects with its margins a factor 30 clear of the tolerance is rejected here (the round-5 certificate
        accepted it: the T of the doubling iteration does not see the scale);
      * a system it accepts a factor 30 clear is accepted, with the same T;
      * in between -- sigma_min(Q2 pi) or the smallest diagonal pair within a factor 30 of tol -- the verdict is the device QZ's and may
        differ from LAPACK's: both depend on the order in which the respective QZ leaves the eigenvalues on the diagonal
        (tests/device_models/scale_cases.py::lapack_margins).  Those systems are counted, not asserted; what the certificate itself
        may accept there is pinned by the order-independent bounds of tests/test_device_models.py."""
    from tests.device_models.scale_cases import lapack_margins, scaled_system

    rng = np.random.default_rng({"row": 81, "global": 82, "col": 83}[kind])
    n_clear_fail = n_clear_ok = n_grey = n_grey_differs = n_col_rejected = 0
    for n in (8, 14, 20):
        systems = [scaled_system(rng, kind, n=n)[:4] for _ in range(48)]
        by_k = {}  # (scaled_system draws n_state, n_lead, k at random: group by k for stacking)
        for s_ in systems:
            by_k.setdefault(s_[3].shape[1], []).append(s_)
        for group in by_k.values():
            A, B, C, D = _pad_batch(group)
            out = batched.gensys_batched(A, B, C, D, tol=1e-8, options=DBL)
            for i in range(len(group)):
                T_ref, succ, eu = oracle.gensys_T_success(A[i], B[i], C[i], D[i], 1e-8)
                sv_min, min_pair = lapack_margins(A[i], B[i], C[i], D[i], 1e-8)
                margin = min(sv_min, min_pair) / 1e-8
                if succ and margin >= 30.0:
                    n_clear_ok += 1
                    if kind == "col":
                        # a rescaled VARIABLE: its pair has beta = 0 (non-state) and an alpha that is tiny or not depending on where the
                        # QZ leaves it on the diagonal -- LAPACK's order keeps it above tol on systems where the device's order
                        # (static columns deflated first) does not: a known, order-dependent difference of the two QZs, counted
                        n_col_rejected += not out["success"][i]
                        if not out["success"][i]:
                            continue
                    assert out["success"][i] and list(out["eu"][i]) == [1, 1, 0], (kind, n, i, out["eu"][i], margin)
                    # (cond(B + C T) grows with 1 / scale: two float64 solvers agree to cond x eps)
                    assert_allclose(out["T"][i], T_ref, rtol=0, atol=1e-6 * max(1.0, np.abs(T_ref).max()))
                elif not succ and margin <= 1.0 / 30.0:
                    n_clear_fail += 1
                    assert not out["success"][i], (kind, n, i, out["eu"][i], eu, margin)
                else:
                    n_grey += 1
                    n_grey_differs += bool(out["success"][i]) != bool(succ)
    print(f"{kind}: clearly regular {n_clear_ok}, clearly rejected {n_clear_fail}, within a factor 30 of the tolerance {n_grey} "
          f"(verdict differs from LAPACK's on {n_grey_differs}); rescaled variables rejected by the device's order: {n_col_rejected}")
    assert n_clear_ok >= 20 and (kind == "col" or n_clear_fail >= 10)


@pytest.mark.parametrize("tol", [1e-8, 1e-6])
def test_existence_and_lead_column_tolerance_sweeps(tol):
    """sigma_min(Q2 pi) swept through realsmall from 1e-4 below to 1e4 above (the construction hits the target to 1e-3: the
    singular values of Q2 pi are 1 / sqrt(1 + sigma_i(N_L)^2), tests/test_device_models.py), and a lead column of C with
    sum|C_ij| from 1e-3 tol to 1e3 tol (gensys.py:587): eu identical to the oracle's at every point, T where it succeeds."""
    from tests.device_models.scale_cases import existence_sweep_system, lead_column_sweep_system

    factors = [1e-4, 1e-2, 0.3, 0.7, 1.4, 3.0, 1e2, 1e4]  # (0.95 / 1.05 are the CPU model's: there two QZs may differ by rounding)
    systems = [existence_sweep_system(seed, f * tol)[:4] for f in factors for seed in (7001, 7002, 7003)]
    A, B, C, D = _pad_batch(systems)
    out = batched.gensys_batched(A, B, C, D, tol=tol, options=DBL)
    for i in range(len(systems)):
        T_ref, succ, eu = oracle.gensys_T_success(A[i], B[i], C[i], D[i], tol)
        assert list(out["eu"][i]) == [int(e) for e in eu], (tol, factors[i // 3], out["eu"][i], eu)
        if succ:
            assert_allclose(out["T"][i], T_ref, rtol=0, atol=1e-7 * max(1.0, np.abs(T_ref).max()))
    assert out["success"][: 3 * 4].sum() == 0 and out["success"][-3 * 3 :].all()
    cf = [1e-3, 0.5, 0.999, 1.001, 2.0, 1e3]
    systems = [lead_column_sweep_system(seed, f * tol) for f in cf for seed in (7101, 7102)]
    A, B, C, D = _pad_batch(systems)
    out = batched.gensys_batched(A, B, C, D, tol=tol, options=DBL)
    for i in range(len(systems)):
        T_ref, succ, eu = oracle.gensys_T_success(A[i], B[i], C[i], D[i], tol)
        assert list(out["eu"][i]) == [int(e) for e in eu], (tol, cf[i // 2], out["eu"][i], eu)


def test_fuzz_against_the_oracle():
    """tools/fuzz_gensys.py (random sizes, structures, explosive draws) with the option on: eu exact, T at the suite's bar."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_gensys

    with _lib.options_scope(DBL):
        assert fuzz_gensys.run(7101, 150, verbose=True) == 0


def test_full_headline_batch_spectral_division_vs_qz():
    """BASELINE configs[2] at full size (4096 distinct SW-shaped draws, T = 200): the fused evaluation with gensys by spectral
    division against the same call with the ordered QZ for every draw -- identical status words, logp within the north star's 1e-8
    (measured: 6.1e-10 on the nearly singular draw 752, 3e-16 in the median), and no draw without a certificate in this batch."""
    nb = 4096
    b = wl.sw_shaped_batch(nb)
    om = wl.sw_shaped_observation_model()
    q = b["sigma"] ** 2
    kw = dict(Hdiag=om["Hdiag"], solver="gensys", tol=1e-8, return_policy=True)
    f_qz = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], om["y"], options=QZ, **kw)
    f_sd = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], om["y"], options=DBL, **kw)
    assert np.array_equal(f_qz["status"], f_sd["status"]) and np.all(f_sd["status"] == 0)
    rel = np.abs(f_sd["logp"] - f_qz["logp"]) / np.abs(f_qz["logp"])
    assert rel.max() <= 1e-8 and np.median(rel) <= 1e-14, (rel.max(), int(rel.argmax()), np.median(rel))
    dT = np.abs(f_sd["T"] - f_qz["T"]).reshape(nb, -1).max(axis=1)
    assert np.quantile(dT, 0.99) <= 1e-10 and dT.max() <= 1e-6, (dT.max(), int(dT.argmax()))
