"""Models with 65 .. 96 variables (round 4, csrc/dsge_big.hpp): cycle reduction with one workgroup per draw, the selection matrix,
and the fused solve + Kalman log-likelihood whose filter runs on the model restricted to its state and observed variables.
The reference has no size limit (gEconpy/model/statespace.py:822-839); its cycle reduction is
gEconpy/solvers/cycle_reduction.py:127-183 (njit rule) / :246-294 (scan rule), restated in oracle/cycle_reduction.py."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import oracle
from geconpy_amd import _lib, batched
from geconpy_amd import workloads as wl

pytestmark = pytest.mark.gpu

SHAPES = {65: dict(n_state=28, n_lead=20, k=8), 72: dict(n_state=30, n_lead=20, k=8), 80: dict(n_state=36, n_lead=24, k=10),
          81: dict(n_state=36, n_lead=26, k=10), 96: dict(n_state=44, n_lead=30, k=12)}


def _systems(n, nb, seed0=7000):
    sh = SHAPES[n]
    sysm = [wl.sw_shaped_system(seed0 + i, n=n, n_state=sh["n_state"], n_lead=sh["n_lead"], k=sh["k"]) for i in range(nb)]
    return tuple(np.stack([s[j] for s in sysm]) for j in range(5))


@pytest.mark.parametrize("n", [65, 72, 80, 81, 96])
def test_big_cycle_reduction_and_selection_vs_oracle(n):
    """T, the iteration count and the status of `dsge_cycle_reduction_batched` against `oracle.cycle_reduction_core` (the njit
    rule), R and the policy residual of `dsge_selection_batched` against the oracle's selection matrix."""
    nb = 6
    A, B, C, D, Tst = _systems(n, nb)
    T, status, n_iter = batched.cycle_reduction_batched(A, B, C, max_iter=1000, tol=1e-9)
    assert np.all(status == 0)
    assert_allclose(T, Tst, atol=1e-8)
    R, resid = batched.selection_batched(B, C, D, T, A=A)
    for i in range(nb):
        Tc, conv, it = oracle.cycle_reduction_core(A[i], B[i], C[i], 1000, 1e-9)
        assert conv and it == n_iter[i]
        assert_allclose(T[i], Tc, atol=1e-9 * max(1.0, np.abs(Tc).max()))
        Rc = oracle.compute_selection_matrix(B[i], C[i], D[i], Tc)
        assert_allclose(R[i], Rc, atol=1e-8 * max(1.0, np.abs(Rc).max()))
        assert_allclose(resid[i], oracle.policy_residual(A[i], B[i], C[i], T[i]), rtol=1e-4, atol=1e-24)
        assert resid[i] < 1e-14
    # bit-identical repeats, and a draw's result does not depend on its place in the batch
    T2, status2, n_iter2 = batched.cycle_reduction_batched(A[::-1].copy(), B[::-1].copy(), C[::-1].copy(), max_iter=1000, tol=1e-9)
    assert np.array_equal(T2[::-1], T) and np.array_equal(n_iter2[::-1], n_iter)


@pytest.mark.parametrize("n", [72, 96])
def test_big_scan_cycle_reduction_vs_oracle(n):
    A, B, C, D, Tst = _systems(n, 4, seed0=7100)
    T, status, n_steps = batched.scan_cycle_reduction_batched(A, B, C, max_iter=50, tol=1e-8)
    assert np.all(status == 0)
    for i in range(4):
        Tc, steps = oracle.scan_cycle_reduction(A[i], B[i], C[i], max_iter=50, tol=1e-8)
        assert steps == n_steps[i]
        assert_allclose(T[i], Tc, atol=1e-9 * max(1.0, np.abs(Tc).max()))


def test_big_cycle_reduction_failure_codes():
    """A draw that cannot converge within max_iter and a draw with a NaN: status as the n <= 64 kernels report it, T = 0
    (cycle_reduction.py:176-181)."""
    n = 80
    A, B, C, D, _ = _systems(n, 3, seed0=7200)
    A[2, 3, 5] = np.nan
    T, status, n_iter = batched.cycle_reduction_batched(A, B, C, max_iter=2, tol=1e-12)
    assert status[0] == _lib.ST_NOT_CONVERGED and status[1] == _lib.ST_NOT_CONVERGED and n_iter[0] == 2
    assert status[2] == (_lib.ST_NOT_CONVERGED | _lib.ST_NAN) and n_iter[2] == 1
    assert np.all(T == 0.0)


@pytest.mark.parametrize("n,observed", [(72, None), (80, (40, 45, 50, 55, 60, 70, 79)), (96, (2, 50, 60, 70, 80, 90, 95))])
def test_big_solve_kalman_logp_vs_oracle(n, observed):
    """The fused evaluation A, B, C, D -> logp for n > 64 against `oracle.solve_kalman_logp` (full-size cycle reduction and
    full-size filter): 1e-9 relative, the contract of the n <= 64 path; observed states and observed jump variables; T, R, the
    residual and the iteration counts on request (explicit selection route) and not (R from the last elimination)."""
    sh = SHAPES[n]
    nb = 12
    b = wl.sw_shaped_batch(nb, n=n, p=7, T_len=60, **sh)
    om = wl.sw_shaped_observation_model(observed=observed, n=n, p=7, T_len=60, **sh)
    q = b["sigma"] ** 2
    r = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], om["y"], Hdiag=om["Hdiag"], tol=1e-8,
                                          max_iter=1000, q_mode="diag_batched")
    rp = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], q, om["Z"], om["y"], Hdiag=om["Hdiag"], tol=1e-8,
                                           max_iter=1000, return_policy=True, q_mode="diag_batched")
    assert np.all(r["status"] == 0) and np.all(rp["status"] == 0)
    for i in range(nb):
        ref = oracle.solve_kalman_logp(b["A"][i], b["B"][i], b["C"][i], b["D"][i], np.diag(q[i]), om["Z"], om["y"],
                                       H=np.diag(om["Hdiag"]), tol=1e-8, max_iter=1000)
        assert ref["success"]
        assert abs(r["logp"][i] - ref["logp"]) <= 1e-9 * abs(ref["logp"]), (i, r["logp"][i], ref["logp"])
        assert abs(rp["logp"][i] - ref["logp"]) <= 1e-9 * abs(ref["logp"])
        assert_allclose(rp["T"][i], ref["T"], atol=1e-9 * max(1.0, np.abs(ref["T"]).max()))
        assert_allclose(rp["R"][i], ref["R"], atol=1e-8 * max(1.0, np.abs(ref["R"]).max()))
        assert rp["n_iter"][i] == ref["n_iter"]
    # a dense design matrix (every observed series loads on three variables), a batched one, an intercept
    rng = np.random.default_rng(n)
    Zd = np.zeros((nb, 7, n))
    for i in range(nb):
        for s in range(7):
            Zd[i, s, rng.choice(n // 2, 3, replace=False)] = rng.uniform(0.5, 1.5, 3)
    dvec = rng.standard_normal(7) * 0.01
    rd = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], q, Zd, om["y"], d=dvec, Hdiag=om["Hdiag"], tol=1e-8,
                                           max_iter=1000, q_mode="diag_batched")
    assert np.all(rd["status"] == 0)
    for i in range(0, nb, 3):
        ref = oracle.solve_kalman_logp(b["A"][i], b["B"][i], b["C"][i], b["D"][i], np.diag(q[i]), Zd[i], om["y"],
                                       H=np.diag(om["Hdiag"]), d=dvec, tol=1e-8, max_iter=1000)
        assert abs(rd["logp"][i] - ref["logp"]) <= 1e-9 * abs(ref["logp"])


def test_big_failed_draw_and_too_many_filtered_variables():
    """A failed solve inside the fused call: logp = -inf and the solver's status for that draw only.  More than 64 state and
    observed variables: DSGE_ERR_TOO_LARGE (a return code), nothing computed.  The ordered QZ keeps n <= 64."""
    n = 80
    sh = SHAPES[n]
    b = wl.sw_shaped_batch(4, n=n, p=7, T_len=30, **sh)
    om = wl.sw_shaped_observation_model(n=n, p=7, T_len=30, **sh)
    A = b["A"].copy()
    A[1, 0, 0] = np.nan
    r = batched.solve_kalman_logp_batched(A, b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], om["y"], Hdiag=om["Hdiag"],
                                          tol=1e-8, max_iter=1000)
    assert r["status"][1] & _lib.ST_NAN and r["logp"][1] == -np.inf
    assert np.all(r["status"][[0, 2, 3]] == 0) and np.all(np.isfinite(r["logp"][[0, 2, 3]]))
    wide = wl.sw_shaped_batch(2, n=n, n_state=70, n_lead=6, k=8, p=7, T_len=30)
    omw = wl.sw_shaped_observation_model(n=n, n_state=70, n_lead=6, k=8, p=7, T_len=30)
    with pytest.raises(_lib.DsgeTooLargeError):
        batched.solve_kalman_logp_batched(wide["A"], wide["B"], wide["C"], wide["D"], wide["sigma"] ** 2, omw["Z"], omw["y"],
                                          Hdiag=omw["Hdiag"], tol=1e-8, max_iter=1000)
    # the ordered QZ keeps n + #lead <= 64 (gensys beyond that exists by spectral division only: test_big_gensys_*)
    with pytest.raises(_lib.DsgeHipError):
        batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], om["y"], Hdiag=om["Hdiag"],
                                          tol=1e-8, solver="gensys", options={"gensys_doubling": 0})
    with pytest.raises(_lib.DsgeHipError):
        batched.gensys_batched(b["A"], b["B"], b["C"], b["D"], tol=1e-8, options={"gensys_doubling": 0})


@pytest.mark.parametrize("n", [65, 72, 80, 96])
def test_big_gensys_by_spectral_division_vs_oracle(n):
    """solver = gensys beyond 64 variables (the reference has no size limit and gensys is its default, statespace.py:822-839):
    the doubling iteration + the certificate of eu = [1, 1, 0] (csrc/dsge_big.hpp: gensys_certify_big_kernel).  Regular draws:
    T, R, eu of the oracle's gensys (LAPACK's ordered QZ of the (n + #lead)-dimensional pencil); an explosive and an indeterminate
    draw: the oracle's gensys says eu != [1, 1], the device says 'no verdict at this size' -- a failed draw either way."""
    sh = SHAPES[n]
    nb = 10
    b = wl.sw_shaped_batch(nb, n=n, p=7, T_len=30, **sh)
    A, B, C, D = b["A"].copy(), b["B"].copy(), b["C"].copy(), b["D"]
    A[3] *= 25.0  # explosive: no stable solvent
    M = B[6] + C[6] @ b["T_star"][6]  # indeterminate: a root of the forward block pulled inside the unit circle
    G = np.linalg.solve(M, C[6])
    C[6] = M @ (G * (1.5 / np.max(np.abs(np.linalg.eigvals(G)))))
    B[6] = M - C[6] @ b["T_star"][6]
    SD = {"gensys_doubling": 1}  # (the library's default; explicit, so that the test means the same under DSGE_GENSYS_DOUBLING=0)
    out = batched.gensys_batched(A, B, C, D, tol=1e-8, options=SD)
    for i in range(nb):
        T_ref, succ, eu = oracle.gensys_T_success(A[i], B[i], C[i], D[i], 1e-8)
        assert bool(out["success"][i]) == bool(succ), (i, eu, out["eu"][i], out["status"][i])
        if succ:
            assert list(out["eu"][i]) == [1, 1, 0]
            assert_allclose(out["T"][i], T_ref, rtol=0, atol=1e-9 * max(1.0, np.abs(T_ref).max()))
            R_ref = -np.linalg.solve(C[i] @ T_ref + B[i], D[i])
            assert_allclose(out["R"][i], R_ref, rtol=0, atol=1e-9 * max(1.0, np.abs(R_ref).max()))
        else:
            assert i in (3, 6)
            assert list(out["eu"][i]) == [-3, -3, 0] and out["status"][i] & _lib.ST_GENSYS_TOO_BIG
            assert not out["T"][i].any() and not out["R"][i].any()
    # the fused evaluation against the oracle's, same solver
    om = wl.sw_shaped_observation_model(n=n, p=7, T_len=30, **sh)
    q = b["sigma"] ** 2
    f = batched.solve_kalman_logp_batched(A, B, C, D, q, om["Z"], om["y"], Hdiag=om["Hdiag"], tol=1e-8, solver="gensys",
                                          q_mode="diag_batched", options=SD)
    assert np.array_equal(f["status"] == 0, out["success"]) and np.all(f["logp"][[3, 6]] == -np.inf)
    for i in (0, 1, 5, 9):
        ref = oracle.solve_kalman_logp(A[i], B[i], C[i], D[i], np.diag(q[i]), om["Z"], om["y"], H=np.diag(om["Hdiag"]), solver="gensys")
        assert abs(f["logp"][i] - ref["logp"]) <= 1e-8 * abs(ref["logp"]), (i, f["logp"][i], ref["logp"])
    # same T as the cycle-reduction solver's (bit for bit: the same kernel computed it)
    T_cr = batched.cycle_reduction_batched(A, B, C, tol=1e-9, max_iter=50)[0]
    ok = out["success"]
    assert np.array_equal(T_cr[ok], out["T"][ok])


def test_big_host_chunks_and_reference_default_gating():
    """The host twin stages a batch >= 512 in chunks over two streams (each chunk measures its own filtered variables): same
    logp to the bit as the draws evaluated in a small batch.  `add_solver_success_check=False` (the reference's default graph,
    statespace.py:1148): a failed cycle reduction carries T = 0 on and the draw gets the finite log-likelihood of that system,
    as on the n <= 64 path."""
    n = 72
    sh = SHAPES[n]
    nd = 40
    b = wl.sw_shaped_batch(nd, n=n, p=7, T_len=40, **sh)
    om = wl.sw_shaped_observation_model(n=n, p=7, T_len=40, **sh)
    rep = 16
    big = {x: np.tile(b[x], (rep, 1, 1)) for x in "ABCD"}
    qb = np.tile(b["sigma"] ** 2, (rep, 1))
    small = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], om["y"],
                                              Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000)
    large = batched.solve_kalman_logp_batched(big["A"], big["B"], big["C"], big["D"], qb, om["Z"], om["y"], Hdiag=om["Hdiag"],
                                              tol=1e-8, max_iter=1000)
    assert np.all(small["status"] == 0) and np.all(large["status"] == 0)
    assert np.array_equal(large["logp"], np.tile(small["logp"], rep))
    # max_iter = 2: every solve fails; with the reference's default gating the filter runs on T = 0, R = 0
    gated = batched.solve_kalman_logp_batched(b["A"][:4], b["B"][:4], b["C"][:4], b["D"][:4], b["sigma"][:4] ** 2, om["Z"], om["y"],
                                              Hdiag=om["Hdiag"], tol=1e-12, max_iter=2)
    assert np.all(gated["status"] == _lib.ST_NOT_CONVERGED) and np.all(gated["logp"] == -np.inf)
    ungated = batched.solve_kalman_logp_batched(b["A"][:4], b["B"][:4], b["C"][:4], b["D"][:4], b["sigma"][:4] ** 2, om["Z"],
                                                om["y"], Hdiag=om["Hdiag"], tol=1e-12, max_iter=2, add_solver_success_check=False)
    assert np.all(ungated["status"] == _lib.ST_NOT_CONVERGED) and np.all(np.isfinite(ungated["logp"]))
    ref = oracle.solve_kalman_logp(b["A"][0], b["B"][0], b["C"][0], b["D"][0], np.diag(b["sigma"][0] ** 2), om["Z"], om["y"],
                                   H=np.diag(om["Hdiag"]), tol=1e-12, max_iter=2, add_solver_success_check=False)
    assert abs(ungated["logp"][0] - ref["logp"]) <= 1e-9 * abs(ref["logp"])


def test_big_standalone_filter_on_the_restricted_model():
    """`kalman_logp_batched` with more than 64 variables: the numpy front end restricts (T, R, Z) to the state and observed
    variables (exact-zero columns of T only) and the device filters that model: same logp as the oracle's full-size filter."""
    n = 80
    sh = SHAPES[n]
    nb = 4
    b = wl.sw_shaped_batch(nb, n=n, p=7, T_len=50, **sh)
    om = wl.sw_shaped_observation_model(observed=(1, 5, 40, 50, 60, 70, 79), n=n, p=7, T_len=50, **sh)
    T, status, _ = batched.cycle_reduction_batched(b["A"], b["B"], b["C"], max_iter=1000, tol=1e-10)
    assert np.all(status == 0)
    assert np.count_nonzero(np.any(T != 0.0, axis=(0, 1))) == sh["n_state"]  # the solver writes exact zeros in the other columns
    R = batched.selection_batched(b["B"], b["C"], b["D"], T)
    q = b["sigma"] ** 2
    logp, st = batched.kalman_logp_batched(T, R, q, om["Z"], om["y"], Hdiag=om["Hdiag"])
    assert np.all(st == 0)
    for i in range(nb):
        ref = oracle.kalman_filter_logp(om["y"], T[i], R[i], np.diag(q[i]), om["Z"], H=np.diag(om["Hdiag"]))
        assert abs(logp[i] - ref) <= 1e-9 * abs(ref)
    noisy = T + 1e-18
    with pytest.raises(_lib.DsgeTooLargeError):
        batched.kalman_logp_batched(noisy, R, q, om["Z"], om["y"], Hdiag=om["Hdiag"])


def test_big_device_entry_chunks_and_stage_timing():
    """The device-pointer entry with `pipeline_chunks` (chunks alternating over library-owned streams, each with its own slice
    of the scratch arena and its own workspace) and the stage-timing entry (`LogpEngine.profile_kernels`) on a 72-variable
    model: same logp to the bit as the plain call."""
    import torch

    from geconpy_amd.engine import LogpEngine

    n = 72
    sh = SHAPES[n]
    nd = 48
    b = wl.sw_shaped_batch(nd, n=n, p=7, T_len=30, **sh)
    om = wl.sw_shaped_observation_model(n=n, p=7, T_len=30, **sh)
    rep = 25  # 1200 draws
    eng = LogpEngine(0)
    A, B, C, D = (eng.to_device(np.tile(b[x], (rep, 1, 1))) for x in "ABCD")
    q = eng.to_device(np.tile(b["sigma"] ** 2, (rep, 1)))
    Z, y, H = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
    lp0, st0 = eng.solve_kalman_logp(A, B, C, D, q, Z, y, Hdiag=H, q_mode=1, tol=1e-8, max_iter=1000, z_selector_hint=1)
    lp1, st1 = eng.solve_kalman_logp(A, B, C, D, q, Z, y, Hdiag=H, q_mode=1, tol=1e-8, max_iter=1000, z_selector_hint=1,
                                     options={"pipeline_chunks": 2})
    torch.cuda.synchronize()
    assert int((st0 != 0).sum()) == 0 and int((st1 != 0).sum()) == 0
    assert torch.equal(lp0, lp1)
    ref = batched.solve_kalman_logp_batched(b["A"][:4], b["B"][:4], b["C"][:4], b["D"][:4], b["sigma"][:4] ** 2, om["Z"], om["y"],
                                            Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000)
    assert np.array_equal(lp0[:4].cpu().numpy(), ref["logp"])
    ms = eng.profile_kernels(A, B, C, D, q, Z, y, Hdiag=H, q_mode=1, tol=1e-8, max_iter=1000, reps=1, z_selector_hint=1)
    assert ms["solver"] > 0.0 and ms["kalman"] > 0.0


def test_big_numpy_entry_points_with_the_reference_names():
    """`solve_policy_function_with_cycle_reduction` (cycle_reduction.py:328-398) and the batched solvability driver on an
    80-variable system: the reference's return shapes and message, T and R against the oracle."""
    from geconpy_amd import solvers

    n = 80
    A, B, C, D, _ = _systems(n, 3, seed0=7300)
    T, R, msg, _log_norm = solvers.solve_policy_function_with_cycle_reduction(A[0], B[0], C[0], D[0], max_iter=1000, tol=1e-9,
                                                                             verbose=False)
    Tc, Rc, msg_c, _ = oracle.solve_policy_function_with_cycle_reduction(A[0], B[0], C[0], D[0], max_iter=1000, tol=1e-9)
    assert msg == msg_c == solvers.MSG_OK
    assert_allclose(T, Tc, atol=1e-9 * max(1.0, np.abs(Tc).max()))
    assert_allclose(R, Rc, atol=1e-8 * max(1.0, np.abs(Rc).max()))
    out = solvers.solve_policy_functions_batched(A, B, C, D, solver="cycle_reduction", max_iter=1000, tol=1e-9)
    assert out["success"].all() and np.all(out["resid"] < 1e-14)
    assert_allclose(out["T"][0], Tc, atol=1e-9 * max(1.0, np.abs(Tc).max()))
