"""kalman_nt2_kernel (csrc/dsge_kalman_nt2.hpp): the two-wavefront filter kernel for the head of the dispatch order.  Same recursion
as kalman_nt_kernel written for the PREDICTED covariance (P' = Tc P Tc' - (Tc K)(Tc V)' + jit_P Tc Tc' + Q), the measurement update on
one wavefront next to the two products on the other.  Checked against the oracle, against the one-wavefront kernel, with the head /
bulk split on two streams (dsge_options.kalman_head_draws), on the 8-, 24- and 32-wide tiles."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import oracle
from geconpy_amd import batched
from geconpy_amd import workloads as wl

pytestmark = pytest.mark.gpu
LOGP_RTOL = 1e-9


def _fused(b, om, y=None, **options):
    return batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], om["y"] if y is None else y,
                                             Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000, options=options or None)


@pytest.mark.parametrize("observed", [None, wl.SW_OBSERVED_JUMPS], ids=["states_24wide", "jumps_32wide"])
def test_two_wavefront_kernel_matches_oracle_and_one_wavefront_kernel(observed):
    nb = 96
    b = wl.sw_shaped_batch(nb)
    om = wl.sw_shaped_observation_model(observed=observed)
    y = om["y"].copy()
    y[17, 2] = np.nan
    y[60:64] = np.nan
    y[150:, 5] = oracle.MISSING_FILL  # a series that stops: a second fixed point under another mask
    one = _fused(b, om, y)
    two = _fused(b, om, y, kalman_head_draws=-1)
    assert np.all(one["status"] == 0) and np.all(two["status"] == 0)
    assert_allclose(two["logp"], one["logp"], rtol=1e-11)
    for i in (0, 5, 31, 95):
        ref = oracle.solve_kalman_logp(b["A"][i], b["B"][i], b["C"][i], b["D"][i], np.diag(b["sigma"][i] ** 2), om["Z"], y,
                                       H=np.diag(om["Hdiag"]), tol=1e-8, max_iter=1000)
        assert_allclose(two["logp"][i], ref["logp"], rtol=LOGP_RTOL)
    # the full recursion (no steady-state switch): every step through the two-wavefront full step
    two_full = _fused(b, om, y, kalman_head_draws=-1, kalman_steady_tol=0.0)
    assert_allclose(two_full["logp"], one["logp"], rtol=1e-11)


def test_head_and_bulk_on_two_streams_cover_every_draw_once():
    """A batch large enough for the dispatch order (>= 512 draws): the first `head` entries of the order go to the two-wavefront
    kernel on the library's second stream, the rest to the one-wavefront kernel on the caller's; every draw is written exactly
    once and the result does not depend on where the cut is."""
    nb = 640
    b = wl.sw_shaped_batch(nb)
    om = wl.sw_shaped_observation_model()
    base = _fused(b, om)
    assert np.all(base["status"] == 0)
    for head in (1, 64, 300, 639, 640, 5000):
        r = _fused(b, om, kalman_head_draws=head)
        assert np.all(r["status"] == 0), head
        assert_allclose(r["logp"], base["logp"], rtol=1e-11, err_msg=f"head={head}")
    # a failed solve in the batch: its draw gets -inf from whichever kernel it lands in
    A = b["A"].copy()
    A[3] *= 50.0  # no stable solution: cycle reduction fails
    bad = batched.solve_kalman_logp_batched(A, b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], om["y"], Hdiag=om["Hdiag"],
                                            tol=1e-8, max_iter=60, options={"kalman_head_draws": 128})
    assert bad["status"][3] != 0 and bad["logp"][3] == -np.inf
    ok = np.arange(nb) != 3
    assert_allclose(bad["logp"][ok], base["logp"][ok], rtol=1e-11)


def test_two_wavefront_kernel_small_model_and_hint_violation():
    """The RBC model (8-wide tile) through the two-wavefront kernel, and a draw whose state count exceeds the hint: flagged by the
    kernel and re-run by the cascade behind it, as with the one-wavefront kernel."""
    rng = np.random.default_rng(0)
    nb = 16
    th = wl.rbc_prior_draws(nb, seed=4)
    A, B, C, D = wl.rbc_linearized_jacobians(**th)
    q = (th["sigma_A"] ** 2)[:, None]
    Z = np.zeros((2, 8))
    Z[0, wl.RBC_VARIABLES.index("Y")] = 1.0
    Z[1, wl.RBC_VARIABLES.index("C")] = 0.5
    y = rng.normal(0, 0.05, (80, 2))
    y[7, 0] = np.nan
    kw = dict(d=np.array([0.01, -0.02]), Hdiag=np.array([1e-4, 2e-4]), tol=1e-12, max_iter=500, q_mode="diag_batched")
    one = batched.solve_kalman_logp_batched(A, B, C, D, q, Z, y, options={"kalman_tiny": 0}, **kw)
    two = batched.solve_kalman_logp_batched(A, B, C, D, q, Z, y, options={"kalman_tiny": 0, "kalman_head_draws": -1}, **kw)
    assert np.all(two["status"] == 0)
    assert_allclose(two["logp"], one["logp"], rtol=1e-11)
    for i in (0, 9):
        ref = oracle.solve_kalman_logp(A[i], B[i], C[i], D[i], np.diag(q[i]), Z, y, H=np.diag(kw["Hdiag"]), d=kw["d"], tol=1e-12,
                                       max_iter=500)
        assert_allclose(two["logp"][i], ref["logp"], rtol=LOGP_RTOL)
    # hint violation on the SW-shaped model: one draw with an extra state column
    b = wl.sw_shaped_batch(8)
    om = wl.sw_shaped_observation_model()
    T = np.stack([oracle.cycle_reduction_core(b["A"][i], b["B"][i], b["C"][i], 1000, 1e-12)[0] for i in range(8)])
    R = np.stack([oracle.compute_selection_matrix(b["B"][i], b["C"][i], b["D"][i], T[i]) for i in range(8)])
    T[5][:, 25] = 0.01 * rng.standard_normal(40)  # a 19th state column
    lp, st = batched.kalman_logp_batched(T, R, b["sigma"] ** 2, om["Z"], om["y"], Hdiag=om["Hdiag"], q_mode="diag_batched",
                                         n_state_hint=18, options={"kalman_head_draws": -1})
    assert np.all(st == 0)
    for i in (4, 5):
        ref = oracle.kalman_filter_logp(om["y"], T[i], R[i], np.diag(b["sigma"][i] ** 2), om["Z"], H=np.diag(om["Hdiag"]))
        assert_allclose(lp[i], ref, rtol=LOGP_RTOL)


@pytest.mark.parametrize("observed", [None, wl.SW_OBSERVED_JUMPS], ids=["states_24wide", "jumps_32wide"])
def test_tail_hand_off_from_the_fast_kernel_four_steps_per_trip(observed):
    """dsge_options.kalman_block = 1 with the NT fast kernel (round 5): once the covariance is frozen and the missing-data mask no
    longer changes, kalman_nt_kernel writes a record and kalman_tail4_kernel finishes the sample four steps per trip (32 + m affine
    forms of the state and the data, one per lane).  Same recursion: logp against the in-kernel single-step loop and the oracle;
    a mask that changes late in the sample (the hand-off happens behind it), conventions other than the default, the head /
    bulk split on top."""
    nb = 96
    b = wl.sw_shaped_batch(nb)
    om = wl.sw_shaped_observation_model(observed=observed)
    y = om["y"].copy()
    y[17, 2] = np.nan
    y[60:64] = np.nan
    y[120:, 5] = oracle.MISSING_FILL  # from here on the mask is constant: one series missing
    d = np.random.default_rng(4).normal(0, 0.01, 7)

    def run(**options):
        return batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], y, d=d, Hdiag=om["Hdiag"],
                                                 tol=1e-8, max_iter=1000, options=options or None)

    base = run()
    for opts in ({"kalman_block": 1}, {"kalman_block": 2, "kalman_nt_products": 0}, {"kalman_block": 1, "kalman_head_draws": -1},
                 {"kalman_block": 1, "mask_d": 1, "ll_constant": 1}):
        ref = run(**{k_: v for k_, v in opts.items() if k_ in ("mask_d", "ll_constant")}) if "mask_d" in opts else base
        r = run(**opts)
        assert np.all(r["status"] == 0), opts
        assert_allclose(r["logp"], ref["logp"], rtol=1e-11, err_msg=str(opts))
    for i in (0, 31):
        o = oracle.solve_kalman_logp(b["A"][i], b["B"][i], b["C"][i], b["D"][i], np.diag(b["sigma"][i] ** 2), om["Z"], y,
                                     H=np.diag(om["Hdiag"]), d=d, tol=1e-8, max_iter=1000)
        assert_allclose(run(kalman_block=1)["logp"][i], o["logp"], rtol=LOGP_RTOL)
    # complete data (the mask never changes): every steady draw hands over
    r0 = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], om["y"], Hdiag=om["Hdiag"], tol=1e-8,
                                           max_iter=1000)
    r1 = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], om["y"], Hdiag=om["Hdiag"], tol=1e-8,
                                           max_iter=1000, options={"kalman_block": 1})
    assert_allclose(r1["logp"], r0["logp"], rtol=1e-11)
    assert np.any(r1["logp"] != r0["logp"])  # (the blocked tail rounds differently: it did run)


def test_tail_hand_off_of_a_draw_smaller_than_its_tile():
    """A draw with far fewer state variables than the hint (9 of 18: reduced dimension m = 9 inside the 24-wide tile) is handed
    over like the others; the narrowest tail instance launched must take it (regression: instance selection by m)."""
    nb = 8
    b = wl.sw_shaped_batch(nb)
    om = wl.sw_shaped_observation_model()
    T = np.stack([oracle.cycle_reduction_core(b["A"][i], b["B"][i], b["C"][i], 1000, 1e-12)[0] for i in range(nb)])
    R = np.stack([oracle.compute_selection_matrix(b["B"][i], b["C"][i], b["D"][i], T[i]) for i in range(nb)])
    T[3][:, 9:18] = 0.0   # states 9..17 of draw 3 stop feeding back: 9 state columns (the observed variables 0..6 are among them)
    T[5][:, 2:18] = 0.0   # and a draw with 2 states + 5 observed non-states
    q = b["sigma"] ** 2
    base, st0 = batched.kalman_logp_batched(T, R, q, om["Z"], om["y"], Hdiag=om["Hdiag"], q_mode="diag_batched", n_state_hint=18)
    tail, st1 = batched.kalman_logp_batched(T, R, q, om["Z"], om["y"], Hdiag=om["Hdiag"], q_mode="diag_batched", n_state_hint=18,
                                            options={"kalman_block": 1})
    assert np.all(st0 == 0) and np.all(st1 == 0)
    assert_allclose(tail, base, rtol=1e-11)
    for i in (3, 5):
        ref = oracle.kalman_filter_logp(om["y"], T[i], R[i], np.diag(q[i]), om["Z"], H=np.diag(om["Hdiag"]))
        assert_allclose(tail[i], ref, rtol=LOGP_RTOL)
