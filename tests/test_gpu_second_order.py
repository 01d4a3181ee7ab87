"""SURVEY 8 (f4) / BASELINE configs[4] on the device: second-order coefficients (generalised Sylvester by doubling) and the
pruned-state-space quasi-likelihood (207-dimensional filter on the FP64 matrix core) against oracle/second_order.py.
*** parity unpinned against the reference: gEconpy has no second-order solver (perturbation.py:97-98 raises). ***"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import oracle
from geconpy_amd import batched
from geconpy_amd import workloads as wl
from oracle import second_order as so

pytestmark = pytest.mark.gpu
LOGP_RTOL = 1e-8


def _small_batch(n, ns, nl, k, nb, seed, nnz_per_eq=6):
    sysm = [wl.sw_shaped_system(seed + i, n=n, n_state=ns, n_lead=nl, k=k) for i in range(nb)]
    A, B, C, D = (np.stack([s_[j] for s_ in sysm]) for j in range(4))
    idx = wl.second_order_hessian_pattern(A[0], C[0], k, nnz_per_eq=nnz_per_eq, seed=seed)
    val = np.random.default_rng(seed + 99).standard_normal((nb, len(idx)))
    return A, B, C, D, idx, val


@pytest.mark.parametrize("n,ns,nl,k,obs,T_len", [(6, 3, 2, 2, (0, 4, 5), 40), (8, 2, 3, 1, (0, 1), 40), (12, 5, 4, 3, (0, 1, 2, 7), 40),
                                                 (20, 9, 6, 4, (0, 1, 2, 3), 40), (26, 12, 8, 5, (0, 1, 20), 150),
                                                 (32, 15, 9, 6, (0, 1, 2, 3, 4), 60)])
def test_second_order_small_models(n, ns, nl, k, obs, T_len):
    """Coefficients g_yy, g_yu, g_uu, g_ss and the pruned likelihood on random systems (some with observed non-states,
    i.e. u > s) against the reduced oracle; pruned states of 16, 9, 27, 63, 104 and 150 dimensions, i.e. the 2-, 4-, 7- and
    10-tile kernel instances (the 13-tile one runs the SW-shaped test); missing observations; the longer samples reach the
    steady-state switch."""
    nb = 5 if n <= 20 else 3
    A, B, C, D, idx, val = _small_batch(n, ns, nl, k, nb, 3100 + n)
    rng = np.random.default_rng(n)
    q = rng.uniform(0.5e-4, 4e-4, (nb, k))
    p = len(obs)
    Z = np.zeros((p, n))
    Z[np.arange(p), list(obs)] = 1.0
    y = rng.normal(0, 0.02, (T_len, p))
    y[7, 0] = np.nan
    y[20] = np.nan
    H = np.full(p, 1e-5)
    d = rng.normal(0, 0.01, p)
    out = batched.second_order_logp_batched(A, B, C, D, idx, val, q, Z, y, d=d, Hdiag=H, tol=1e-12, return_solution=True)
    assert (out["status"] == 0).all(), out["status"]
    for i in range(nb):
        r = so.solve_second_order_logp(A[i], B[i], C[i], D[i], idx, val[i], np.diag(q[i]), Z, y, H=np.diag(H), d=d, tol=1e-12)
        S = r["sol"]["S"]
        assert np.array_equal(S, out["S"])
        for key in ("g_yy", "g_yu", "g_uu", "g_ss"):
            ref = r["sol"][key]
            assert_allclose(out[key][i], ref, atol=1e-9 * max(1.0, np.abs(ref).max()), err_msg=f"{key} draw {i}")
        assert abs(out["logp"][i] - r["logp"]) <= LOGP_RTOL * abs(r["logp"]), (i, out["logp"][i], r["logp"])


def test_second_order_brock_mirman_closed_form():
    """The closed-form Brock-Mirman policy (tests/test_oracle_second_order.py): the device's second derivatives against the
    exact ones."""
    import os, sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_oracle_second_order import _brock_mirman

    A, B, C, D, H, T, R, G2 = _brock_mirman()
    n, k, m = 3, 1, 10
    Hd = H.reshape(n, m, m)
    idx = np.array([(i, a, b) for i in range(n) for a in range(m) for b in range(a, m) if Hd[i, a, b] != 0], dtype=np.int32)
    val = np.array([Hd[i, a, b] for i, a, b in idx])[None]
    Z = np.zeros((2, n))
    Z[0, 0] = Z[1, 1] = 1.0
    y = np.random.default_rng(0).normal(0, 0.02, (30, 2))
    out = batched.second_order_logp_batched(A[None], B[None], C[None], D[None], idx, val, np.array([0.05 ** 2]), Z, y,
                                            Hdiag=np.full(2, 1e-6), tol=1e-13, return_solution=True)
    assert out["status"][0] == 0
    S = out["S"]
    assert_allclose(out["T"][0], T, atol=1e-10)
    assert_allclose(out["g_yy"][0], G2[:, S][:, :, S], atol=1e-8)
    assert_allclose(out["g_yu"][0][:, :, 0], G2[:, S, 3], atol=1e-8)
    assert_allclose(out["g_uu"][0][:, 0, 0], G2[:, 3, 3], atol=1e-8)
    assert_allclose(out["g_ss"][0], 0.0, atol=1e-10)
    sol = so.second_order_solution_reduced(B, C, out["T"][0], out["R"][0], idx, val[0], np.array([[0.05 ** 2]]), S=S)
    ref = so.pruned_kalman_logp(out["T"][0], out["R"][0], sol, np.array([[0.05 ** 2]]), Z, y, H=np.diag(np.full(2, 1e-6)))
    assert abs(out["logp"][0] - ref) <= LOGP_RTOL * abs(ref)


def test_second_order_sw_shaped_draws():
    """BASELINE configs[4]: SW-shaped draws (n = 40, 18 states, 7 shocks, 7 observables, T = 200): the 207-dimensional pruned
    filter against the oracle, 1e-8 relative; a failed first-order solve gives -inf; repeated calls are bit-identical."""
    nb = 6
    b = wl.sw_second_order_batch(nb)
    om = wl.sw_shaped_observation_model()
    A = b["A"].copy()
    A[4, 0, 0] = np.nan
    q = b["sigma"] ** 2
    out = batched.second_order_logp_batched(A, b["B"], b["C"], b["D"], b["hess_idx"], b["hess_val"], q, om["Z"], om["y"],
                                            Hdiag=om["Hdiag"], tol=1e-8)
    out2 = batched.second_order_logp_batched(A, b["B"], b["C"], b["D"], b["hess_idx"], b["hess_val"], q, om["Z"], om["y"],
                                             Hdiag=om["Hdiag"], tol=1e-8)
    assert np.array_equal(out["logp"], out2["logp"]) and np.array_equal(out["status"], out2["status"])
    assert out["status"][4] != 0 and out["logp"][4] == -np.inf
    for i in (0, 1, 2, 3, 5):
        r = so.solve_second_order_logp(b["A"][i], b["B"][i], b["C"][i], b["D"][i], b["hess_idx"], b["hess_val"][i],
                                       np.diag(q[i]), om["Z"], om["y"], H=np.diag(om["Hdiag"]), tol=1e-8)
        assert out["status"][i] == 0
        assert abs(out["logp"][i] - r["logp"]) <= LOGP_RTOL * abs(r["logp"]), (i, out["logp"][i], r["logp"])


def test_second_order_sw_shaped_64_distinct_draws_against_fixture():
    """the full configs[4] shape (n = 40, 18 states, 7 shocks, 7 observables, T = 200, 480 Hessian entries) on 64
    DISTINCT draws against oracle/second_order.py through a committed fixture (tests/golden/second_order_sw64.npz, written by
    tests/golden/make_second_order_golden.py: 3 s of CPU per draw).  logp to 1e-8 relative, g_ss to 1e-9; the same 64 draws inside
    a 1024-draw batch (tiled; the dispatch order and the chunking are active there) come back bit-identical.
    *** parity unpinned against the reference by construction: gEconpy raises for order != 1 (perturbation.py:97-98). ***"""
    import os

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "second_order_sw64.npz"))
    nb = int(g["n_draws"])
    b = wl.sw_second_order_batch(nb)
    om = wl.sw_shaped_observation_model()
    q = b["sigma"] ** 2
    out = batched.second_order_logp_batched(b["A"], b["B"], b["C"], b["D"], b["hess_idx"], b["hess_val"], q, om["Z"], om["y"],
                                            Hdiag=om["Hdiag"], tol=1e-8, return_solution=True)
    assert (out["status"] == 0).all()
    rel = np.abs(out["logp"] - g["logp"]) / np.abs(g["logp"])
    assert rel.max() <= LOGP_RTOL, (int(rel.argmax()), rel.max())
    assert_allclose(out["g_ss"], g["g_ss"], atol=1e-9 * max(1.0, np.abs(g["g_ss"]).max()))
    rep = 16
    t3 = lambda x: np.tile(x, (rep, 1, 1))  # noqa: E731
    big = batched.second_order_logp_batched(t3(b["A"]), t3(b["B"]), t3(b["C"]), t3(b["D"]), b["hess_idx"], np.tile(b["hess_val"], (rep, 1)),
                                            np.tile(q, (rep, 1)), om["Z"], om["y"], Hdiag=om["Hdiag"], tol=1e-8)
    assert (big["status"] == 0).all()
    lp = big["logp"].reshape(rep, nb)
    assert np.array_equal(lp, np.broadcast_to(out["logp"], lp.shape))


def test_second_order_structure_violations_are_flagged():
    """A Hessian entry list that is not sorted by equation, and a design matrix that observes a variable outside the declared
    retained set: DSGE_ST_SECOND_ORDER_UNSUPPORTED (128) and logp = -inf for every draw, never a wrong number."""
    A, B, C, D, idx, val = _small_batch(8, 3, 2, 2, 3, 77)
    Z = np.zeros((1, 8))
    Z[0, 0] = 1.0
    y = np.random.default_rng(1).normal(0, 0.02, (10, 1))
    q = np.full(2, 1e-4)
    good = batched.second_order_logp_batched(A, B, C, D, idx, val, q, Z, y, Hdiag=np.full(1, 1e-5), tol=1e-12)
    assert (good["status"] == 0).all() and np.isfinite(good["logp"]).all()
    bad_idx = idx[::-1].copy()
    out = batched.second_order_logp_batched(A, B, C, D, bad_idx, val[:, ::-1].copy(), q, Z, y, Hdiag=np.full(1, 1e-5), tol=1e-12)
    assert ((out["status"] & 128) != 0).all() and (out["logp"] == -np.inf).all()
    S, Lc, U = batched.second_order_structure(A, C, Z)
    Z2 = Z.copy()
    Z2[0, 7] = 1.0  # observes a non-state that U does not list
    out = batched.second_order_logp_batched(A, B, C, D, idx, val, q, Z2, y, Hdiag=np.full(1, 1e-5), tol=1e-12, structure=(S, Lc, U))
    assert ((out["status"] & 128) != 0).all() and (out["logp"] == -np.inf).all()


def test_second_order_large_batch_is_chunked_and_ordered():
    """2560 draws (40 distinct, tiled; T = 60): more than one workspace chunk (the launcher keeps the per-draw workspace of the
    three kernels below 6 GiB: 2.7 MB per draw at the SW size), the slow-draws-first dispatch order of the filter launch
    (active from 512 draws) and the per-chunk offsets of every batched argument -- every copy of a draw must come back
    bit-identical to its first occurrence, in draw order, and equal to a small-batch evaluation (no dispatch order)."""
    nd, rep = 40, 64
    b = wl.sw_second_order_batch(nd)
    om = wl.sw_shaped_observation_model()
    y = om["y"][:60]
    q = b["sigma"] ** 2
    small = batched.second_order_logp_batched(b["A"], b["B"], b["C"], b["D"], b["hess_idx"], b["hess_val"], q, om["Z"], y,
                                              Hdiag=om["Hdiag"], tol=1e-8)
    assert (small["status"] == 0).all()
    tile3 = lambda x: np.tile(x, (rep, 1, 1))  # noqa: E731
    big = batched.second_order_logp_batched(tile3(b["A"]), tile3(b["B"]), tile3(b["C"]), tile3(b["D"]), b["hess_idx"],
                                            np.tile(b["hess_val"], (rep, 1)), np.tile(q, (rep, 1)), om["Z"], y, Hdiag=om["Hdiag"],
                                            tol=1e-8)
    assert (big["status"] == 0).all()
    lp = big["logp"].reshape(rep, nd)
    assert np.array_equal(lp, np.broadcast_to(lp[0], lp.shape))
    assert np.array_equal(lp[0], small["logp"])


def test_second_order_on_top_of_gensys():
    """solver="gensys": the first-order T, R come from the ordered QZ instead of cycle reduction; coefficients and the pruned
    likelihood agree with the cycle-reduction route to the accuracy of T (1e-9), and an explosive draw (no stable solution:
    eu != [1, 1]) is flagged, not evaluated."""
    nb = 6
    b = wl.sw_second_order_batch(nb)
    om = wl.sw_shaped_observation_model()
    q = b["sigma"] ** 2
    y = om["y"][:60]
    kw = dict(Hdiag=om["Hdiag"], tol=1e-10, return_solution=True)
    cr = batched.second_order_logp_batched(b["A"], b["B"], b["C"], b["D"], b["hess_idx"], b["hess_val"], q, om["Z"], y,
                                           solver="cycle_reduction", **kw)
    gs = batched.second_order_logp_batched(b["A"], b["B"], b["C"], b["D"], b["hess_idx"], b["hess_val"], q, om["Z"], y,
                                           solver="gensys", **kw)
    assert (cr["status"] == 0).all() and (gs["status"] == 0).all()
    assert np.abs(gs["logp"] - cr["logp"]).max() <= 1e-8 * np.abs(cr["logp"]).max()
    for key in ("g_yy", "g_yu", "g_uu", "g_ss"):
        assert np.abs(gs[key] - cr[key]).max() <= 1e-8 * max(1.0, np.abs(cr[key]).max()), key
    C = b["C"].copy()
    C[2] *= 25.0  # far too much weight on the expectations of draw 2
    bad = batched.second_order_logp_batched(b["A"], b["B"], C, b["D"], b["hess_idx"], b["hess_val"], q, om["Z"], y, solver="gensys",
                                            Hdiag=om["Hdiag"], tol=1e-10)
    assert bad["status"][2] != 0 and bad["logp"][2] == -np.inf
    ok = np.arange(nb) != 2
    assert (bad["status"][ok] == 0).all() and np.array_equal(bad["logp"][ok], gs["logp"][ok])
