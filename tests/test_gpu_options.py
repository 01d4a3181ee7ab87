"""Per-call options (include/dsge_hip.h: dsge_options) and host-thread safety of the C ABI ("re-entrant per
stream"): two host threads with different kernel variants get the results of their OWN settings, on the host twins
(which lease separate staging arenas) and on the device entry points on two streams."""
import threading

import numpy as np
import pytest

from geconpy_amd import _lib, batched
from geconpy_amd import workloads as wl

pytestmark = pytest.mark.gpu

VARIANT_A = {"kalman_steady_tol": 0.0, "cr_deflation": 0, "kalman_order": 0}   # step-for-step filter, full-size iteration
VARIANT_B = {"kalman_steady_tol": 1e-7, "cr_compact": 0, "cr_fused_selection": 0}  # early freeze, dense kernel, explicit R


def _inputs(nb, first=0):
    b = wl.sw_shaped_batch(nb, first_draw=first)
    om = wl.sw_shaped_observation_model()
    return b, om


def _run(b, om, options):
    return batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], om["y"],
                                             Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000, options=options)["logp"]


def test_two_host_threads_keep_their_own_options():
    b1, om = _inputs(192)
    b2, _ = _inputs(160, first=500)
    ref_a, ref_b = _run(b1, om, VARIANT_A), _run(b2, om, VARIANT_B)
    # the two variants are observably different computations (else the test proves nothing)
    assert not np.array_equal(_run(b1, om, VARIANT_B), ref_a)
    assert np.allclose(_run(b1, om, VARIANT_B), ref_a, rtol=1e-5)
    out = {"a": [], "b": []}
    errs = []

    def worker(key, b, opts, reps):
        try:
            for _ in range(reps):
                out[key].append(_run(b, om, opts))
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ta = threading.Thread(target=worker, args=("a", b1, VARIANT_A, 12))
    tb = threading.Thread(target=worker, args=("b", b2, VARIANT_B, 12))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert not errs, errs
    for r in out["a"]:
        assert np.array_equal(r, ref_a)  # bit-identical to the single-threaded run with the same options
    for r in out["b"]:
        assert np.array_equal(r, ref_b)
    # and the process-wide defaults were never touched
    assert _lib.make_options().kalman_steady_tol == 1e-14


def test_two_streams_two_threads_device_entry_points():
    import torch

    from geconpy_amd.engine import LogpEngine

    eng = LogpEngine(0)
    b, om = _inputs(512)
    dev = [eng.to_device(b[x]) for x in "ABCD"]
    dq = eng.to_device(b["sigma"] ** 2)
    dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
    ns, zs = eng.structure_hints(dev[0], dZ)
    nst = eng.static_hint(dev[0], dev[2])
    assert nst == 10
    kw = dict(Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, n_state_hint=ns, z_selector_hint=zs)
    opt_a = dict(VARIANT_A, n_static_hint=nst)
    opt_b = dict(kalman_steady_tol=1e-7, n_static_hint=nst)

    def once(opts):
        lp, st = eng.solve_kalman_logp(*dev, dq, dZ, dy, options=opts, **kw)
        torch.cuda.synchronize()
        assert (st.cpu().numpy() == 0).all()
        return lp.cpu().numpy().copy()

    ref_a, ref_b = once(opt_a), once(opt_b)
    assert not np.array_equal(ref_a, ref_b)
    res, errs = {}, []

    def worker(key, opts):
        try:
            torch.cuda.set_device(0)
            s = torch.cuda.Stream()
            outs = []
            with torch.cuda.stream(s):
                for _ in range(8):
                    lp, _st = eng.solve_kalman_logp(*dev, dq, dZ, dy, options=opts, **kw)
                    outs.append(lp.clone())
            s.synchronize()
            res[key] = [o.cpu().numpy() for o in outs]
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=worker, args=("a", opt_a)), threading.Thread(target=worker, args=("b", opt_b))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert all(np.array_equal(r, ref_a) for r in res["a"])
    assert all(np.array_equal(r, ref_b) for r in res["b"])


def test_static_hint_makes_the_fused_call_a_pure_enqueue():
    """With n_static_hint the launcher measures nothing (no read-back, no synchronisation): same bits as the measured
    path, a too-large hint is caught per draw on the device, and the call can be captured in a HIP graph."""
    import torch

    from geconpy_amd.engine import LogpEngine

    eng = LogpEngine(0)
    b, om = _inputs(256)
    dev = [eng.to_device(b[x]) for x in "ABCD"]
    dq = eng.to_device(b["sigma"] ** 2)
    dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
    ns, zs = eng.structure_hints(dev[0], dZ)
    kw = dict(Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, n_state_hint=ns, z_selector_hint=zs)
    # the number of static variables is measured once per model SIZE and reused: a test that ran 40-variable systems without
    # static variables before this one leaves a record under which the measured path does not deflate (same results to rounding,
    # not to the bit) -- start from a clean slate, as a caller who changes the model behind a fixed size does
    _lib.check(_lib.load().dsge_forget_measured_shapes())

    def run(opts):
        lp, st = eng.solve_kalman_logp(*dev, dq, dZ, dy, options=opts, **kw)
        torch.cuda.synchronize()
        return lp.cpu().numpy().copy(), st.cpu().numpy().copy()

    measured, st0 = run(None)
    hinted, st1 = run({"n_static_hint": 10})
    assert np.array_equal(measured, hinted) and not st0.any() and not st1.any()
    over, st2 = run({"n_static_hint": 13})  # more than the model has: every draw falls back to the full-size kernels
    assert not st2.any()
    full, _ = run({"cr_deflation": 0})
    assert np.array_equal(over, full)
    # stream capture: nothing in the hinted call may synchronise
    logp = torch.empty(256, dtype=torch.float64, device="cuda")
    status = torch.empty(256, dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        eng.solve_kalman_logp(*dev, dq, dZ, dy, options={"n_static_hint": 10}, logp=logp, status=status, **kw)  # warm: sizes the scratch
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    logp.zero_()
    with torch.cuda.graph(g, stream=s):
        eng.solve_kalman_logp(*dev, dq, dZ, dy, options={"n_static_hint": 10}, logp=logp, status=status, **kw)
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(logp.cpu().numpy(), hinted)


def test_nt_kernel_matches_round1_kernel():
    """kalman_nt_kernel (NT products, 16-byte LDS loads) vs kalman_sel_kernel: same algorithm, products summed in a different
    order -- logp agrees to rounding on SW-shaped draws incl. missing data, and both stay within 1e-9 of the oracle."""
    import oracle

    b, om = _inputs(96)
    y = om["y"].copy()
    y[5, 2] = np.nan
    y[17, :] = np.nan
    y[40, 0] = -9999.0
    for yy in (om["y"], y):
        kw = dict(Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000)
        a = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], yy, options={"kalman_nt_products": 1}, **kw)
        c = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], yy, options={"kalman_nt_products": 0}, **kw)
        assert not a["status"].any() and not c["status"].any()
        assert np.max(np.abs(a["logp"] - c["logp"]) / np.abs(c["logp"])) < 1e-11
        for i in (0, 50, 95):
            ref = oracle.solve_kalman_logp(b["A"][i], b["B"][i], b["C"][i], b["D"][i], np.diag(b["sigma"][i] ** 2), om["Z"], yy,
                                           H=np.diag(om["Hdiag"]), tol=1e-8, max_iter=1000)
            assert abs(a["logp"][i] - ref["logp"]) <= 1e-9 * abs(ref["logp"])


def test_two_host_threads_in_the_other_twins():
    """every host twin runs on the calling thread's own streams (twin_streams), so the device scratch of the entry
    points -- keyed by (device, stream) -- is never shared between two host threads.  Two threads hammer the gradient twin
    (the NUTS path), the gensys twin and the standalone filter twin with DIFFERENT batches: every result is bit-identical to
    the single-threaded one."""
    b1, om = _inputs(96)
    b2, _ = _inputs(80, first=700)

    def grad(b):
        g = batched.solve_kalman_logp_grad_batched(b["A"], b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], om["y"],
                                                   Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000)
        return np.concatenate([g["logp"], g["A_bar"].ravel(), g["q_bar"].ravel()])

    def gensys(b):
        g = batched.gensys_batched(b["A"], b["B"], b["C"], b["D"], tol=1e-8)
        return np.concatenate([g["T"].ravel(), g["R"].ravel(), g["eu"].ravel().astype(float)])

    def filt(b):
        g = batched.gensys_batched(b["A"][:32], b["B"][:32], b["C"][:32], b["D"][:32], tol=1e-8)
        lp, st = batched.kalman_logp_batched(g["T"], g["R"], b["sigma"][:32] ** 2, om["Z"], om["y"], Hdiag=om["Hdiag"])
        return lp

    for fn in (grad, gensys, filt):
        ref1, ref2 = fn(b1), fn(b2)
        out = {1: [], 2: []}
        errs = []

        def worker(key, b, reps=6):
            try:
                for _ in range(reps):
                    out[key].append(fn(b))
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        t1 = threading.Thread(target=worker, args=(1, b1))
        t2 = threading.Thread(target=worker, args=(2, b2))
        t1.start(); t2.start(); t1.join(); t2.join()
        assert not errs, (fn.__name__, errs)
        assert all(np.array_equal(r, ref1) for r in out[1]), fn.__name__
        assert all(np.array_equal(r, ref2) for r in out[2]), fn.__name__


def test_many_short_lived_threads_do_not_exhaust_the_arenas():
    """Thread-local twin streams release their scratch arenas when the thread exits: forty threads, one after the other (far
    more than the slots of a pool), keep getting correct, identical results."""
    b, om = _inputs(64)
    ref = _run(b, om, None)
    res = []

    def worker():
        res.append(_run(b, om, None))

    for _ in range(40):
        t = threading.Thread(target=worker)
        t.start()
        t.join()
    assert len(res) == 40 and all(np.array_equal(r, ref) for r in res)
