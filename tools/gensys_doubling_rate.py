"""gensys leg of the headline step with and without dsge_options.gensys_doubling; fraction of certified draws."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from geconpy_amd import _lib, workloads as wl
from geconpy_amd.engine import LogpEngine
from geconpy_amd.batched import lead_hint
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b = wl.sw_shaped_batch(nb)
om = wl.sw_shaped_observation_model()
eng = LogpEngine(0)
dev = [eng.to_device(b[x]) for x in "ABCD"]
dq = eng.to_device(b["sigma"] ** 2)
dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
ns, zs = eng.structure_hints(dev[0], dZ)
hs = eng.static_hint(dev[0], dev[2])
nl = lead_hint(b["C"], 1e-8)
res = {}
for dbl in (0, 1, 2, 3):
    lp = torch.empty(nb, dtype=torch.float64, device="cuda"); st = torch.empty(nb, dtype=torch.int32, device="cuda")
    opts = {"gensys_doubling": dbl, "n_static_hint": hs}
    f = lambda: eng.solve_kalman_logp(*dev, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, logp=lp, status=st, solver="gensys",
                                      n_state_hint=ns, z_selector_hint=zs, n_lead_hint=nl, options=opts)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    with _lib.options_scope(opts):
        ms = eng.profile_kernels(*dev, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, reps=5, n_state_hint=ns, z_selector_hint=zs,
                                 solver="gensys", n_lead_hint=nl)
    res[dbl] = lp.cpu().numpy().copy()
    print(f"gensys_doubling {dbl}: {dt * 1e3:.4f} ms per step = {nb / dt / 1e6:.3f} M evals/s; stages", {k: round(v, 4) for k, v in ms.items()},
          "failed", int((st != 0).sum()), flush=True)
rel = np.abs(res[1] - res[0]) / np.abs(res[0])
print("max / median relative difference of logp, doubling vs QZ:", rel.max(), np.median(rel), "argmax", int(rel.argmax()))

# batches in which a share of the draws is NOT regular (explosive state block: the certificate fails, the ordered QZ decides)
for share in (0.02, 0.5, 1.0):
    A2 = b["A"].copy()
    bad = np.random.default_rng(5).random(nb) < share
    A2[bad] *= 25.0
    dA2 = eng.to_device(A2)
    out = {}
    for dbl in (0, 1, 2, 3):
        lp = torch.empty(nb, dtype=torch.float64, device="cuda"); st = torch.empty(nb, dtype=torch.int32, device="cuda")
        opts = {"gensys_doubling": dbl, "n_static_hint": hs}
        f = lambda: eng.solve_kalman_logp(dA2, *dev[1:], dq, dZ, dy, Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, logp=lp, status=st,
                                          solver="gensys", n_state_hint=ns, z_selector_hint=zs, n_lead_hint=nl, options=opts)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            f()
        torch.cuda.synchronize()
        out[dbl] = ((time.perf_counter() - t0) / 10, st.cpu().numpy().copy(), lp.cpu().numpy().copy())
    same = all(np.array_equal(out[0][1], out[d][1]) for d in (1, 2))
    okm = out[0][1] == 0
    dl = max(float(np.max(np.abs(out[d][2][okm] - out[0][2][okm]) / np.abs(out[0][2][okm]))) if okm.any() else 0.0 for d in (1, 2))
    print(f"share of non-regular draws {share}: QZ {out[0][0] * 1e3:.3f} ms, spectral division (window fall-back) {out[1][0] * 1e3:.3f} ms, "
          f"(single-launch fall-back) {out[2][0] * 1e3:.3f} ms; failed {int((out[0][1] != 0).sum())}; status identical {same}; max rel logp diff {dl:.2e}", flush=True)
