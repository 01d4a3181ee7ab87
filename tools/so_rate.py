"""GPU box: stage timings of the second-order path (BASELINE configs[4]) on SW-shaped draws."""
import sys, os, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import workloads as wl
from geconpy_amd.engine import LogpEngine
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nd = min(nb, 256)
b = wl.sw_second_order_batch(nd); om = wl.sw_shaped_observation_model(); rep = (nb + nd - 1) // nd
eng = LogpEngine(0)
A, B, C, D = (eng.to_device(np.tile(b[x], (rep, 1, 1))[:nb]) for x in "ABCD")
q = eng.to_device(np.tile(b["sigma"] ** 2, (rep, 1))[:nb])
hv = eng.to_device(np.tile(b["hess_val"], (rep, 1))[:nb])
hi = torch.as_tensor(b["hess_idx"], dtype=torch.int32, device=eng.device).contiguous()
Z = eng.to_device(om["Z"]); y = eng.to_device(om["y"]); H = eng.to_device(om["Hdiag"])
st = eng.second_order_structure(A, C, Z)
ms = (ctypes.c_float * 4)()
from geconpy_amd import _lib
cyc = (ctypes.c_longlong * 8)()
_lib.check(eng.lib.dsge_debug_second_order_phases(1, None))
at = torch.full((nb,), -1, dtype=torch.int32, device=eng.device)
lp_full = None
for tol in (0.0, 1e-16, 1e-15, 1e-14, 1e-13):  # (kalman_steady_tol; the second-order filter tests against 100 x the value)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lp, stt = eng.second_order_logp(A, B, C, D, hi, hv, q, Z, y, st, Hdiag=H, tol=1e-8, stage_ms=ms if it == 2 else None,
                                        options={"kalman_steady_tol": tol})
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"steady_tol {tol:g}: {nb} draws: {dt*1e3:.1f} ms = {nb/dt:.0f} evals/s; stages (ms): solver {ms[0]:.2f}, set-up {ms[1]:.2f}, "
          f"P0 {ms[2]:.2f}, filter {ms[3]:.2f}; failed {int((stt != 0).sum())}; logp[0] {float(lp[0]):.6f}")
    if tol == 0.0:
        lp_full = lp.clone()
    at.fill_(-1); eng.record_steady_steps(at)
    eng.second_order_logp(A, B, C, D, hi, hv, q, Z, y, st, Hdiag=H, tol=1e-8, options={"kalman_steady_tol": tol}); torch.cuda.synchronize()
    eng.record_steady_steps(None)
    ah = at.cpu().numpy(); nfull = np.where(ah < 0, 200, ah)
    print(f"   full steps: mean {nfull.mean():.1f}, median {np.median(nfull):.0f}, never steady {int((ah < 0).sum())}; max rel logp diff vs full recursion {float((torch.abs(lp - lp_full) / torch.abs(lp_full)).max()):.2e}")
    _lib.check(eng.lib.dsge_debug_second_order_phases(1, ctypes.addressof(cyc)))
    nf, ns = max(cyc[5], 1), max(cyc[6], 1)
    print(f"   draw 0: {cyc[5]} full steps x (update + mean {cyc[0]/nf:.0f} + Az K, Az V {cyc[1]/nf:.0f} + product 1 {cyc[2]/nf:.0f} + product 2 {cyc[3]/nf:.0f}) cycles, "
          f"{cyc[6]} steady steps x {cyc[4]/ns:.0f} cycles, total {cyc[7]} cycles")
