"""GPU box: A/B of dsge_options.kalman_mfma (0 = VALU products, 2 = 4 x 4 x 4 FP64 matrix instruction in the NT kernel) on the headline
step: per-phase cycles of a lone draw (tools/kalman_phases.py's hook), stage times with and without the steady-state switch, logp
difference between the two."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import _lib, workloads as wl
from geconpy_amd.engine import LogpEngine
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b = wl.sw_shaped_batch(nb); om = wl.sw_shaped_observation_model()
eng = LogpEngine(0)
dev = [eng.to_device(b[x]) for x in "ABCD"]
q = eng.to_device(b["sigma"] ** 2); Z = eng.to_device(om["Z"]); y = eng.to_device(om["y"]); H = eng.to_device(om["Hdiag"])
ns, zs = eng.structure_hints(dev[0], Z)
hs = eng.static_hint(dev[0], dev[2])
lib = _lib.load()
names = ["F+GJinv", "gain K,V,a+", "P+ downdate+Pc", "a pred + GEMM1", "GEMM2+sym+PZt"]
res = {}
for mf in (0, 2):
    for stol in (1e-14, 0.0):
        opts = {"kalman_mfma": mf, "kalman_steady_tol": stol, "n_static_hint": hs}
        lp = torch.empty(nb, dtype=torch.float64, device="cuda"); st = torch.empty(nb, dtype=torch.int32, device="cuda")
        f = lambda: eng.solve_kalman_logp(*dev, q, Z, y, Hdiag=H, q_mode=1, tol=1e-8, max_iter=1000, logp=lp, status=st,
                                          n_state_hint=ns, z_selector_hint=zs, options=opts)
        for _ in range(5): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40): f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
        res[(mf, stol)] = lp.cpu().numpy().copy()
        print(f"kalman_mfma {mf} steady_tol {stol:g}: {dt*1e3:.4f} ms per step, {nb/dt/1e6:.3f} M evals/s, failed {int((st != 0).sum())}", flush=True)
    # lone-draw phases
    _lib.check(lib.dsge_debug_kalman_phases(1, None))
    opts = {"kalman_mfma": mf, "n_static_hint": hs}
    for _ in range(2):
        eng.solve_kalman_logp(*dev, q, Z, y, Hdiag=H, q_mode=1, tol=1e-8, max_iter=1000, n_state_hint=ns, z_selector_hint=zs, options=opts)
    torch.cuda.synchronize()
    cyc = (ctypes.c_longlong * 16)()
    _lib.check(lib.dsge_debug_kalman_phases(0, ctypes.addressof(cyc)))
    c = np.array(list(cyc))[:8]; n_ss = int(c[6]); n_full = 200 - n_ss
    print(f"kalman_mfma {mf}: draw 0 in the full batch: {n_full} full steps; per FULL step:", {n: int(v / max(n_full, 1)) for n, v in zip(names, c[:5])},
          "total", int(c[:5].sum() / max(n_full, 1)), "; per STEADY step:", int(c[5] / max(n_ss, 1)), flush=True)
for stol in (1e-14, 0.0):
    a, c = res[(0, stol)], res[(2, stol)]
    print(f"steady_tol {stol:g}: max rel logp difference mfma vs valu {np.max(np.abs(a - c) / np.abs(a)):.3e}")
