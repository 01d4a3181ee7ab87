"""Per-phase shader cycles of kalman_nt_kernel (draw 0) with and without the blocked steady recursion."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import _lib, workloads as wl
from geconpy_amd.engine import LogpEngine
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
b = wl.sw_shaped_batch(min(nb, 64)); om = wl.sw_shaped_observation_model()
rep = (nb + 63) // 64
eng = LogpEngine(0)
dev = [eng.to_device(np.tile(b[x], (rep, 1, 1))[:nb]) for x in "ABCD"]
q = eng.to_device(np.tile(b["sigma"] ** 2, (rep, 1))[:nb]); Z = eng.to_device(om["Z"]); y = eng.to_device(om["y"]); H = eng.to_device(om["Hdiag"])
lib = _lib.load()
for blk in (0, 1):
    _lib.check(lib.dsge_debug_kalman_phases(1, None))
    for _ in range(2):
        eng.solve_kalman_logp(*dev, q, Z, y, Hdiag=H, q_mode=1, tol=1e-8, max_iter=1000, n_state_hint=18, z_selector_hint=1,
                              options={"kalman_steady_block": blk})
    torch.cuda.synchronize()
    cyc = (ctypes.c_longlong * 16)()
    _lib.check(lib.dsge_debug_kalman_phases(0, ctypes.addressof(cyc)))
    c = np.array(list(cyc))
    n_ss = int(c[6]); n_full = 200 - n_ss
    names = ["F+GJinv", "gain K,V,a+", "P+ downdate+Pc", "a pred + GEMM1", "GEMM2+sym+PZt"]
    print(f"steady_block {blk} ({nb} draws): draw 0: {n_full} full steps, {n_ss} steady; kernel total {int(c[7])} cycles")
    print("  per FULL step:", {n: int(v / max(n_full, 1)) for n, v in zip(names, c[:5])}, "total", int(c[:5].sum() / max(n_full, 1)))
    print("  steady segment(s) total", int(c[5]), "cycles = per STEADY step:", int(c[5] / max(n_ss, 1)))
