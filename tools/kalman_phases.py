"""GPU box: per-phase shader cycles of kalman_sel_kernel (draw 0) on the SW-shaped workload."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import _lib, workloads as wl
from geconpy_amd.engine import LogpEngine
from _opts import set_option  # (tools/_opts.py: per-thread dsge_options)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b = wl.sw_shaped_batch(min(nb, 64)); om = wl.sw_shaped_observation_model()
rep = (nb + 63) // 64
eng = LogpEngine(0)
dev = [eng.to_device(np.tile(b[x], (rep, 1, 1))[:nb]) for x in "ABCD"]
q = eng.to_device(np.tile(b["sigma"] ** 2, (rep, 1))[:nb]); Z = eng.to_device(om["Z"]); y = eng.to_device(om["y"]); H = eng.to_device(om["Hdiag"])
lib = _lib.load()
_lib.check(lib.dsge_debug_kalman_phases(1, None))
for _ in range(2):
    eng.solve_kalman_logp(*dev, q, Z, y, Hdiag=H, q_mode=1, tol=1e-8, max_iter=1000, n_state_hint=18, z_selector_hint=1)
torch.cuda.synchronize()
cyc = (ctypes.c_longlong * 16)()  # (the hook returns 16 values since ABI 8)
_lib.check(lib.dsge_debug_kalman_phases(0, ctypes.addressof(cyc)))
names = ["F+GJinv", "gain K,V,a+", "P+ downdate+Pc", "a pred + GEMM1", "GEMM2+sym+PZt"]
c = np.array(list(cyc))[:8]; n_ss = int(c[6]); n_full = 200 - n_ss
print("draw 0:", n_full, "full steps,", n_ss, "steady steps; kernel total", int(c[7]), "cycles")
print("per FULL step:", {n: int(v / max(n_full, 1)) for n, v in zip(names, c[:5])}, "total", int(c[:5].sum() / max(n_full, 1)))
print("per STEADY step:", int(c[5] / max(n_ss, 1)))
at = torch.full((nb,), -7, dtype=torch.int32, device=eng.device)
eng.record_steady_steps(at)
eng.solve_kalman_logp(*dev, q, Z, y, Hdiag=H, q_mode=1, tol=1e-8, max_iter=1000, n_state_hint=18, z_selector_hint=1)
torch.cuda.synchronize(); eng.record_steady_steps(None)
a = at.cpu().numpy()[:64]
print("first steady step over 64 distinct draws: min", a.min(), "median", int(np.median(a)), "max", a.max(), "never", int((a < 0).sum()))
for tol in (0.0, 1e-14):
    set_option("kalman_steady_tol", tol)
    print("steady_tol", tol, eng.profile_kernels(*dev, q, Z, y, Hdiag=H, q_mode=1, tol=1e-8, max_iter=1000, n_state_hint=18, z_selector_hint=1))
