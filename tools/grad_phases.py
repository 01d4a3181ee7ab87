"""GPU box: per-phase shader cycles of kalman_grad_kernel for draw 0 (debug buffer of dsge_debug_kalman_phases)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import _lib, workloads as wl
from geconpy_amd.engine import LogpEngine
if os.environ.get("DSGE_TEST_LIB"):  # (A/B of a differently built library)
    _lib.LIB_PATH = os.path.abspath(os.environ["DSGE_TEST_LIB"])
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b = wl.sw_shaped_batch(min(nb, 64)); om = wl.sw_shaped_observation_model(); rep = (nb + 63) // 64
eng = LogpEngine(0); lib = _lib.load()
A, B, C, D = (eng.to_device(np.tile(b[x], (rep, 1, 1))[:nb]) for x in "ABCD")
q = eng.to_device(np.tile(b["sigma"] ** 2, (rep, 1))[:nb]); Z = eng.to_device(om["Z"]); y = eng.to_device(om["y"]); H = eng.to_device(om["Hdiag"])
_lib.check(lib.dsge_debug_kalman_phases(1, None))
out = None
for it in range(2):
    out = eng.solve_kalman_logp_grad(A, B, C, D, q, Z, y, Hdiag=H, tol=1e-8, max_iter=1000, n_filter_hint=18, out=out)
torch.cuda.synchronize()
cyc = (ctypes.c_longlong * 16)()  # (the hook returns 16 values since ABI 8)
_lib.check(lib.dsge_debug_kalman_phases(0, ctypes.addressof(cyc)))
c = np.array(list(cyc), dtype=np.int64)
nf, ns = int(c[6]), int(c[7])
print(f"draw 0: {nf} full steps, {ns} steady steps")
print({"setup+P0": int(c[0]), "forward full (per step)": int(c[1] // max(nf, 1)), "forward steady (per step)": int(c[2] // max(ns, 1)),
       "reverse full (per step)": int(c[3] // max(nf, 1)), "reverse steady (per step)": int(c[4] // max(ns, 1)), "tail (dlyap adjoint + scatter)": int(c[5])})
print("totals", {"forward": int(c[1] + c[2]), "reverse": int(c[3] + c[4]), "kernel": int(c[:6].sum())})
q = c[8:16]
print("inside FULL steps (cycles per full step):", {"fwd update_cov+record": int(q[0] // max(nf, 1)), "fwd mean": int(q[1] // max(nf, 1)), "fwd predict": int(q[2] // max(nf, 1)),
      "rev record load": int(q[3] // max(nf, 1)), "rev products": int(q[5] // max(nf, 1)), "rev panels": int(q[6] // max(nf, 1)), "rev Pbar sym + end": int(q[7] // max(nf, 1))})
