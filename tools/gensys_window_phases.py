"""GPU box: per-phase shader cycles of the window-path gensys kernels for draw 0 of an SW-shaped batch."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from geconpy_amd import _lib, batched, workloads as wl
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b = wl.sw_shaped_batch(nb)
lib = _lib.load()
A, B, C, D = (b[x] for x in "ABCD")
_lib.check(lib.dsge_debug_gensys_window_phases(1, None))
for _ in range(2):
    out = batched.gensys_batched(A, B, C, D, tol=1e-8)
cyc = (ctypes.c_longlong * 32)()
_lib.check(lib.dsge_debug_gensys_window_phases(0, ctypes.addressof(cyc)))
c = np.array(list(cyc), dtype=np.int64)
def split(names, base):
    return {nm: int(c[base + i + 1] - c[base + i]) for i, nm in enumerate(names)}
print("reduce ", {"zero + pencil": int(c[0] - c[14]), "deflation (reflectors)": int(c[7] - c[0]), "store": int(c[1] - c[7])})
print("hesstri", {"load": 0, "T22 triangular(hh) incl. load": int(c[2] - c[5]), "Hessenberg(givens)": int(c[3] - c[2]), "real double-shift sweeps": int(c[6] - c[3]), "store": int(c[4] - c[6])}, "total", int(c[4] - c[5]))
print("real double-shift stage:", int(c[27]), "sweep steps in", int(c[28]), "sweeps")
print("pair kernel (two draws per wavefront): steps of all pairs (sum over the calls since the debug buffer was reset)", int(c[29]), "slowest pair", int(c[30]), "stage-A cycles of pair 0", int(c[31]))
print("qz    ", split(["qz", "reorder", "store"], 8), "total", int(c[11] - c[8]))
print("qz sweep steps", int(c[12]) // 2, "sweeps", int(c[13]) // 2, "(per call)")
print("eu    ", split(["load", "svd", "eu+Bm+Phi"], 16), "total", int(c[19] - c[16]))
print("post  ", split(["load", "rhs", "backsub", "Wb+BB+RR", "E+R0 solve", "T write"], 20), "total", int(c[26] - c[20]))
