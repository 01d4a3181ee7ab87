"""GPU box: phase cycles of cr_big_kernel (dsge_debug_big_phases) for one workgroup.  python tools/big_phases.py [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from geconpy_amd import _lib, batched
from geconpy_amd import workloads as wl

SHAPES = {72: dict(n_state=30, n_lead=20, k=8), 80: dict(n_state=36, n_lead=24, k=10), 96: dict(n_state=44, n_lead=30, k=12)}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
sh = SHAPES[n]
sysm = [wl.sw_shaped_system(7000 + i, n=n, n_state=sh["n_state"], n_lead=sh["n_lead"], k=sh["k"]) for i in range(256)]
A, B, C = (np.stack([s[j] for s in sysm]) for j in range(3))
lib = _lib.load()
_lib.check(lib.dsge_debug_big_phases(1, None))
batched.cycle_reduction_batched(A, B, C, max_iter=1000, tol=1e-8)
out = (np.zeros(16, dtype=np.int64))
_lib.check(lib.dsge_debug_big_phases(0, out.ctypes.data))
names = ["block loads", "eliminations", "scatters", "products+norms"]
it = int(out[4])
print(f"n = {n}: {it} iterations, total {int(out[5])} cycles")
for i, nm in enumerate(names):
    print(f"  {nm:16s} {int(out[i]):9d} cycles = {int(out[i]) // max(it, 1):7d} per iteration")
piv = it * n
for i, nm in enumerate(["panel published", "barrier 1", "panel eliminated (wavefront 0)", "barrier 2", "pivot rows + barrier + update"]):
    print(f"  elimination / {nm:32s} {int(out[8 + i]) // max(piv, 1):6d} cycles per pivot step")
