"""GPU box: print the per-phase shader cycles of gensys_kernel for draw 0 of an SW-shaped batch."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import _lib, workloads as wl
from geconpy_amd.batched import lead_hint
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
b = wl.sw_shaped_batch(nb)
lib = _lib.load()
dev = torch.device("cuda", 0)
A, B, C = (torch.as_tensor(b[x], device=dev) for x in "ABC")
T = torch.empty_like(A); eu = torch.empty((nb, 3), dtype=torch.int32, device=dev); st = torch.empty(nb, dtype=torch.int32, device=dev)
cyc = (ctypes.c_longlong * 6)()
for _ in range(2):
    _lib.check(lib.dsge_debug_gensys_phases(A.data_ptr(), B.data_ptr(), C.data_ptr(), nb, 40, 1e-8, lead_hint(b["C"]), T.data_ptr(), eu.data_ptr(), st.data_ptr(), ctypes.addressof(cyc)))
c = np.array(list(cyc), dtype=np.int64)
names = ["hess_tri", "qz", "reorder", "svd+eu", "phi+solve+T"]
print({n: int(c[i + 1] - c[i]) for i, n in enumerate(names)}, "total", int(c[5] - c[0]))
