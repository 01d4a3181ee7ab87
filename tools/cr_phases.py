"""GPU box: per-phase shader cycles of cr_compact_kernel (draw 0) on the SW-shaped workload."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import _lib, workloads as wl
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
if N == 40:
    b = wl.sw_shaped_batch(min(nb, 64))
else:  # e.g. 30: the size the deflated SW-shaped system iterates on (18 states, 12 leads, no static variables)
    sysm = [wl.sw_shaped_system(100 + i, n=N, n_state=18, n_lead=12, k=7) for i in range(64)]
    b = {x: np.stack([s_[j] for s_ in sysm]) for j, x in enumerate("ABC")}
rep = (nb + 63) // 64
dev = torch.device("cuda", 0); lib = _lib.load()
A, B, C = (torch.as_tensor(np.tile(b[x], (rep, 1, 1))[:nb], device=dev) for x in "ABC")
T = torch.empty_like(A); st = torch.empty(nb, dtype=torch.int32, device=dev); it = torch.empty_like(st)
_lib.check(lib.dsge_debug_cr_phases(1, None))
for _ in range(2):
    _lib.check(lib.dsge_cycle_reduction_batched(A.data_ptr(), B.data_ptr(), C.data_ptr(), nb, N, 1000, 1e-8, T.data_ptr(), st.data_ptr(), it.data_ptr(), None))
torch.cuda.synchronize()
cyc = (ctypes.c_longlong * 8)()
_lib.check(lib.dsge_debug_cr_phases(0, ctypes.addressof(cyc)))
c = np.array(list(cyc)); names = ["GJ panels", "GJ trailing", "gather+stage", "products", "scatter+norms"]
print("iterations", int(c[7]), "total", int(c[6]), "final solve", int(c[5]))
print("per iteration:", {n: int(v / max(c[7], 1)) for n, v in zip(names, c[:5])}, "(GJ numbers include the final solve's)")
