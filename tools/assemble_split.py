"""GPU box: time the two halves of assemble_kernel (selection+resid vs RQR'+Lyapunov) on the SW-shaped workload."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import _lib, workloads as wl
from geconpy_amd.engine import LogpEngine
nb = 4096
b = wl.sw_shaped_batch(64); rep = nb // 64
eng = LogpEngine(0); lib = _lib.load()
A, B, C, D = (eng.to_device(np.tile(b[x], (rep, 1, 1))) for x in "ABCD")
q = eng.to_device(np.tile(b["sigma"] ** 2, (rep, 1)))
n, k = 40, 7
T = torch.empty_like(A); R = torch.empty_like(D); P0 = torch.empty_like(A); RQR = torch.empty_like(A)
st = torch.zeros(nb, dtype=torch.int32, device=eng.device); it = torch.zeros_like(st); resid = torch.empty(nb, dtype=torch.float64, device=eng.device)
s = eng._stream()
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
cr = lambda: _lib.check(lib.dsge_cycle_reduction_batched(A.data_ptr(), B.data_ptr(), C.data_ptr(), nb, n, 1000, 1e-8, T.data_ptr(), st.data_ptr(), it.data_ptr(), s))
sel = lambda: _lib.check(lib.dsge_selection_batched(A.data_ptr(), B.data_ptr(), C.data_ptr(), D.data_ptr(), T.data_ptr(), nb, n, k, R.data_ptr(), resid.data_ptr(), s))
lyap = lambda: _lib.check(lib.dsge_lyapunov_batched(T.data_ptr(), R.data_ptr(), q.data_ptr(), 1, nb, n, k, P0.data_ptr(), RQR.data_ptr(), st.data_ptr(), s))
print("cycle reduction ms", timeit(cr)); print("selection+resid ms", timeit(sel)); print("RQR+lyapunov ms", timeit(lyap))
