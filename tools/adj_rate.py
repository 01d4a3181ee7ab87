"""GPU box: the two passes of the policy-adjoint launch on SW-shaped draws (rocprofv3 shows their durations)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from geconpy_amd import batched, workloads as wl, _lib
nb = 1024
b = wl.sw_shaped_batch(64)
rep = nb // 64
B, C, T = (np.tile(b[x], (rep, 1, 1)) for x in ("B", "C", "T_star"))
rng = np.random.default_rng(0)
Tb = rng.standard_normal(T.shape) * (T != 0).any(axis=1)[:, None, :]
lib = _lib.load()
for mode in (0, 2, 0):
    lib.dsge_debug_adjoint_refine(mode)
    Ab, Bb, Cb, st = batched.policy_adjoints_batched(B, C, T, Tb)
    print("mode", mode, "status nonzero", int((st != 0).sum()), "|A_bar|", float(np.abs(Ab).max()))
lib.dsge_debug_adjoint_refine(0)
# growth of the powers of G on the host, for the same draws
g = []
for i in range(64):
    M = b["B"][i] + b["C"][i] @ b["T_star"][i]
    G = -np.linalg.solve(M.T, b["C"][i].T)
    gm = 0.0
    for _ in range(12):
        gm = max(gm, np.abs(G).max()); G = G @ G
    g.append(gm)
print("host growth: median", np.median(g), "max", np.max(g), "above 100:", int((np.array(g) > 100).sum()))
