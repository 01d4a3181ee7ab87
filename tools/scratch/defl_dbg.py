import sys; sys.path.insert(0, '.')
import numpy as np, torch
from geconpy_amd import _lib, workloads as wl
from geconpy_amd.engine import LogpEngine
from tests.test_gpu_parity import _fused_policy
lib = _lib.load()
nb = 64
b = wl.sw_shaped_batch(nb, first_draw=9100); om = wl.sw_shaped_observation_model()
eng = LogpEngine(torch.device("cuda", 0))
dev = {x: eng.to_device(b[x]) for x in "ABCD"}
dq = eng.to_device(b["sigma"] ** 2)
dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"][:40]), eng.to_device(om["Hdiag"])
hints = eng.structure_hints(dev["A"], dZ)
lib.dsge_set_cr_deflation(0); lp0, st0, T0, R0 = _fused_policy(eng, dev, dq, dZ, dy, dH, hints)
lib.dsge_set_cr_deflation(1); lp1, st1, T1, R1 = _fused_policy(eng, dev, dq, dZ, dy, dH, hints)
A = b["A"][0]; C = b["C"][0]
static = np.where(~(A != 0).any(0) & ~(C != 0).any(0))[0]
print("static", static)
print("dT by row", np.abs(T1 - T0).max(axis=(0, 2)))
print("dR by row", np.abs(R1 - R0).max(axis=(0, 2)))
print("dlogp", np.abs(lp1 - lp0).max(), st1[:5])
import oracle
Rref = oracle.compute_selection_matrix(b["B"][0], b["C"][0], b["D"][0], T0[0])
print("R0 vs oracle", np.abs(R0[0]-Rref).max(), "R1 vs oracle", np.abs(R1[0]-Rref).max())
