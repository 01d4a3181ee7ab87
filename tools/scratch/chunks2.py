import sys, os
sys.path.insert(0, '.')
import numpy as np, torch
from geconpy_amd import _lib, workloads as wl
from geconpy_amd.engine import LogpEngine
nb = 4096
b = wl.sw_shaped_batch(nb); om = wl.sw_shaped_observation_model()
eng = LogpEngine(torch.device("cuda", 0)); lib = _lib.load()
dA, dB, dC, dD = (eng.to_device(b[x]) for x in "ABCD"); dq = eng.to_device(b["sigma"] ** 2)
dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
hints = eng.structure_hints(dA, dZ)
def run():
    return eng.solve_kalman_logp(dA, dB, dC, dD, dq, dZ, dy, Hdiag=dH, tol=1e-8, max_iter=1000, n_state_hint=hints[0], z_selector_hint=hints[1])
for ch in (0, 2, 3, 4, 6, 8):
    _lib.check(lib.dsge_set_pipeline_chunks(ch))
    out = run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): out = run()
    e1.record(); torch.cuda.synchronize()
    print(f"chunks {ch}: {e0.elapsed_time(e1)/10:.3f} ms/step")
_lib.check(lib.dsge_set_pipeline_chunks(0))
