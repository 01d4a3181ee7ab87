"""GPU box: PCIe-inclusive rate of the fused evaluation through the *_host twins (numpy in, numpy out)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from geconpy_amd import batched, workloads as wl
nb = 4096
b = wl.sw_shaped_batch(nb); om = wl.sw_shaped_observation_model()
args = (b["A"], b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], om["y"])
kw = dict(Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000, n_state_hint=18, z_selector_hint=1)
for _ in range(2): batched.solve_kalman_logp_batched(*args, **kw)
t0 = time.perf_counter(); n = 5
for _ in range(n): out = batched.solve_kalman_logp_batched(*args, **kw)
dt = (time.perf_counter() - t0) / n
print(f"host-pointer path: {dt*1e3:.2f} ms per {nb} draws = {nb/dt:.0f} evals/s (includes H2D of {sum(x.nbytes for x in args[:5])/1e6:.0f} MB, pageable memory)")
