"""GPU box: distribution of the number of full covariance steps (first steady-state step) over the 4096 SW-shaped bench draws."""
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import _lib, workloads as wl
from geconpy_amd.engine import LogpEngine
nb = 4096
b = wl.sw_shaped_batch(nb); om = wl.sw_shaped_observation_model()
eng = LogpEngine(torch.device("cuda", 0))
dev = {x: eng.to_device(b[x]) for x in "ABCD"}
dq = eng.to_device(b["sigma"] ** 2)
dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
hints = eng.structure_hints(dev["A"], dZ)
at = torch.full((nb,), -1, dtype=torch.int32, device=eng.device)
eng.record_steady_steps(at)
eng.solve_kalman_logp(dev["A"], dev["B"], dev["C"], dev["D"], dq, dZ, dy, Hdiag=dH, tol=1e-8, max_iter=1000, n_state_hint=hints[0], z_selector_hint=hints[1])
torch.cuda.synchronize(); eng.record_steady_steps(None)
a = at.cpu().numpy(); full = np.where(a < 0, 200, a)
s = np.sort(full)[::-1]
print("top 40 full-step counts:", s[:40])
print("quantiles 50/90/99/99.9:", np.percentile(full, [50, 90, 99, 99.9]))
for thr in (40, 60, 80, 100, 150): print("draws above", thr, ":", int((full > thr).sum()))
