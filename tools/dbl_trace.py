"""GPU box: a few evaluations with solver = gensys under each value of dsge_options.gensys_doubling, for rocprofv3 --kernel-trace
(which launches run, and for how long: tools/trace_summary.py reads the trace)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from geconpy_amd import workloads as wl
from geconpy_amd.engine import LogpEngine
from geconpy_amd.batched import lead_hint
nb = 4096
b = wl.sw_shaped_batch(nb)
om = wl.sw_shaped_observation_model()
eng = LogpEngine(0)
dev = [eng.to_device(b[x]) for x in "ABCD"]
dq = eng.to_device(b["sigma"] ** 2)
dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
ns, zs = eng.structure_hints(dev[0], dZ)
hs = eng.static_hint(dev[0], dev[2])
nl = lead_hint(b["C"], 1e-8)
lp = torch.empty(nb, dtype=torch.float64, device="cuda"); st = torch.empty(nb, dtype=torch.int32, device="cuda")
for _ in range(8):
    eng.solve_kalman_logp(*dev, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, logp=lp, status=st, solver="gensys", n_state_hint=ns,
                          z_selector_hint=zs, n_lead_hint=nl, options={"n_static_hint": hs, "gensys_doubling": 1})
torch.cuda.synchronize()
