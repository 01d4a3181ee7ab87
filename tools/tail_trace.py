import sys
import numpy as np, torch
sys.path.insert(0, ".")
from geconpy_amd import workloads as wl
from geconpy_amd.engine import LogpEngine
blk = int(sys.argv[1])
nb = 4096
b = wl.sw_shaped_batch(nb)
om = wl.sw_shaped_observation_model()
eng = LogpEngine(0)
dev = [eng.to_device(b[x]) for x in "ABCD"]
dq = eng.to_device(b["sigma"] ** 2)
dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
ns, zs = eng.structure_hints(dev[0], dZ)
hs = eng.static_hint(dev[0], dev[2])
lp = torch.empty(nb, dtype=torch.float64, device="cuda"); st = torch.empty(nb, dtype=torch.int32, device="cuda")
for _ in range(12):
    eng.solve_kalman_logp(*dev, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, logp=lp, status=st, n_state_hint=ns,
                          z_selector_hint=zs, options={"n_static_hint": hs, "kalman_block": blk, "kalman_head_draws": int(sys.argv[2]) if len(sys.argv) > 2 else 0})
torch.cuda.synchronize()
