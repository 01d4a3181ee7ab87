"""GPU box: fused step time of the bench workload with the batch split into chunks on library-owned streams
(dsge_options.pipeline_chunks): does a chunk's Kalman straggler tail overlap the next chunk's solver?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import workloads as wl
from geconpy_amd.engine import LogpEngine
nb = 4096
b = wl.sw_shaped_batch(nb); om = wl.sw_shaped_observation_model()
eng = LogpEngine(0)
dev = [eng.to_device(b[x]) for x in "ABCD"]; dq = eng.to_device(b["sigma"] ** 2)
dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
ns, zs = eng.structure_hints(dev[0], dZ); nst = eng.static_hint(dev[0], dev[2])
ref = None
for chunks in (0, 2, 3, 4):
    opts = {"pipeline_chunks": chunks, "n_static_hint": nst}
    kw = dict(Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, n_state_hint=ns, z_selector_hint=zs, options=opts)
    for _ in range(3):
        lp, st = eng.solve_kalman_logp(*dev, dq, dZ, dy, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        lp, st = eng.solve_kalman_logp(*dev, dq, dZ, dy, **kw)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    l = lp.cpu().numpy()
    if ref is None: ref = l
    print(f"chunks {chunks}: {dt*1e3:.3f} ms per step, {nb/dt/1e6:.3f} M evals/s, identical results: {np.array_equal(l, ref)}")
