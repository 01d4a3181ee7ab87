"""Headline step against dsge_options.pipeline_chunks / kalman_head_draws (one call, one caller stream)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from geconpy_amd import workloads as wl
from geconpy_amd.engine import LogpEngine
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b = wl.sw_shaped_batch(nb)
om = wl.sw_shaped_observation_model()
eng = LogpEngine(0)
dev = [eng.to_device(b[x]) for x in "ABCD"]
dq = eng.to_device(b["sigma"] ** 2)
dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
ns, zs = eng.structure_hints(dev[0], dZ)
hs = eng.static_hint(dev[0], dev[2])
lp = torch.empty(nb, dtype=torch.float64, device="cuda")
st = torch.empty(nb, dtype=torch.int32, device="cuda")
for chunks, head in ((0, 0), (2, 0), (3, 0), (4, 0), (2, 32), (4, 32)):
    opts = {"n_static_hint": hs, "pipeline_chunks": chunks, "kalman_head_draws": head}
    f = lambda: eng.solve_kalman_logp(*dev, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, logp=lp, status=st,
                                      n_state_hint=ns, z_selector_hint=zs, options=opts)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40
    print(f"chunks {chunks} head {head}: {dt * 1e3:.4f} ms per step  {nb / dt / 1e6:.3f} M evals/s  failed {int((st != 0).sum())}", flush=True)
