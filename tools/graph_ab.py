"""GPU box: the headline step eager against replayed from a HIP graph (torch.cuda.CUDAGraph around the fused C-ABI call: the library
only enqueues on the caller's stream when n_static_hint is given, so the call is capturable).  One step = 11 launches, seven of them
empty second passes of ~5 us each: the graph replays them back to back."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import workloads as wl
from geconpy_amd.engine import LogpEngine
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
solver = sys.argv[2] if len(sys.argv) > 2 else "cycle_reduction"
b = wl.sw_shaped_batch(nb); om = wl.sw_shaped_observation_model()
eng = LogpEngine(0)
dev = [eng.to_device(b[x]) for x in "ABCD"]
q = eng.to_device(b["sigma"] ** 2); Z = eng.to_device(om["Z"]); y = eng.to_device(om["y"]); H = eng.to_device(om["Hdiag"])
ns, zs = eng.structure_hints(dev[0], Z)
hs = eng.static_hint(dev[0], dev[2])
nl = int((np.abs(b["C"][0]).sum(axis=0) > 1e-8).sum())
lp = torch.empty(nb, dtype=torch.float64, device="cuda"); st = torch.empty(nb, dtype=torch.int32, device="cuda")
opts = {"n_static_hint": hs}
f = lambda: eng.solve_kalman_logp(*dev, q, Z, y, Hdiag=H, q_mode=1, tol=1e-8, max_iter=1000, logp=lp, status=st, solver=solver,
                                  n_state_hint=ns, z_selector_hint=zs, n_lead_hint=nl, options=opts)
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
dt = timeit(f)
ref = lp.cpu().numpy().copy()
print(f"eager: {dt*1e3:.4f} ms per step, {nb/dt/1e6:.3f} M evals/s")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): f()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s, capture_error_mode="relaxed"):
    f()
lp.zero_()
dt = timeit(g.replay)
print(f"graph: {dt*1e3:.4f} ms per step, {nb/dt/1e6:.3f} M evals/s; logp identical to eager: {np.array_equal(lp.cpu().numpy(), ref)}; failed {int((st != 0).sum())}")
