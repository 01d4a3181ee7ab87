"""GPU box: does the ORDER of the draws in the batch change the fused evaluation time?  (The Kalman launch's makespan is
set by its slowest draws; the hardware dispatches workgroups in index order.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import _lib, batched, workloads as wl
from geconpy_amd.engine import LogpEngine
from _opts import set_option  # (tools/_opts.py: per-thread dsge_options)
nb = 4096
b = wl.sw_shaped_batch(nb); om = wl.sw_shaped_observation_model()
eng = LogpEngine(torch.device("cuda", 0)); lib = _lib.load()
set_option("pipeline_chunks", 0); set_option("kalman_block", 0)
dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
def timed(perm, label):
    dA, dB, dC, dD = (eng.to_device(b[x][perm]) for x in "ABCD"); dq = eng.to_device((b["sigma"] ** 2)[perm])
    hints = eng.structure_hints(dA, dZ)
    buf = torch.full((nb,), -2, dtype=torch.int32, device="cuda")
    eng.record_steady_steps(buf)
    run = lambda: eng.solve_kalman_logp(dA, dB, dC, dD, dq, dZ, dy, Hdiag=dH, tol=1e-8, max_iter=1000, n_state_hint=hints[0], z_selector_hint=hints[1])
    out = run(); torch.cuda.synchronize()
    eng.record_steady_steps(None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): out = run()
    e1.record(); torch.cuda.synchronize()
    st = buf.cpu().numpy()
    print(f"{label:34s}: {e0.elapsed_time(e1)/10:.3f} ms/step; never steady at positions {np.flatnonzero(st < 0).tolist()[:6]}, latest 5 at {np.argsort(-np.where(st<0,999,st))[:5].tolist()}")
    return st
ident = np.arange(nb)
st = timed(ident, "original order")
timed(ident[::-1].copy(), "reversed")
key = np.where(st < 0, 999, st)
timed(np.argsort(-key, kind="stable"), "slowest first (oracle knowledge)")
timed(np.argsort(key, kind="stable"), "slowest last")
# predictor: cycle-reduction iteration count
T, status, n_iter = batched.cycle_reduction_batched(b["A"], b["B"], b["C"], max_iter=1000, tol=1e-8)
timed(np.argsort(-n_iter, kind="stable"), "most CR iterations first")
print("corr(first steady step, CR iterations) =", np.corrcoef(key, n_iter)[0, 1])
