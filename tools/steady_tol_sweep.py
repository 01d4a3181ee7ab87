"""GPU box: the headline step under a range of dsge_options.kalman_steady_tol -- rate and the change of logp against the full
recursion (kalman_steady_tol = 0)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from geconpy_amd import batched, workloads as wl
from geconpy_amd.engine import LogpEngine
from _opts import set_option  # (tools/_opts.py: per-thread dsge_options)
nb = 4096
b = wl.sw_shaped_batch(nb); om = wl.sw_shaped_observation_model()
eng = LogpEngine(torch.device("cuda", 0))
dA, dB, dC, dD = (eng.to_device(b[x]) for x in "ABCD"); dq = eng.to_device(b["sigma"] ** 2)
dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
hints = eng.structure_hints(dA, dZ)
def run():
    return eng.solve_kalman_logp(dA, dB, dC, dD, dq, dZ, dy, Hdiag=dH, tol=1e-8, max_iter=1000, n_state_hint=hints[0], z_selector_hint=hints[1])
ref = None
for tol in (0.0, 1e-14, 1e-13, 1e-12, 1e-11, 1e-10):
    set_option("kalman_steady_tol", tol)
    out = run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): out = run()
    e1.record(); torch.cuda.synchronize()
    lp = out[0].cpu().numpy() if isinstance(out, tuple) else out["logp"].cpu().numpy()
    if ref is None: ref = lp
    ok = np.isfinite(ref)
    print(f"tol {tol:g}: {e0.elapsed_time(e1)/5:.3f} ms/step, max rel diff vs tol=0: {np.max(np.abs(lp[ok]-ref[ok])/np.abs(ref[ok])):.2e}")
set_option("kalman_steady_tol", 1e-14)
