"""GPU box: the `conservative` leg of bench.py (observed jump variables, 10 % of y missing at random, kalman_steady_tol = 0) a few
times, for rocprofv3 --kernel-trace --stats; prints the stage times.  argv[1]: 0 = with the missing entries, 1 = complete data."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import _lib, workloads as wl
from geconpy_amd.engine import LogpEngine
from geconpy_amd.batched import lead_hint
complete = len(sys.argv) > 1 and sys.argv[1] == "1"
nb = 4096
b = wl.sw_shaped_batch(nb)
om = wl.sw_shaped_observation_model(observed=wl.SW_OBSERVED_JUMPS)
y = om["y"].copy()
if not complete:
    rng = np.random.default_rng(20261003)
    y[rng.random(y.shape) < 0.10] = np.nan
eng = LogpEngine(0)
dev = [eng.to_device(b[x]) for x in "ABCD"]
dq = eng.to_device(b["sigma"] ** 2)
dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(y), eng.to_device(om["Hdiag"])
ns, zs = eng.structure_hints(dev[0], dZ)
nl = lead_hint(b["C"], 1e-8)
for solver, opts in (("gensys", {"kalman_steady_tol": 0.0, "gensys_doubling": 0}), ("cycle_reduction", {"kalman_steady_tol": 0.0})):
    with _lib.options_scope(opts):
        ms = eng.profile_kernels(*dev, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, reps=5, n_state_hint=ns, z_selector_hint=zs,
                                 solver=solver, n_lead_hint=nl)
    print("complete data" if complete else "10 % missing", solver, {k: round(v, 4) for k, v in ms.items()}, flush=True)
