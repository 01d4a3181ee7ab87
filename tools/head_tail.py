"""The slowest draws alone: Kalman stage time of the one- and the two-wavefront kernel on 64 draws starting at draw 3437 (the
never-steady one), no bulk around them -- the floor a launch's tail can reach."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from geconpy_amd import _lib, workloads as wl
from geconpy_amd.engine import LogpEngine
first, nb = 3437, 64
b = wl.sw_shaped_batch(nb, first_draw=first)
om = wl.sw_shaped_observation_model()
eng = LogpEngine(0)
dev = [eng.to_device(b[x]) for x in "ABCD"]
dq = eng.to_device(b["sigma"] ** 2)
dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
ns, zs = eng.structure_hints(dev[0], dZ)
for head in (0, -1):
    with _lib.options_scope({"kalman_head_draws": head}):
        ms = eng.profile_kernels(*dev, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, reps=10, n_state_hint=ns, z_selector_hint=zs)
    print("head", head, {k: round(v, 4) for k, v in ms.items()})
