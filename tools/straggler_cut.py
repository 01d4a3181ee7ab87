"""Is the Kalman launch bound by its slowest draws?  Replace the draws with >= thr full steps by copies of draw 0 and time the stages."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from geconpy_amd import _lib, workloads as wl
from geconpy_amd.engine import LogpEngine
nb = 4096
b = wl.sw_shaped_batch(nb)
om = wl.sw_shaped_observation_model()
full = np.load("profiles/r5/full_steps_4096.npy")
eng = LogpEngine(0)
dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
for thr in (300, 150, 100, 70, 50):
    idx = np.arange(nb)
    idx[full >= thr] = 0
    dev = [eng.to_device(b[x][idx]) for x in "ABCD"]
    dq = eng.to_device((b["sigma"] ** 2)[idx])
    ns, zs = eng.structure_hints(dev[0], dZ)
    for head in (0, 256):
        with _lib.options_scope({"kalman_head_draws": head}):
            ms = eng.profile_kernels(*dev, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, reps=10, n_state_hint=ns, z_selector_hint=zs)
        print(f"draws with >= {thr} full steps replaced ({int((full >= thr).sum())}); head {head}:", {k: round(v, 4) for k, v in ms.items()}, flush=True)
# the solver launch and the nearly singular draw 752 (all iterations refined in doubled precision)
for drop in ((), (752,), tuple(range(700, 800))):
    idx = np.arange(nb)
    idx[list(drop)] = 0
    dev = [eng.to_device(b[x][idx]) for x in "ABCD"]
    dq = eng.to_device((b["sigma"] ** 2)[idx])
    ns, zs = eng.structure_hints(dev[0], dZ)
    ms = eng.profile_kernels(*dev, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, reps=10, n_state_hint=ns, z_selector_hint=zs)
    print(f"draws {drop[:3]}{'...' if len(drop) > 3 else ''} replaced by draw 0 ({len(drop)}):", {k: round(v, 4) for k, v in ms.items()}, flush=True)
# batch-size scaling of the two launches (is a launch a fixed tail + throughput?)
for n2 in (512, 1024, 2048, 3072, 4096, 6144, 8192):
    bb = wl.sw_shaped_batch(n2)
    dev = [eng.to_device(bb[x]) for x in "ABCD"]
    dq = eng.to_device(bb["sigma"] ** 2)
    ms = eng.profile_kernels(*dev, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, reps=10, n_state_hint=ns, z_selector_hint=zs)
    print(f"batch {n2}:", {k: round(v, 4) for k, v in ms.items()}, flush=True)
