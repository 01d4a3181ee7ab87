"""GPU box: Kalman kernel time with a dense design matrix (observation-equation models) on SW-shaped draws."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import workloads as wl
from geconpy_amd.engine import LogpEngine
nb = 4096
b = wl.sw_shaped_batch(64); om = wl.sw_shaped_observation_model()
rep = nb // 64
eng = LogpEngine(0)
dev = [eng.to_device(np.tile(b[x], (rep, 1, 1))) for x in "ABCD"]
q = eng.to_device(np.tile(b["sigma"] ** 2, (rep, 1))); y = eng.to_device(om["y"]); H = eng.to_device(om["Hdiag"])
rng = np.random.default_rng(0)
Zd = om["Z"] + 0.05 * rng.standard_normal(om["Z"].shape)
for name, Z, sel in (("selector", om["Z"], 1), ("dense", Zd, 0)):
    Zt = eng.to_device(Z)
    ms = eng.profile_kernels(*dev, q, Zt, y, Hdiag=H, q_mode=1, tol=1e-8, max_iter=1000, reps=3, n_state_hint=18, z_selector_hint=sel)
    print(name, ms)
