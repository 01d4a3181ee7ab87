"""GPU box: run the fused pipeline with the experimental FP64-MFMA Kalman variant switched on (for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import _lib, workloads as wl
from geconpy_amd.engine import LogpEngine
from _opts import set_option  # (tools/_opts.py: per-thread dsge_options)
nb = 4096
b = wl.sw_shaped_batch(64); om = wl.sw_shaped_observation_model(); rep = nb // 64
eng = LogpEngine(0); lib = _lib.load()
dev = [eng.to_device(np.tile(b[x], (rep, 1, 1))) for x in "ABCD"]
q = eng.to_device(np.tile(b["sigma"] ** 2, (rep, 1))); Z = eng.to_device(om["Z"]); y = eng.to_device(om["y"]); H = eng.to_device(om["Hdiag"])
set_option("kalman_mfma", int(sys.argv[1]) if len(sys.argv) > 1 else 1)
for _ in range(3):
    lp, st = eng.solve_kalman_logp(*dev, q, Z, y, Hdiag=H, q_mode=1, tol=1e-8, max_iter=1000, n_state_hint=18, z_selector_hint=1)
torch.cuda.synchronize(); print("ok", float(lp[0]), int((st != 0).sum()))
