"""GPU box: rate of the 65 .. 96-variable path (csrc/dsge_big.hpp): cycle reduction alone and the fused solve + Kalman logp, on
SW-shaped systems scaled to n variables.  python tools/big_rate.py [n] [batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from geconpy_amd import workloads as wl
from geconpy_amd.engine import LogpEngine

SHAPES = {72: dict(n_state=30, n_lead=20, k=8), 80: dict(n_state=36, n_lead=24, k=10), 96: dict(n_state=44, n_lead=30, k=12)}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
sh = SHAPES[n]
nd = min(NB, 256)
b = wl.sw_shaped_batch(nd, n=n, p=7, T_len=200, **sh)
om = wl.sw_shaped_observation_model(n=n, p=7, T_len=200, **sh)
rep = (NB + nd - 1) // nd
eng = LogpEngine(0)
A, B, C, D = (eng.to_device(np.tile(b[x], (rep, 1, 1))[:NB]) for x in "ABCD")
q = eng.to_device(np.tile(b["sigma"] ** 2, (rep, 1))[:NB])
Z, y, H = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
lp = torch.empty(NB, dtype=torch.float64, device="cuda")
st = torch.empty(NB, dtype=torch.int32, device="cuda")


def call():
    eng.solve_kalman_logp(A, B, C, D, q, Z, y, Hdiag=H, q_mode=1, tol=1e-8, max_iter=1000, solver=(sys.argv[3] if len(sys.argv) > 3 else "cycle_reduction"), z_selector_hint=1,
                          logp=lp, status=st)


for _ in range(2):
    call()
torch.cuda.synchronize()
t0 = time.perf_counter()
R = 5
for _ in range(R):
    call()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / R
print(f"n = {n}: {NB} draws, {dt * 1e3:.2f} ms per solve + Kalman logp batch = {NB / dt / 1e3:.1f} k evals/s; failed {int((st != 0).sum())}")
