"""GPU box: ONE fused call on one caller stream with dsge_options.pipeline_chunks = 0 / 2 / 3 / 4 (the call splits the batch into
chunks that alternate over library-owned streams: the Kalman tail of one chunk overlaps the solver launch of the next) on the
bench batch (4096 SW-shaped draws) and at 8192 draws; logp must be bit-identical to the unsplit call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from geconpy_amd import workloads as wl
from geconpy_amd.engine import LogpEngine

eng = LogpEngine(0)
om = wl.sw_shaped_observation_model()
SOLVER = sys.argv[1] if len(sys.argv) > 1 else "cycle_reduction"  # (or "gensys": five launches at 2..10 draws per CU each)
print("solver", SOLVER)
for NB in (4096, 8192):
    b = wl.sw_shaped_batch(NB)
    A, B, C, D = (eng.to_device(b[x]) for x in "ABCD")
    q = eng.to_device(b["sigma"] ** 2)
    Z, y, H = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
    hints = eng.structure_hints(A, Z)
    ref = None
    for chunks in (0, 2, 3, 4):
        opts = {"n_static_hint": eng.static_hint(A, C), "pipeline_chunks": chunks}
        lp = torch.empty(NB, dtype=torch.float64, device="cuda")
        st = torch.empty(NB, dtype=torch.int32, device="cuda")

        def call():
            eng.solve_kalman_logp(A, B, C, D, q, Z, y, Hdiag=H, q_mode=1, tol=1e-8, max_iter=1000, solver=SOLVER, n_state_hint=hints[0],
                                  z_selector_hint=hints[1], logp=lp, status=st, options=opts)

        for _ in range(3):
            call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        if ref is None:
            ref = lp.clone()
        print(f"{NB} draws, pipeline_chunks = {chunks}: {dt * 1e3:.3f} ms per call = {NB / dt / 1e6:.3f} M evals/s; failed "
              f"{int((st != 0).sum())}; bit-identical to the unsplit call: {bool(torch.equal(lp, ref))}", flush=True)
