"""GPU box: fused evaluations IN FLIGHT ON SEVERAL STREAMS of one GPU (the library keeps its scratch per (device, stream):
include/dsge_hip.h "re-entrant per stream") -- what two PyMC chains sharing a GPU, or a sampler that splits its particles,
get.  One launch sequence leaves the chip idle while its slowest draws finish (the Kalman launch ends with ONE wavefront);
a second sequence on another stream fills that time.  Prints whole-GPU evals/s for
  1 stream x 4096,  2 streams x 4096,  2 streams x 2048 (one 4096-draw batch split in halves),  4 streams x 2048."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from geconpy_amd import workloads as wl
from geconpy_amd.engine import LogpEngine

NB = 4096
eng = LogpEngine(0)
b = wl.sw_shaped_batch(NB)
om = wl.sw_shaped_observation_model()
A, B, C, D = (eng.to_device(b[x]) for x in "ABCD")
q = eng.to_device(b["sigma"] ** 2)
Z, y, H = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
hints = eng.structure_hints(A, Z)
opts = {"n_static_hint": eng.static_hint(A, C)}


def run(n_streams, per_stream, reps=30):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    outs = [(torch.empty(per_stream, dtype=torch.float64, device="cuda"), torch.empty(per_stream, dtype=torch.int32, device="cuda"))
            for _ in range(n_streams)]
    sl = [slice((i * per_stream) % NB, (i * per_stream) % NB + per_stream) for i in range(n_streams)]

    def enqueue(i):
        with torch.cuda.stream(streams[i]):
            eng.solve_kalman_logp(A[sl[i]], B[sl[i]], C[sl[i]], D[sl[i]], q[sl[i]], Z, y, Hdiag=H, q_mode=1, tol=1e-8, max_iter=1000,
                                  n_state_hint=hints[0], z_selector_hint=hints[1], logp=outs[i][0], status=outs[i][1], options=opts)

    for _ in range(3):
        for i in range(n_streams):
            enqueue(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for i in range(n_streams):
            enqueue(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = all(int((o[1] != 0).sum()) == 0 for o in outs)
    return reps * n_streams * per_stream / dt, dt / reps * 1e3, ok, outs


ref = None
for n_streams, per in ((1, 4096), (2, 4096), (2, 2048), (4, 2048), (3, 4096)):
    rate, ms, ok, outs = run(n_streams, per)
    if ref is None:
        ref = outs[0][0].clone()
    same = all(torch.equal(o[0], ref[(i * per) % NB:(i * per) % NB + per]) for i, o in enumerate(outs))
    print(f"{n_streams} stream(s) x {per} draws: {rate / 1e6:.3f} M evals/s ({ms:.3f} ms per round of {n_streams * per} draws); "
          f"all status 0: {ok}; logp bit-identical to the one-stream run: {same}", flush=True)
