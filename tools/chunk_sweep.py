"""GPU box: fused device evaluation (4096 SW-shaped draws) vs the number of pipeline chunks and the tail hand-off."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import _lib, workloads as wl
from geconpy_amd.engine import LogpEngine
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b = wl.sw_shaped_batch(nb); om = wl.sw_shaped_observation_model()
eng = LogpEngine(torch.device("cuda", 0)); lib = _lib.load()
dA, dB, dC, dD = (eng.to_device(b[x]) for x in "ABCD"); dq = eng.to_device(b["sigma"] ** 2)
dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
hints = eng.structure_hints(dA, dZ)
def run():
    return eng.solve_kalman_logp(dA, dB, dC, dD, dq, dZ, dy, Hdiag=dH, tol=1e-8, max_iter=1000, n_state_hint=hints[0], z_selector_hint=hints[1])
ref = None
for order, blk, ch in ((0, 0, 0), (1, 0, 0), (2, 0, 0), (2, 1, 0), (2, 0, 2)):
    if True:
        _lib.check(lib.dsge_set_kalman_order(order)); _lib.check(lib.dsge_set_kalman_block(blk))
        _lib.check(lib.dsge_set_pipeline_chunks(ch))
        out = run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): out = run()
        e1.record(); torch.cuda.synchronize()
        lp = out[0].cpu().numpy()
        if ref is None: ref = lp
        ok = np.isfinite(ref)
        print(f"order {order} tail hand-off {blk} chunks {ch:2d}: {e0.elapsed_time(e1)/10:.3f} ms/step, max rel diff {np.max(np.abs(lp[ok]-ref[ok])/np.abs(ref[ok])):.1e}")
_lib.check(lib.dsge_set_kalman_block(1)); _lib.check(lib.dsge_set_pipeline_chunks(0)); _lib.check(lib.dsge_set_kalman_order(1))
