"""GPU box: how far is the dispatch order of the Kalman launch (slow draws first, keyed by the cycle-reduction iteration
count) from the best possible one (draws sorted by their true number of full filter steps)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import _lib, batched, workloads as wl
from geconpy_amd.engine import LogpEngine
nb = 4096
b = wl.sw_shaped_batch(nb); om = wl.sw_shaped_observation_model()
eng = LogpEngine(torch.device("cuda", 0)); lib = _lib.load()
dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
def timed(perm, label, opts):
    dA, dB, dC, dD = (eng.to_device(b[x][perm]) for x in "ABCD"); dq = eng.to_device((b["sigma"] ** 2)[perm])
    hints = eng.structure_hints(dA, dZ)
    buf = torch.full((nb,), -2, dtype=torch.int32, device="cuda")
    with _lib.options_scope(opts):
        eng.record_steady_steps(buf)
        run = lambda: eng.solve_kalman_logp(dA, dB, dC, dD, dq, dZ, dy, Hdiag=dH, tol=1e-8, max_iter=1000, n_state_hint=hints[0], z_selector_hint=hints[1])
        run(); torch.cuda.synchronize()
        eng.record_steady_steps(None)
        pk = eng.profile_kernels(dA, dB, dC, dD, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, n_state_hint=hints[0], z_selector_hint=hints[1])
    st = buf.cpu().numpy()
    print(f"{label:44s}: kalman {pk['kalman']:.3f} ms, solver {pk['solver']:.3f} ms")
    return st
ident = np.arange(nb)
st = timed(ident, "library order (CR iteration count)", {})
key = np.where(st < 0, 999, st)
timed(ident, "index order (kalman_order = 0)", {"kalman_order": 0})
timed(np.argsort(-key, kind="stable"), "true slowest first (kalman_order = 0)", {"kalman_order": 0})
timed(ident, "persistence key (kalman_order = 2)", {"kalman_order": 2})
T, status, n_iter = batched.cycle_reduction_batched(b["A"], b["B"], b["C"], max_iter=1000, tol=1e-8)
slow = np.argsort(-key)[:12]
print("slowest draws:", slow.tolist(), "full steps", key[slow].tolist(), "CR iterations", n_iter[slow].tolist())
print("CR iteration histogram:", np.bincount(n_iter).tolist())
