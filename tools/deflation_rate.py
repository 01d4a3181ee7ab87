"""GPU box: fused device evaluation (4096 SW-shaped draws) with the static-variable deflation off / on (argv: 0 1 ...)."""
import sys, time; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import _lib, workloads as wl
from geconpy_amd.engine import LogpEngine
from _opts import set_option  # (tools/_opts.py: per-thread dsge_options)
lib = _lib.load()
nb = 4096
b = wl.sw_shaped_batch(nb); om = wl.sw_shaped_observation_model()
eng = LogpEngine(torch.device("cuda", 0))
dev = {x: eng.to_device(b[x]) for x in "ABCD"}
dq = eng.to_device(b["sigma"] ** 2)
dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
hints = eng.structure_hints(dev["A"], dZ)
modes = [int(x) for x in sys.argv[1:]] or [0, 1]
for on in modes:
    set_option("cr_deflation", on)
    def run():
        return eng.solve_kalman_logp(dev["A"], dev["B"], dev["C"], dev["D"], dq, dZ, dy, Hdiag=dH, tol=1e-8, max_iter=1000,
                                     n_state_hint=hints[0], z_selector_hint=hints[1])
    for _ in range(3): run()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize(); print("deflation", on, "ms/step", (time.perf_counter() - t) / 20 * 1e3)
