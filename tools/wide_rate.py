"""GPU box: cycle reduction on 49..64 variables WITHOUT the static-variable deflation -- one wavefront per draw
(cr_compact_kernel<7|8>) against four (cr_wide_kernel), 4096 draws, plus the whole fused evaluation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import _lib, workloads as wl
from geconpy_amd.engine import LogpEngine
nb = 4096
eng = LogpEngine(torch.device("cuda", 0))
for n, ns, nl in ((44, 20, 13), (48, 21, 14), (50, 22, 15), (56, 25, 16), (64, 28, 19)):
    k = p = 7
    sysm = [wl.sw_shaped_system(7000 + 5 * n + i, n=n, n_state=ns, n_lead=nl, k=k) for i in range(64)]
    A, B, C, D = (np.tile(np.stack([s_[j] for s_ in sysm]), (nb // 64, 1, 1)) for j in range(4))
    q = np.full((nb, k), 1e-4)
    Z = np.zeros((p, n)); Z[np.arange(p), np.arange(p)] = 1.0
    y = np.random.default_rng(n).normal(0, 0.02, (200, p)); H = np.full(p, 1e-4)
    dev = [eng.to_device(x) for x in (A, B, C, D)]
    dq, dZ, dy, dH = eng.to_device(q), eng.to_device(Z), eng.to_device(y), eng.to_device(H)
    hints = eng.structure_hints(dev[0], dZ)
    for label, opts in (("one wave, no deflation", {"cr_deflation": 0, "cr_four_waves": 0}),
                        ("four waves, no deflation", {"cr_deflation": 0, "cr_four_waves": 1}),
                        ("deflation, three launches, one wave", {"cr_fused_deflation": 0, "cr_four_waves": 0}),
                        ("deflation, three launches, four waves", {"cr_fused_deflation": 0, "cr_four_waves": 1}),
                        ("default (deflation, one launch)", {})):
        with _lib.options_scope(opts):
            pk = eng.profile_kernels(*dev, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, n_state_hint=hints[0],
                                     z_selector_hint=hints[1])
        tot = pk["solver"] + pk["assemble"] + pk["kalman"]
        print(f"n={n:2d} {label:38s}: solver {pk['solver']:.3f} ms, kalman {pk['kalman']:.3f} ms -> {nb / tot:.0f} k evals/s")
