"""GPU box: fused pipeline timings (ms per 4096 draws) across model sizes, SW-like structure (45% states, 30% leads)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import workloads as wl
from geconpy_amd.engine import LogpEngine
eng = LogpEngine(0)
nb = 4096
for n in (tuple(int(x) for x in sys.argv[1:]) if len(sys.argv) > 1 else (8, 16, 24, 32, 40, 48, 56, 64)):
    ns, nl, k, p = max(2, int(0.45 * n)), max(1, int(0.3 * n)), min(7, n // 2), min(7, n // 2)
    base = [wl.sw_shaped_system(1000 + i, n=n, n_state=ns, n_lead=nl, k=k) for i in range(16)]
    rep = nb // 16
    A, B, C, D = (eng.to_device(np.tile(np.stack([b[x] for b in base]), (rep, 1, 1))) for x in range(4))
    q = eng.to_device(np.full((nb, k), 1e-4))
    Z = np.zeros((p, n)); Z[np.arange(p), np.arange(p)] = 1.0
    y = np.random.default_rng(0).normal(0, 0.02, (200, p))
    dZ, dy, dH = eng.to_device(Z), eng.to_device(y), eng.to_device(np.full(p, 1e-4))
    hints = eng.structure_hints(A, dZ)
    try:
        eng.profile_kernels(A, B, C, D, q, dZ, dy, Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, reps=1,
                            n_state_hint=hints[0], z_selector_hint=hints[1])  # warm-up (code objects, arenas)
        ms = eng.profile_kernels(A, B, C, D, q, dZ, dy, Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, reps=2,
                                 n_state_hint=hints[0], z_selector_hint=hints[1])
        lp, st = eng.solve_kalman_logp(A, B, C, D, q, dZ, dy, Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000,
                                       n_state_hint=hints[0], z_selector_hint=hints[1])
        torch.cuda.synchronize()
        tot = sum(ms.values())
        print(f"n={n:2d} states={ns:2d} leads={nl:2d} p={p}: solver {ms['solver']:.3f}  assemble {ms['assemble']:.3f}  kalman {ms['kalman']:.3f}  "
              f"total {tot:.3f} ms = {nb / tot * 1e3:,.0f} evals/s; failed {int((st != 0).sum())}")
    except Exception as e:  # noqa: BLE001
        print(f"n={n}: {e}")
