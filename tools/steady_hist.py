"""GPU box: distribution of the first steady step over the 4096 distinct SW-shaped draws of the bench (how long the filter's full
recursion runs per draw: what bounds the launches whose slowest draws start first)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import workloads as wl
from geconpy_amd.engine import LogpEngine
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b = wl.sw_shaped_batch(nb); om = wl.sw_shaped_observation_model()
eng = LogpEngine(0)
dev = [eng.to_device(b[x]) for x in "ABCD"]
q = eng.to_device(b["sigma"] ** 2); Z = eng.to_device(om["Z"]); y = eng.to_device(om["y"]); H = eng.to_device(om["Hdiag"])
at = torch.full((nb,), -7, dtype=torch.int32, device=eng.device)
eng.record_steady_steps(at)
eng.solve_kalman_logp(*dev, q, Z, y, Hdiag=H, q_mode=1, tol=1e-8, max_iter=1000, n_state_hint=18, z_selector_hint=1)
torch.cuda.synchronize(); eng.record_steady_steps(None)
a = at.cpu().numpy()
T_len = om["y"].shape[0]
full = np.where(a < 0, T_len, a)
print(f"{nb} draws: first steady step min {full.min()} median {int(np.median(full))} mean {full.mean():.1f} p90 {int(np.percentile(full, 90))} p99 {int(np.percentile(full, 99))} max {full.max()}; never steady {int((a < 0).sum())}")
print("histogram (full steps: draws):", {f"<{e}": int((full < e).sum()) for e in (25, 30, 40, 50, 75, 100, 150, 200)}, "200:", int((full >= 200).sum()))
