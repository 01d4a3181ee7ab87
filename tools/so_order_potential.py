"""GPU box: how good is the dispatch order of the second-order filter launch?  Per draw: number of full covariance steps of the
pruned filter (its cost), the current key (cycle-reduction iteration count), and a candidate key -- the step at which the
FIRST-order filter of the same draw reaches its steady state (data-independent, one 0.3 ms launch).  List-scheduling
simulation on 256 CUs (one workgroup per CU) with the measured per-step costs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from geconpy_amd import batched, workloads as wl
from geconpy_amd.engine import LogpEngine

nb = 1024
b = wl.sw_second_order_batch(nb)
om = wl.sw_shaped_observation_model()
eng = LogpEngine(0)
A, B, C, D = (eng.to_device(b[x]) for x in "ABCD")
q = eng.to_device(b["sigma"] ** 2)
hv = eng.to_device(b["hess_val"])
hi = torch.as_tensor(b["hess_idx"], dtype=torch.int32, device=eng.device).contiguous()
Z, y, H = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
st = eng.second_order_structure(A, C, Z)
at2 = torch.full((nb,), -1, dtype=torch.int32, device=eng.device)
eng.record_steady_steps(at2)
eng.second_order_logp(A, B, C, D, hi, hv, q, Z, y, st, Hdiag=H, tol=1e-8)
torch.cuda.synchronize()
at1 = torch.full((nb,), -1, dtype=torch.int32, device=eng.device)
eng.record_steady_steps(at1)
ns, zs = eng.structure_hints(A, Z)
eng.solve_kalman_logp(A, B, C, D, q, Z, y, Hdiag=H, q_mode=1, tol=1e-8, max_iter=1000, n_state_hint=ns, z_selector_hint=zs)
torch.cuda.synchronize()
eng.record_steady_steps(None)
_, _, it = batched.cycle_reduction_batched(b["A"], b["B"], b["C"], tol=1e-8, max_iter=1000)
f2 = at2.cpu().numpy(); f2 = np.where(f2 < 0, 200, f2)
f1 = at1.cpu().numpy(); f1 = np.where(f1 < 0, 200, f1)
cost = f2 * 713e3 + (200 - f2) * 33e3  # cycles (tools/so_rate.py phase stamps)
print("full steps of the pruned filter: mean %.1f, never steady %d; first-order filter: mean %.1f, never steady %d" %
      (f2.mean(), (f2 == 200).sum(), f1.mean(), (f1 == 200).sum()))
print("rank correlation with the pruned filter's full steps: CR iterations %.3f, first-order steady step %.3f" %
      (np.corrcoef(np.argsort(np.argsort(it)), np.argsort(np.argsort(f2)))[0, 1],
       np.corrcoef(np.argsort(np.argsort(f1)), np.argsort(np.argsort(f2)))[0, 1]))


def makespan(order, machines=256):
    t = np.zeros(machines)
    for i in order:
        j = np.argmin(t)
        t[j] += cost[i]
    return t.max() / 2.1e9 * 1e3


print("simulated launch (ms at 2.1 GHz): index order %.1f, CR-iteration key %.1f, first-order steady-step key %.1f, "
      "exact (LPT) %.1f, lower bound %.1f" % (makespan(np.arange(nb)), makespan(np.argsort(-it, kind="stable")),
                                              makespan(np.argsort(-f1, kind="stable")), makespan(np.argsort(-cost, kind="stable")),
                                              max(cost.max(), cost.sum() / 256) / 2.1e9 * 1e3))
rho = np.array([np.abs(np.linalg.eigvals(b["T_star"][i])).max() for i in range(nb)])
print("simulated launch with the spectral radius of T as the key: %.1f ms; rank correlation %.3f" %
      (makespan(np.argsort(-rho, kind="stable")), np.corrcoef(np.argsort(np.argsort(rho)), np.argsort(np.argsort(f2)))[0, 1]))
nv = np.flatnonzero(f2 == 200)
print("never-steady draws: CR iterations", it[nv].tolist(), "\n  first-order steady step", f1[nv].tolist(), "\n  rho(T)", np.round(rho[nv], 3).tolist())
print("rho(T) quantiles of all draws", np.round(np.quantile(rho, [0.5, 0.9, 0.95, 0.98, 1.0]), 3).tolist(), "; draws with rho >= min over never-steady:", int((rho >= rho[nv].min()).sum()))
top = np.argsort(-f2)[:64]
print("64 slowest: CR it", np.bincount(it[top]).tolist(), "of all", np.bincount(it).tolist())
