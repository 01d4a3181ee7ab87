"""Distribution of the number of full filter steps over the 4096 SW-shaped draws + a list-scheduling model of the Kalman launch."""
import sys, heapq
import numpy as np, torch
sys.path.insert(0, ".")
from geconpy_amd import batched, workloads as wl
from geconpy_amd.engine import LogpEngine
nb = 4096
b = wl.sw_shaped_batch(nb)
om = wl.sw_shaped_observation_model()
eng = LogpEngine(0)
dev = [eng.to_device(b[x]) for x in "ABCD"]
dq = eng.to_device(b["sigma"] ** 2)
dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
ns, zs = eng.structure_hints(dev[0], dZ)
at = torch.full((nb,), -7, dtype=torch.int32, device="cuda")
eng.record_steady_steps(at)
eng.solve_kalman_logp(*dev, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, n_state_hint=ns, z_selector_hint=zs)
torch.cuda.synchronize()
eng.record_steady_steps(None)
a = at.cpu().numpy()
full = np.where(a < 0, 200, a)
it = batched.cycle_reduction_batched(b["A"], b["B"], b["C"], max_iter=1000, tol=1e-8)[2]
print("full steps: mean %.1f median %d  p90 %d p99 %d max %d; >=100: %d, >=150: %d" % (full.mean(), np.median(full), np.percentile(full, 90), np.percentile(full, 99), full.max(), (full >= 100).sum(), (full >= 150).sum()))
print("hist (bins of 20):", np.histogram(full, bins=np.arange(0, 221, 20))[0].tolist())
for k in np.unique(it):
    f = full[it == k]
    print(f"CR iterations {k}: {len(f)} draws, full steps mean {f.mean():.1f} max {f.max()}")
np.save("gpurun_out/full_steps.npy", full); np.save("gpurun_out/cr_iters.npy", it)
for cf, cs in ((8.9e3, 1.2e3), (10.4e3, 1.0e3)):
    cost = full * cf + (200 - full) * cs
    for name, order in (("index", np.arange(nb)), ("cr-iters desc", np.argsort(-it, kind="stable")), ("hindsight", np.argsort(-cost))):
        slots = [0.0] * 2048
        heapq.heapify(slots)
        end = 0.0
        for i in order:
            t0 = heapq.heappop(slots)
            t1 = t0 + cost[i]
            end = max(end, t1)
            heapq.heappush(slots, t1)
        print(f"  step cost full {cf:.0f} steady {cs:.0f}: order {name:14s} makespan {end / 1e6:.3f} M cycles; sum/2048 = {cost.sum() / 2048 / 1e6:.3f}; max draw {cost.max() / 1e6:.3f}")
