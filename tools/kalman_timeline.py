"""GPU box: the schedule of one Kalman launch of the headline step (dsge_debug_kalman_timeline): start / end of every draw on the
100 MHz wall clock, the SIMD it ran on, its number of full steps.  Prints the launch's span, the distribution of draw durations, how
many wavefronts are resident over time and what the last ones to finish are."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import _lib, workloads as wl
from geconpy_amd.engine import LogpEngine
if os.environ.get("DSGE_TEST_LIB"):  # (A/B of a differently built library)
    _lib.LIB_PATH = os.path.abspath(os.environ["DSGE_TEST_LIB"])
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
b = wl.sw_shaped_batch(nb); om = wl.sw_shaped_observation_model()
eng = LogpEngine(0); lib = _lib.load()
dev = [eng.to_device(b[x]) for x in "ABCD"]
dq = eng.to_device(b["sigma"] ** 2); dZ, dy, dH = eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
ns, zs = eng.structure_hints(dev[0], dZ)
lp = torch.empty(nb, dtype=torch.float64, device="cuda"); st = torch.empty(nb, dtype=torch.int32, device="cuda")
f = lambda: eng.solve_kalman_logp(*dev, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, logp=lp, status=st, n_state_hint=ns, z_selector_hint=zs)
for _ in range(5):
    f()
torch.cuda.synchronize()
tl = torch.zeros(nb, 8, dtype=torch.int64, device="cuda")
_lib.check(lib.dsge_debug_kalman_timeline(tl.data_ptr()))
f(); torch.cuda.synchronize()
_lib.check(lib.dsge_debug_kalman_timeline(None))
t = tl.cpu().numpy()
os.makedirs('gpurun_out', exist_ok=True); np.save(f'gpurun_out/kalman_timeline_{nb}.npy', t)
start, end, hw, sst = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
t0 = start.min(); s_us = (start - t0) / 100.0; e_us = (end - t0) / 100.0; dur = e_us - s_us
full = np.where(sst < 0, 200, sst)
print(f"{nb} draws: launch span {e_us.max():.1f} us; draw duration us: min {dur.min():.0f} median {np.median(dur):.0f} p90 {np.percentile(dur, 90):.0f} max {dur.max():.0f}; "
      f"sum of durations / 2048 slots = {dur.sum() / 2048:.1f} us")
first = s_us < 20
print(f"draws started in the first 20 us: {first.sum()}; last start at {s_us.max():.1f} us")
for q in (0.25, 0.5, 0.75, 0.9, 0.95, 0.99, 1.0):
    print(f"  {int(q * 100)} % of the draws have finished by {np.quantile(e_us, q):.1f} us")
grid = np.arange(0, e_us.max() + 25, 25.0)
res = [(int(((s_us <= g) & (e_us > g)).sum())) for g in grid]
print("resident wavefronts every 25 us:", res)
last = np.argsort(-e_us)[:12]
print("last to finish (draw, start, end, full steps):", [(int(i), round(float(s_us[i]), 1), round(float(e_us[i]), 1), int(full[i])) for i in last])
# cost model: duration ~ a * full + b * steady + c, least squares over the draws
X = np.stack([full, 200 - full, np.ones(nb)], axis=1).astype(float)
coef, *_ = np.linalg.lstsq(X, dur, rcond=None)
print(f"least squares: {coef[0]:.3f} us per full step, {coef[1]:.3f} us per steady step, {coef[2]:.1f} us fixed (prologue: P0 by doubling, loads)")
pro = (t[:, 4] - start) / 100.0
has = t[:, 5] > 0
fullphase = np.where(has, (t[:, 5] - t[:, 4]) / 100.0, (end - t[:, 4]) / 100.0)
steadyphase = np.where(has, (end - t[:, 5]) / 100.0, 0.0)
print(f"prologue (reduction, R Q R', P0 by doubling) us: median {np.median(pro):.1f} p90 {np.percentile(pro, 90):.1f}; full-step phase: median {np.median(fullphase):.1f} = {np.median(fullphase / np.maximum(full, 1)):.2f} us per full step; steady phase: median {np.median(steadyphase):.1f} = {np.median(steadyphase[has] / np.maximum(200 - full[has], 1)):.3f} us per steady step")
print(f"shares of the wavefront-time: prologue {pro.sum() / dur.sum():.3f}, full steps {fullphase.sum() / dur.sum():.3f}, steady steps {steadyphase.sum() / dur.sum():.3f}")
print("distinct (wave, simd, cu...) ids:", len(np.unique(hw)))
