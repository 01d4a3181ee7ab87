"""GPU box: per-call latency of the drop-in Ops.  PyMC evaluates ONE point per logp call (SURVEY F6;
gEconpy/solvers/gensys.py:657-666 is the Op.perform this replaces), so what a sampler sees is the time of one ``perform`` at
batch 1: H2D staging, the launches, D2H.  Median over repetitions of ``HipSolveKalmanLogp.perform`` and
``HipCycleReduction.perform`` (+ the gradient Op) at batch 1, 8, 64 on the SW-shaped model (configs[2]), beside the CPU
oracle's time for one draw on one host core."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from geconpy_amd import pytensor_ops as ops, workloads as wl


def perform(op, inputs, n_out):
    outs = [[None] for _ in range(n_out)]
    op.perform(None, [np.asarray(x) for x in inputs], outs)
    return [o[0] for o in outs]


def median_ms(fn, reps):
    fn(); fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


b = wl.sw_shaped_batch(64); om = wl.sw_shaped_observation_model(); q = b["sigma"] ** 2; d = np.zeros(7)
t0 = time.perf_counter()
for i in range(3):
    oracle.solve_kalman_logp(b["A"][i], b["B"][i], b["C"][i], b["D"][i], np.diag(q[i]), om["Z"], om["y"], H=np.diag(om["Hdiag"]), tol=1e-8, max_iter=1000)
cpu_ms = (time.perf_counter() - t0) / 3 * 1e3
t0 = time.perf_counter()
for i in range(20):
    oracle.cycle_reduction_core(b["A"][i], b["B"][i], b["C"][i], 1000, 1e-8)
cpu_cr_ms = (time.perf_counter() - t0) / 20 * 1e3
print(f"CPU oracle, one draw on one core: solve + Kalman logp {cpu_ms:.1f} ms; cycle reduction alone {cpu_cr_ms:.2f} ms")
logp_op = ops.HipSolveKalmanLogp(solver="cycle_reduction", tol=1e-8, max_iter=1000)
grad_op = ops.HipSolveKalmanLogpGrad(solver="cycle_reduction", tol=1e-8, max_iter=1000)
cr_op = ops.HipCycleReduction(max_iter=1000, tol=1e-8)
crb_op = ops.HipCycleReductionBatched(max_iter=1000, tol=1e-8)
print(f"HipCycleReduction.perform (one system, (n, n) inputs): {median_ms(lambda: perform(cr_op, (b['A'][0], b['B'][0], b['C'][0]), 1), 50):.3f} ms")
for nb in (1, 8, 64):
    args = (b["A"][:nb], b["B"][:nb], b["C"][:nb], b["D"][:nb], q[:nb], om["Z"], om["y"], d, om["Hdiag"])
    ms_l = median_ms(lambda: perform(logp_op, args, 2), 50)
    ms_g = median_ms(lambda: perform(grad_op, args, 7), 30)
    ms_c = median_ms(lambda: perform(crb_op, args[:3], 2), 50)
    print(f"batch {nb:3d}: HipSolveKalmanLogp.perform {ms_l:.3f} ms ({ms_l / nb:.3f} per draw; CPU oracle {cpu_ms:.0f} ms per draw = x{cpu_ms * nb / ms_l:.0f}), "
          f"HipSolveKalmanLogpGrad.perform {ms_g:.3f} ms, HipCycleReductionBatched.perform {ms_c:.3f} ms")

# where the batch-1 call goes: HIP-event durations of the launch groups (device-resident inputs) against the wall time of the
# device-resident call and of the host twin
import torch
from geconpy_amd.engine import LogpEngine

eng = LogpEngine(0)
dev = [eng.to_device(b[x][:1]) for x in "ABCD"]
dq, dZ, dy, dH = eng.to_device(q[:1]), eng.to_device(om["Z"]), eng.to_device(om["y"]), eng.to_device(om["Hdiag"])
ns, zs = eng.structure_hints(dev[0], dZ)
kms = eng.profile_kernels(*dev, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, reps=20, n_state_hint=ns, z_selector_hint=zs)


def dev_call():
    eng.solve_kalman_logp(*dev, dq, dZ, dy, Hdiag=dH, q_mode=1, tol=1e-8, max_iter=1000, n_state_hint=ns, z_selector_hint=zs)
    torch.cuda.synchronize()


ms_dev = median_ms(dev_call, 50)
print(f"batch 1, device-resident inputs: {ms_dev:.3f} ms per call + synchronize; launch groups by HIP events: solver "
      f"{kms['solver']:.3f} ms, assembly {kms['assemble']:.3f} ms, filter {kms['kalman']:.3f} ms (sum {sum(kms.values()):.3f} ms): "
      f"a single draw is ONE wavefront's dependent chain (7 cycle-reduction iterations, ~30 full + ~170 steady filter steps); "
      f"launch and staging overhead is the difference")
