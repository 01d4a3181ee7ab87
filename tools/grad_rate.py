"""GPU box: rate of logp + gradient evaluations (dsge_solve_kalman_logp_grad_batched) on the SW-shaped workload."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geconpy_amd import workloads as wl
from geconpy_amd.engine import LogpEngine
if os.environ.get("DSGE_TEST_LIB"):  # (A/B of a differently built library)
    from geconpy_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ["DSGE_TEST_LIB"])
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
SOLVER = sys.argv[2] if len(sys.argv) > 2 else "cycle_reduction"  # or "gensys", the reference's default estimation solver
OPTS = {"kalman_grad_split": int(sys.argv[3])} if len(sys.argv) > 3 else None  # (A/B of the reverse sweep's arrangements)
nd = min(nb, 4096)  # distinct draws (a tiled small set clusters the draws that take second passes)
b = wl.sw_shaped_batch(nd); om = wl.sw_shaped_observation_model(); rep = (nb + nd - 1) // nd
eng = LogpEngine(0)
A, B, C, D = (eng.to_device(np.tile(b[x], (rep, 1, 1))[:nb]) for x in "ABCD")
q = eng.to_device(np.tile(b["sigma"] ** 2, (rep, 1))[:nb]); Z = eng.to_device(om["Z"]); y = eng.to_device(om["y"]); H = eng.to_device(om["Hdiag"])
def run(A, B, C, D, q):
    out = None
    for it in range(4):
        if it == 1:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        out = eng.solve_kalman_logp_grad(A, B, C, D, q, Z, y, Hdiag=H, tol=1e-8, max_iter=1000, n_filter_hint=18, out=out, solver=SOLVER, n_lead_hint=12 if SOLVER == "gensys" else 0, options=OPTS)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 3, out


out = None
for it in range(4):
    if it == 1:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    out = eng.solve_kalman_logp_grad(A, B, C, D, q, Z, y, Hdiag=H, tol=1e-8, max_iter=1000, n_filter_hint=18, out=out, solver=SOLVER, n_lead_hint=12 if SOLVER == "gensys" else 0, options=OPTS)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
st = out["status"].cpu().numpy() if hasattr(out["status"], "cpu") else np.asarray(out["status"])
bad = np.flatnonzero(st != 0)
print(f"solver {SOLVER}{'' if OPTS is None else ' ' + str(OPTS)}: {nb} draws: {dt*1e3:.2f} ms per logp+gradient batch = {nb/dt:.0f} gradient evals/s; failed {len(bad)}"
      + (f" (draws {bad[:8].tolist()}, status words {st[bad[:8]].tolist()})" if len(bad) else ""))

# The batch holds ONE draw (752: cond(B + C T) = 3e8) whose policy adjoints need the elimination-based fixed point of the second
# pass (adj_stein_fixed_point: 39 sweeps on one wavefront at the very end of the pipeline); the same batch with that draw replaced:
if nb > 752 and nd > 752:
    A2, B2, C2, D2, q2 = (x.clone() for x in (A, B, C, D, q))
    for x in (A2, B2, C2, D2, q2):
        x[752] = x[0]
    dt2, out2 = run(A2, B2, C2, D2, q2)
    print(f"without the nearly singular draw: {dt2*1e3:.2f} ms per batch = {nb/dt2:.0f} gradient evals/s; failed {int((out2['status'] != 0).sum())}")
