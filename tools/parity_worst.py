"""GPU box: worst-case relative logp difference between the device and the CPU oracle over the 4096 bench draws;
prints the offenders and stores their draw indices + device logp in gpurun_out/parity_worst.npz (the container then
evaluates them in 40-digit arithmetic, tools/parity_worst_mp.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, multiprocessing as mp
from geconpy_amd import workloads as wl
nb = int(os.environ.get("PARITY_DRAWS", "4096"))
SOLVER = os.environ.get("PARITY_SOLVER", "cycle_reduction")
def work(i):
    import oracle
    b = wl.sw_shaped_batch(1, first_draw=i); om = wl.sw_shaped_observation_model()
    r = oracle.solve_kalman_logp(b["A"][0], b["B"][0], b["C"][0], b["D"][0], np.diag(b["sigma"][0] ** 2), om["Z"], om["y"],
                                 H=np.diag(om["Hdiag"]), tol=1e-8, max_iter=1000, solver=SOLVER)
    return r["logp"]
if __name__ == "__main__":
    os.environ["OMP_NUM_THREADS"] = "1"
    with mp.get_context("spawn").Pool(min(128, os.cpu_count())) as pool:
        ref = np.array(pool.map(work, range(nb), chunksize=8))
    from geconpy_amd import batched
    b = wl.sw_shaped_batch(nb); om = wl.sw_shaped_observation_model()
    out = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], om["y"], Hdiag=om["Hdiag"],
                                            tol=1e-8, max_iter=1000, solver=SOLVER)
    rel = np.abs(out["logp"] - ref) / np.abs(ref)
    w = np.argsort(rel)[-6:][::-1]
    print("max rel", rel.max(), "median", np.median(rel), "n > 1e-12:", int((rel > 1e-12).sum()))
    for i in w:
        print(i, "rel %.3e" % rel[i], "gpu %.15g oracle %.15g" % (out["logp"][i], ref[i]))
    os.makedirs("gpurun_out", exist_ok=True)
    np.savez("gpurun_out/parity_worst.npz", idx=w, gpu=out["logp"][w], oracle=ref[w])
