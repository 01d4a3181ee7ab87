import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, multiprocessing as mp
from geconpy_amd import workloads as wl
nb = 1024
def work(i):
    import oracle
    b = wl.sw_shaped_batch(1, first_draw=i); om = wl.sw_shaped_observation_model()
    r = oracle.solve_kalman_logp(b["A"][0], b["B"][0], b["C"][0], b["D"][0], np.diag(b["sigma"][0] ** 2), om["Z"], om["y"], H=np.diag(om["Hdiag"]), tol=1e-8, max_iter=1000)
    Tg, ok, eu = oracle.gensys_T_success(b["A"][0], b["B"][0], b["C"][0], b["D"][0])
    return r["logp"], np.abs(r["T"] - b["T_star"][0]).max(), r["n_iter"], np.max(np.abs(np.linalg.eigvals(b["T_star"][0]))), np.abs(Tg - b["T_star"][0]).max()
if __name__ == "__main__":
    os.environ["OMP_NUM_THREADS"] = "1"
    with mp.get_context("spawn").Pool(64) as pool:
        res = pool.map(work, range(nb))
    ref = np.array([r[0] for r in res]); terr = np.array([r[1] for r in res]); it = np.array([r[2] for r in res]); rho = np.array([r[3] for r in res]); tgerr = np.array([r[4] for r in res])
    from geconpy_amd import batched
    b = wl.sw_shaped_batch(nb); om = wl.sw_shaped_observation_model()
    out = batched.solve_kalman_logp_batched(b["A"], b["B"], b["C"], b["D"], b["sigma"] ** 2, om["Z"], om["y"], Hdiag=om["Hdiag"], tol=1e-8, max_iter=1000, return_policy=True)
    rel = np.abs(out["logp"] - ref) / np.abs(ref)
    w = np.argsort(rel)[-5:]
    print("max rel", rel.max())
    for i in w:
        print(i, "rel %.2e" % rel[i], "oracle CR T err vs T* %.2e" % terr[i], "gpu T err vs T* %.2e" % np.abs(out["T"][i] - b["T_star"][i]).max(), "oracle gensys T err %.2e" % tgerr[i], "iters", it[i], out["n_iter"][i], "rho %.3f" % rho[i])
