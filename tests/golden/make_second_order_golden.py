"""Fixture of the second-order leg (BASELINE configs[4]): oracle/second_order.py on 64 DISTINCT SW-shaped draws at the full shape
(n = 40, 18 states, 7 shocks, 7 observables, T = 200; workloads.sw_second_order_batch(64) + sw_shaped_observation_model()).

    python tests/golden/make_second_order_golden.py          # ~3 s of CPU per draw, one process per core

Writes tests/golden/second_order_sw64.npz: logp (64,) and g_ss (64, 40) of every draw.  The inputs
are regenerated from their seeds by the test.  *** The reference has no second-order solver (perturbation.py:97-98 raises): this pins
the DEVICE path to the oracle, it does not pin the oracle to the reference ("parity unpinned" by construction). ***"""
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def _one(args):
    from oracle import second_order as so

    A, B, C, D, idx, val, q, Z, y, Hd = args
    r = so.solve_second_order_logp(A, B, C, D, idx, val, np.diag(q), Z, y, H=np.diag(Hd), tol=1e-8)
    return r["logp"], r["sol"]["g_ss"], r["T"]


if __name__ == "__main__":
    os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = os.environ["MKL_NUM_THREADS"] = "1"
    from geconpy_amd import workloads as wl

    nb = 64
    b = wl.sw_second_order_batch(nb)
    om = wl.sw_shaped_observation_model()
    jobs = [(b["A"][i], b["B"][i], b["C"][i], b["D"][i], b["hess_idx"], b["hess_val"][i], b["sigma"][i] ** 2, om["Z"], om["y"],
             om["Hdiag"]) for i in range(nb)]
    with mp.get_context("spawn").Pool(os.cpu_count() or 1) as pool:
        res = pool.map(_one, jobs, chunksize=1)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "second_order_sw64.npz")
    np.savez_compressed(out, logp=np.array([r[0] for r in res]), g_ss=np.stack([r[1] for r in res]), n_draws=nb)
    print(out, "logp range", min(r[0] for r in res), max(r[0] for r in res))
