"""Generate tests/golden/gensys_forward.npz (run ONLY in the build container):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_gensys_forward_golden.py

The forward-solution part of gensys' 9-tuple -- f_mat, f_wt, y_wt, loose (gEconpy/solvers/gensys.py:367-393) -- produced by
the REFERENCE's own ``_gensys_core`` body (executed through ``_ref_extract.py``) on

  * the three golden Jacobian sets of the reference (tests/_resources/expected_matrices.py) through ``_gensys_setup``;
  * the non-unique SW-shaped failure case of failure_cases.npz (eu = [1, 0, 6]: the only class with loose != 0);
  * two arbitrary pencils: the one_block pencil and the non-unique one under a random equivalence transformation, with a Pi
    whose columns are neither orthogonal nor normalised (and, for the first, a non-zero c and an extra shock column).

f_mat, f_wt and y_wt depend on the unitary basis LAPACK picks for the unstable block, so the fixture ALSO stores what does
not: the Markov parameters y_wt f_mat^s f_wt (s = 0..3), the spectrum of f_mat, and the projector y_wt pinv(y_wt).
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from _ref_extract import load_reference_goldens, load_reference_solvers  # noqa: E402

REF = load_reference_solvers()


def invariants(fmat, fwt, ywt):
    mk = []
    P = fwt
    for _s in range(4):
        mk.append(ywt @ P)
        P = fmat @ P
    ev = np.linalg.eigvals(fmat) if fmat.size else np.zeros(0, dtype=complex)
    ev = ev[np.lexsort((ev.imag, ev.real))]
    proj = ywt @ np.linalg.pinv(ywt) if ywt.size else np.zeros((ywt.shape[0],) * 2, dtype=complex)
    return np.stack(mk), ev, proj


def run(key, g0, g1, c, psi, pi, tol, out):
    G1, Cc, impact, fmat, fwt, ywt, gev, eu, loose = REF["_gensys_core"](g0, g1, c, psi, pi, tol)
    mk, ev, proj = invariants(fmat, fwt, ywt)
    for name, v in (("g0", g0), ("g1", g1), ("c", c), ("psi", psi), ("pi", pi), ("tol", np.array(tol)),
                    ("ref_G1", G1), ("ref_C", Cc), ("ref_impact", impact), ("ref_fmat", fmat), ("ref_fwt", fwt),
                    ("ref_ywt", ywt), ("ref_gev", gev), ("ref_eu", np.asarray(eu, dtype=np.int64)), ("ref_loose", loose),
                    ("ref_markov", mk), ("ref_fmat_eig", ev), ("ref_ywt_proj", proj)):
        out[f"{key}_{name}"] = np.asarray(v)
    print(key, "N", g0.shape[0], "eu", list(eu), "nu", fmat.shape[0], "max|loose|", float(np.abs(loose).max()) if loose.size else 0.0)


def main():
    out = {}
    g = load_reference_goldens()
    for name, key in (("one_block_1_ss.gcn", "one_block"), ("rbc_2_block_ss.gcn", "rbc_2_block"), ("full_nk.gcn", "full_nk")):
        A, B, C, D = (np.ascontiguousarray(g[name][x], dtype=np.float64) for x in "ABCD")
        run(key, *REF["_gensys_setup"](A, B, C, D, 1e-8), 1e-8, out)
    fc = np.load(os.path.join(HERE, "failure_cases.npz"))
    A, B, C, D = (fc[f"nonunique_{x}"] for x in "ABCD")
    run("nonunique", *REF["_gensys_setup"](A, B, C, D, 1e-8), 1e-8, out)
    # an arbitrary pencil
    A, B, C, D = (np.ascontiguousarray(g["one_block_1_ss.gcn"][x], dtype=np.float64) for x in "ABCD")
    g0, g1, c, psi, pi = REF["_gensys_setup"](A, B, C, D, 1e-8)
    rng = np.random.default_rng(20261003)
    N = g0.shape[0]
    P = np.eye(N) + 0.3 * rng.standard_normal((N, N))
    S = np.eye(N) + 0.3 * rng.standard_normal((N, N))
    W = np.eye(pi.shape[1]) + 0.5 * rng.standard_normal((pi.shape[1],) * 2)
    W *= rng.uniform(0.5, 3.0, pi.shape[1])[None, :]
    run("arbitrary", P @ g0 @ S, P @ g1 @ S, P @ rng.standard_normal((N, 1)), P @ np.hstack([psi, rng.standard_normal((N, 1))]),
        P @ pi @ W, 1e-8, out)
    # ... and a non-unique one (loose != 0 with a Pi that is not orthonormal)
    g0, g1, c, psi, pi = REF["_gensys_setup"](*(fc[f"nonunique_{x}"] for x in "ABCD"), 1e-8)
    N = g0.shape[0]
    P = np.eye(N) + 0.1 * rng.standard_normal((N, N))
    S = np.eye(N) + 0.1 * rng.standard_normal((N, N))
    W = np.eye(pi.shape[1]) + 0.3 * rng.standard_normal((pi.shape[1],) * 2)
    W *= rng.uniform(0.5, 3.0, pi.shape[1])[None, :]
    run("arbitrary_nonunique", P @ g0 @ S, P @ g1 @ S, np.zeros((N, 1)), P @ psi, P @ pi @ W, 1e-8, out)
    np.savez_compressed(os.path.join(HERE, "gensys_forward.npz"), **out)


if __name__ == "__main__":
    main()
