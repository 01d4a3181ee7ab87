"""ONE command that proves (or breaks) the drop-in claim of geconpy_amd.pytensor_ops under a REAL pytensor:

    python tools/pytensor_live_check.py            # needs: pytensor (>= 3.0.4 for `pullback`), a gfx950 device, the built library
    python tools/pytensor_live_check.py --cpu-graph-only   # graph construction / shape inference only (no device)

The build container and the GPU boxes of this project have no pytensor / numba / jax, so
tests/test_pytensor_surface.py exercises the Ops against a 40-line stand-in for `pt` / `Apply`.  This script is what that test
cannot be: it builds the Ops under the installed pytensor and runs, in this order (each step prints PASS / FAIL / SKIP),

  1. `pytensor.function` of HipCycleReduction / HipGensys / HipSelection / HipSolveKalmanLogp on the reference's golden systems
     (tests/golden/reference_goldens.npz) and the values against the numpy front-end (`geconpy_amd.batched`);
  2. `infer_shape` through `pytensor.function(..., [out.shape])` without executing `perform`;
  3. `pytensor.grad` of a scalar of T and R w.r.t. A, B, C, D (the Ops' `pullback`, the contract of
     gEconpy/solvers/gensys.py:668-676 and cycle_reduction.py:212-213) against central differences;
  4. `vectorize_graph` of the per-draw Op over a leading draw axis -> must become the batched Op (ONE launch), values equal;
  5. numba mode (`mode="NUMBA"`): the funcify registrations (gEconpy/solvers/gensys.py:686-713, cycle_reduction.py:222-243);
  6. JAX mode (`mode="JAX"`): the `jax_funcify` registration (gEconpy/pytensorf/real_eig.py:100-117 pattern);
  7. if gEconpy itself is importable: its own `cycle_reduction_pt` / `gensys_pt` graphs on the same inputs, values to 1e-9.

Exit code: 0 = every step that could run passed; 1 = a step failed; 77 = pytensor is not installed (nothing checked).
"""
from __future__ import annotations

import argparse
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

RESULTS = []


def step(name):
    def deco(fn):
        def run(*a, **k):
            try:
                msg = fn(*a, **k)
                RESULTS.append((name, "PASS" if msg is None or not str(msg).startswith("SKIP") else "SKIP", msg or ""))
            except ImportError as exc:
                RESULTS.append((name, "SKIP", f"{exc}"))
            except Exception as exc:  # noqa: BLE001
                RESULTS.append((name, "FAIL", f"{type(exc).__name__}: {exc}\n{traceback.format_exc()}"))
            print(f"[{RESULTS[-1][1]}] {name}: {str(RESULTS[-1][2]).splitlines()[0] if RESULTS[-1][2] else ''}", flush=True)

        return run

    return deco


def goldens():
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_goldens.npz"))
    return {k: tuple(np.ascontiguousarray(g[f"{k}_{x}"]) for x in "ABCD") for k in ("one_block", "rbc_2_block", "full_nk")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu-graph-only", action="store_true")
    args = ap.parse_args()
    try:
        import pytensor
        import pytensor.tensor as pt
    except ImportError as exc:
        print(f"pytensor is not installed ({exc}): nothing checked")
        return 77
    from geconpy_amd import batched, pytensor_ops as ops

    assert ops.available(), "geconpy_amd.pytensor_ops did not bind the installed pytensor"
    print("pytensor", pytensor.__version__, "floatX", pytensor.config.floatX)
    G = goldens()
    run_device = not args.cpu_graph_only

    def sym(shape_like, name):
        return pt.tensor(name, shape=tuple(None for _ in shape_like.shape), dtype="float64")

    @step("1. pytensor.function of the per-draw Ops, values vs the numpy front-end")
    def values():
        if not run_device:
            return "SKIP (--cpu-graph-only)"
        for key, (A, B, C, D) in G.items():
            a, b, c, d = (sym(x, n) for x, n in zip((A, B, C, D), "ABCD"))
            T_cr = ops.HipCycleReduction(max_iter=1000, tol=1e-9)(a, b, c)
            f = pytensor.function([a, b, c], T_cr)
            ref = batched.cycle_reduction_batched(A[None], B[None], C[None], max_iter=1000, tol=1e-9)[0][0]
            np.testing.assert_allclose(f(A, B, C), ref, atol=1e-12)
            T_g, R_g, ok = ops.gensys_pt(a, b, c, d, tol=1e-8)
            fg = pytensor.function([a, b, c, d], [T_g, R_g, ok])
            Tg, Rg, okv = fg(A, B, C, D)
            out = batched.gensys_batched(A[None], B[None], C[None], D[None], tol=1e-8)
            assert bool(okv) and out["success"][0]
            np.testing.assert_allclose(Tg, out["T"][0], atol=1e-12)
            np.testing.assert_allclose(Rg, out["R"][0], atol=1e-10)
            np.testing.assert_allclose(Tg, ref, atol=1e-8)  # the reference's cross-solver tolerance

    @step("2. infer_shape without perform")
    def shapes():
        A, B, C, D = G["full_nk"]
        a, b, c, d = (sym(x, n) for x, n in zip((A, B, C, D), "ABCD"))
        T_cr = ops.HipCycleReduction()(a, b, c)
        T_g, R_g, ok = ops.gensys_pt(a, b, c, d)
        f = pytensor.function([a, b, c, d], [T_cr.shape, T_g.shape, R_g.shape], on_unused_input="ignore")
        s1, s2, s3 = f(A, B, C, D)
        assert tuple(s1) == A.shape and tuple(s2) == A.shape and tuple(s3) == D.shape, (s1, s2, s3)
        perf = [n for n in f.maker.fgraph.toposort() if isinstance(n.op, (ops.HipCycleReduction, ops.HipGensys))]
        assert not perf, "shape graph still contains the solver Ops: infer_shape is not used"

    @step("3. pytensor.grad through pullback vs central differences")
    def grads():
        if not run_device:
            return "SKIP (--cpu-graph-only)"
        A, B, C, D = G["rbc_2_block"]
        rng = np.random.default_rng(0)
        W, V = rng.standard_normal(A.shape), rng.standard_normal(D.shape)
        for label, builder in (("cycle_reduction_pt", lambda a, b, c, d: ops.cycle_reduction_pt(a, b, c, d, max_iter=1000, tol=1e-13)),
                               ("gensys_pt", lambda a, b, c, d: ops.gensys_pt(a, b, c, d, tol=1e-8)[:2])):
            a, b, c, d = (sym(x, n) for x, n in zip((A, B, C, D), "ABCD"))
            T, R = builder(a, b, c, d)
            loss = (T * W).sum() + (R * V).sum()
            gfun = pytensor.function([a, b, c, d], pytensor.grad(loss, [a, b, c, d]))
            lfun = pytensor.function([a, b, c, d], loss)
            g = gfun(A, B, C, D)
            for which, X in enumerate((A, B, C, D)):
                nz = np.argwhere(X != 0)
                for (i, j) in nz[rng.choice(len(nz), size=min(4, len(nz)), replace=False)]:
                    h = 1e-6 * max(1.0, abs(X[i, j]))
                    args_p, args_m = [x.copy() for x in (A, B, C, D)], [x.copy() for x in (A, B, C, D)]
                    args_p[which][i, j] += h
                    args_m[which][i, j] -= h
                    fd = (lfun(*args_p) - lfun(*args_m)) / (2 * h)
                    assert abs(fd - g[which][i, j]) <= 1e-5 * max(1.0, abs(fd)), (label, "ABCD"[which], i, j, fd, g[which][i, j])

    @step("4. vectorize_graph turns the per-draw Op into the batched Op")
    def vectorize():
        from pytensor.graph.replace import vectorize_graph

        A, B, C, D = G["one_block"]
        a, b, c = (sym(x, n) for x, n in zip((A, B, C), "ABC"))
        T = ops.HipCycleReduction(max_iter=1000, tol=1e-9)(a, b, c)
        a3, b3, c3 = (pt.tensor(n + "3", shape=(None, None, None), dtype="float64") for n in "ABC")
        T3 = vectorize_graph(T, {a: a3, b: b3, c: c3})
        f = pytensor.function([a3, b3, c3], T3)
        kinds = [type(n.op).__name__ for n in f.maker.fgraph.toposort()]
        assert "HipCycleReductionBatched" in kinds, kinds
        if run_device:
            rng = np.random.default_rng(1)
            As = A[None] * (1 + 1e-3 * rng.standard_normal((5, 1, 1)))
            out = f(As, np.repeat(B[None], 5, 0), np.repeat(C[None], 5, 0))
            ref = batched.cycle_reduction_batched(As, np.repeat(B[None], 5, 0), np.repeat(C[None], 5, 0), max_iter=1000, tol=1e-9)[0]
            np.testing.assert_allclose(out, ref, atol=1e-12)

    @step("5. numba mode (funcify registrations)")
    def numba_mode():
        import numba  # noqa: F401

        if not run_device:
            return "SKIP (--cpu-graph-only)"
        A, B, C, D = G["one_block"]
        a, b, c, d = (sym(x, n) for x, n in zip((A, B, C, D), "ABCD"))
        T = ops.HipCycleReduction(max_iter=1000, tol=1e-9)(a, b, c)
        Tg, ok = ops.HipGensys(tol=1e-8)(a, b, c, d)
        f = pytensor.function([a, b, c, d], [T, Tg, ok], mode="NUMBA")
        t1, t2, okv = f(A, B, C, D)
        ref = batched.cycle_reduction_batched(A[None], B[None], C[None], max_iter=1000, tol=1e-9)[0][0]
        np.testing.assert_allclose(t1, ref, atol=1e-12)
        np.testing.assert_allclose(t2, ref, atol=1e-8)
        assert bool(okv)

    @step("6. JAX mode (jax_funcify registration)")
    def jax_mode():
        import jax  # noqa: F401

        if not run_device:
            return "SKIP (--cpu-graph-only)"
        A, B, C, D = G["one_block"]
        a, b, c = (sym(x, n) for x, n in zip((A, B, C), "ABC"))
        f = pytensor.function([a, b, c], ops.HipCycleReduction(max_iter=1000, tol=1e-9)(a, b, c), mode="JAX")
        ref = batched.cycle_reduction_batched(A[None], B[None], C[None], max_iter=1000, tol=1e-9)[0][0]
        np.testing.assert_allclose(np.asarray(f(A, B, C)), ref, atol=1e-12)

    @step("7. gEconpy's own Ops on the same inputs")
    def against_geconpy():
        from gEconpy.solvers.cycle_reduction import cycle_reduction_pt as ref_cr_pt
        from gEconpy.solvers.gensys import gensys_pt as ref_gensys_pt

        if not run_device:
            return "SKIP (--cpu-graph-only)"
        for key, (A, B, C, D) in G.items():
            a, b, c, d = (sym(x, n) for x, n in zip((A, B, C, D), "ABCD"))
            f_ref = pytensor.function([a, b, c, d], list(ref_cr_pt(a, b, c, d)) + list(ref_gensys_pt(a, b, c, d)))
            f_hip = pytensor.function([a, b, c, d], list(ops.cycle_reduction_pt(a, b, c, d)) + list(ops.gensys_pt(a, b, c, d)))
            for r, h in zip(f_ref(A, B, C, D), f_hip(A, B, C, D)):
                np.testing.assert_allclose(np.asarray(h, dtype=float), np.asarray(r, dtype=float), atol=1e-9)

    for fn in (values, shapes, grads, vectorize, numba_mode, jax_mode, against_geconpy):
        fn()
    failed = [r for r in RESULTS if r[1] == "FAIL"]
    for name, _st, msg in failed:
        print("\n=== FAILED:", name, "\n", msg)
    print(f"{sum(r[1] == 'PASS' for r in RESULTS)} passed, {sum(r[1] == 'SKIP' for r in RESULTS)} skipped, {len(failed)} failed")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
