"""Container-only: eigenvalues of the reference's own ``compute_bk_eigenvalues`` (gEconpy/model/perturbation.py:412-445,
executed by AST extraction exactly like make_golden.py does for the solver cores) on the golden systems, the failure
systems and two SW-shaped draws -> tests/golden/bk_eigenvalues.npz.  Pins oracle.compute_bk_eigenvalues."""
import ast
import os
import sys

import numpy as np
from scipy import linalg

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import _ref_extract  # noqa: E402

from geconpy_amd import workloads as wl  # noqa: E402


def reference_function():
    ns = _ref_extract.gensys_namespace() if hasattr(_ref_extract, "gensys_namespace") else None
    if ns is None:
        ns = {"np": np, "linalg": linalg}
        ns.update({k: getattr(_ref_extract, k) for k in dir(_ref_extract) if k.startswith("_") and callable(getattr(_ref_extract, k))})
        _ref_extract._extract(os.path.join(_ref_extract.REF, "gEconpy/solvers/gensys.py"), {"_gensys_setup"}, ns)
    ns.setdefault("linalg", linalg)
    ns.setdefault("np", np)
    path = os.path.join(_ref_extract.REF, "gEconpy/model/perturbation.py")
    with open(path) as fh:
        tree = ast.parse(fh.read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "compute_bk_eigenvalues"]
    for n in keep:
        n.decorator_list = []
        n.returns = None
        for a in n.args.args:
            a.annotation = None
    mod = ast.Module(body=keep, type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, path, "exec"), ns)
    return ns["compute_bk_eigenvalues"]


if __name__ == "__main__":
    f = reference_function()
    out = {}
    rg = np.load(os.path.join(HERE, "reference_goldens.npz"))
    fg = np.load(os.path.join(HERE, "failure_cases.npz"))
    cases = {k: tuple(rg[f"{k}_{x}"] for x in "ABCD") for k in ("one_block", "rbc_2_block", "full_nk")}
    cases.update({k: tuple(fg[f"{k}_{x}"] for x in "ABCD") for k in ("ok", "nonunique", "noexist")})
    b = wl.sw_shaped_batch(2)
    cases.update({f"sw{i}": tuple(b[x][i] for x in "ABCD") for i in range(2)})
    for name, (A, B, C, D) in cases.items():
        re, im, nf = f(A, B, C, D, 1e-8)
        out[f"{name}_real"], out[f"{name}_imag"], out[f"{name}_n_forward"] = re, im, np.int64(nf)
        print(name, len(re), nf, int((np.hypot(re, im) > 1).sum()))
    out["names"] = np.array(list(cases))
    np.savez(os.path.join(HERE, "bk_eigenvalues.npz"), **out)
