"""Container-only helper: execute the reference's solver cores by AST extraction.

gEconpy needs Python >= 3.12 plus pytensor/numba, neither of which exists in the
build container, so the reference cannot be imported.  The solver *function bodies*
however are plain numpy + seven LAPACK helper calls.  This module parses the two
reference files, keeps only the listed ``FunctionDef`` nodes, drops the numba
decorators and executes them in a namespace that binds the seven helper names to
scipy equivalents.  Nothing from the reference is copied into the
repository: the source is read from ``/root/reference`` at run time, and this
module is imported only by ``make_golden.py`` (fixture generation) — never by
tests, bench or product code, and never on the GPU box (where ``/root/reference``
does not exist).
"""
from __future__ import annotations

import ast
import os

import numpy as np
import scipy.linalg as sla
from scipy.linalg import lapack

REF = os.environ.get("GECONPY_REFERENCE", "/root/reference")

_GENSYS_FUNCS = {
    "split_matrix_on_eigen_stability",
    "_thin_svd_and_rank",
    "_matrix_rank",
    "_gensys_core",
    "_gensys_setup",
}
_CR_FUNCS = {"cycle_reduction_numpy", "_cycle_reduction_core"}


def _qz_complex_sort_eig(A, B, sort, overwrite_a, overwrite_b):
    AA, BB, alpha, beta, Q, Z = sla.ordqz(A, B, sort=sort, output="complex")
    return AA, BB, alpha, beta, Q, Z


def _svd_gesdd_full(A, full_matrices=False):
    return sla.svd(A, full_matrices=full_matrices, lapack_driver="gesdd")


def _svd_gesdd_no_uv(A):
    return sla.svd(A, compute_uv=False, lapack_driver="gesdd")


def _lu_factor(M, overwrite):
    getrf = lapack.zgetrf if np.iscomplexobj(M) else lapack.dgetrf
    lu, piv, info = getrf(np.asfortranarray(M))
    return lu, piv.astype(np.int32)


def _getrs(lu, b, piv, trans, overwrite_b):
    getrs = lapack.zgetrs if np.iscomplexobj(lu) else lapack.dgetrs
    x, info = getrs(lu, piv - 1, np.asfortranarray(b), trans=trans)
    return x, info


def _solve_triangular(A, b, trans, lower, unit_diagonal, overwrite_b):
    return sla.solve_triangular(A, b, trans=trans, lower=lower, unit_diagonal=unit_diagonal)


def _solve_gen(A, B, *args):
    return np.linalg.solve(A, B)


def _extract(path, names, namespace):
    with open(path) as fh:
        tree = ast.parse(fh.read())
    keep = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            node.decorator_list = []
            keep.append(node)
    mod = ast.Module(body=keep, type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, path, "exec"), namespace)
    return namespace


def load_reference_solvers():
    """Return a dict of the reference's own solver functions, runnable here."""
    ns = {
        "np": np,
        "_qz_complex_sort_eig": _qz_complex_sort_eig,
        "_svd_gesdd_full": _svd_gesdd_full,
        "_svd_gesdd_no_uv": _svd_gesdd_no_uv,
        "_lu_factor": _lu_factor,
        "_getrs": _getrs,
        "_solve_triangular": _solve_triangular,
        "_solve_gen": _solve_gen,
    }
    _extract(os.path.join(REF, "gEconpy/solvers/gensys.py"), _GENSYS_FUNCS, ns)
    _extract(os.path.join(REF, "gEconpy/solvers/cycle_reduction.py"), _CR_FUNCS, ns)
    return ns


def load_reference_goldens():
    """The reference's golden (A,B,C,D) dict, tests/_resources/expected_matrices.py."""
    path = os.path.join(REF, "tests/_resources/expected_matrices.py")
    ns = {}
    with open(path) as fh:
        exec(compile(fh.read(), path, "exec"), ns)
    return ns["expected_linearization_result"]
