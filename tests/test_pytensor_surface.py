"""The pytensor Op surface without pytensor (it is not installed in this image): the classes of
geconpy_amd.pytensor_ops are built against a ~40-line stand-in for ``pt`` / ``Apply`` and their ``make_node`` /
``infer_shape`` / ``pullback`` contracts are compared with the reference's Ops (gEconpy/solvers/gensys.py:634-683,
cycle_reduction.py:186-219, shared.py:74-75).  The ``perform`` methods run on the device: tests/test_gpu_ops.py."""
import types

import numpy as np
import pytest

import geconpy_amd.pytensor_ops as ops


class _Type:
    def __init__(self, dtype, shape):
        self.dtype, self.shape, self.ndim = dtype, tuple(shape), len(shape)
        self.numpy_dtype = np.dtype(dtype)


class _Var:
    def __init__(self, name, dtype, shape, owner=None):
        self.name, self.type, self.owner = name, _Type(dtype, shape), owner


class _Apply:
    def __init__(self, op, inputs, outputs):
        self.op, self.inputs, self.outputs = op, list(inputs), list(outputs)
        for o in outputs:
            o.owner = self


def _fake_pt():
    def as_tensor(x):
        if isinstance(x, _Var):
            return x
        a = np.asarray(x)
        return _Var("const", str(a.dtype), a.shape)

    return types.SimpleNamespace(
        as_tensor=as_tensor,
        tensor=lambda name, dtype="float64", shape=(): _Var(name, dtype, shape),
        zeros_like=lambda v: _Var("zeros", v.type.dtype, v.type.shape),
        eq=lambda a, b: _Var("eq", "bool", ()),
        constant=lambda v: _Var("const", "int64", ()),
    )


@pytest.fixture()
def surface(monkeypatch):
    if ops.available():
        pytest.skip("pytensor is installed: the real classes are exercised by pytensor itself")
    monkeypatch.setattr(ops, "pt", _fake_pt())
    monkeypatch.setattr(ops, "Apply", _Apply)
    monkeypatch.setattr(ops, "_HAVE_PYTENSOR", True)
    return ops


def _m(n, m=None, dtype="float64", name="x"):
    return _Var(name, dtype, (n, m if m is not None else n))


def test_without_pytensor_the_graph_side_fails_loudly():
    if ops.available():
        pytest.skip("pytensor is installed")
    with pytest.raises(ImportError):
        ops.HipCycleReduction()(np.eye(3), np.eye(3), np.eye(3))


def test_props_and_signatures_match_the_reference():
    assert ops.HipCycleReduction.__props__ == ("max_iter", "tol")  # cycle_reduction.py:187
    assert ops.HipCycleReduction.gufunc_signature == "(n,n),(n,n),(n,n)->(n,n)"  # :188
    assert ops.HipGensys.__props__ == ("tol",)  # gensys.py:635
    assert ops.HipGensys.gufunc_signature == "(n,n),(n,n),(n,n),(n,k)->(n,n),()"  # :636
    cr, gs = ops.HipCycleReduction(), ops.HipGensys()
    assert (cr.max_iter, cr.tol, gs.tol) == (1000, 1e-9, 1e-8)  # defaults :190, gensys.py:638


def test_cycle_reduction_make_node_and_dtype(surface):
    A, B, C = _m(12), _m(12), _m(12)
    node = surface.HipCycleReduction(max_iter=50, tol=1e-6).make_node(A, B, C)
    assert len(node.inputs) == 3 and len(node.outputs) == 1
    assert node.outputs[0].type.shape == (12, 12) and node.outputs[0].type.dtype == "float64"
    assert surface.HipCycleReduction().infer_shape(None, node, [(12, 12)] * 3) == [(12, 12)]
    # linalg_output_dtype (cycle_reduction.py:197): all-float32 inputs give a float32 T, mixed gives float64
    n32 = surface.HipCycleReduction().make_node(*(_m(5, dtype="float32") for _ in range(3)))
    assert n32.outputs[0].type.dtype == "float32"
    mix = surface.HipCycleReduction().make_node(_m(5, dtype="float32"), _m(5), _m(5, dtype="float32"))
    assert mix.outputs[0].type.dtype == "float64"


def test_gensys_make_node_and_pullback(surface):
    A, B, C, D = _m(9), _m(9), _m(9), _m(9, 2)
    op = surface.HipGensys(tol=1e-7)
    node = op.make_node(A, B, C, D)
    T, success = node.outputs
    assert T.type.shape == (9, 9) and success.type.dtype == "bool" and success.type.shape == ()  # gensys.py:646-649
    assert op.infer_shape(None, node, [(9, 9)] * 3 + [(9, 2)]) == [(9, 9), ()]  # :653-655
    T_bar = _m(9, name="T_bar")
    grads = op.pullback([A, B, C, D], [T, success], [T_bar, None])
    assert len(grads) == 4  # A_bar, B_bar, C_bar from the adjoint Op, a zero D_bar (gensys.py:668-676)
    assert all(g.type.shape == (9, 9) for g in grads[:3]) and grads[3].type.shape == (9, 2)
    adj = grads[0].owner
    assert isinstance(adj.op, surface.HipPolicyAdjoint) and adj.inputs == [B, C, T, T_bar]
    # batched inputs: one node, leading draw axis on both outputs
    nb = op.make_node(_Var("A", "float64", (64, 9, 9)), _Var("B", "float64", (64, 9, 9)), _Var("C", "float64", (64, 9, 9)),
                      _Var("D", "float64", (64, 9, 2)))
    assert nb.outputs[0].type.shape == (64, 9, 9) and nb.outputs[1].type.shape == (64,)


def test_selection_is_differentiable_in_all_four_inputs(surface):
    """shared.py:74-75 is ordinary differentiable pytensor, so R must carry a pullback to B, C, D AND T."""
    B, C, D, T = _m(8), _m(8), _m(8, 3), _m(8)
    op = surface.HipSelection()
    node = op.make_node(B, C, D, T)
    R = node.outputs[0]
    assert R.type.shape == (8, 3) and op.infer_shape(None, node, [(8, 8), (8, 8), (8, 3), (8, 8)]) == [(8, 3)]
    R_bar = _m(8, 3, name="R_bar")
    B_bar, C_bar, D_bar, T_bar = op.pullback([B, C, D, T], [R], [R_bar])
    assert [g.type.shape for g in (B_bar, C_bar, D_bar, T_bar)] == [(8, 8), (8, 8), (8, 3), (8, 8)]
    adj = B_bar.owner
    assert isinstance(adj.op, surface.HipSelectionAdjoint) and adj.inputs == [B, C, T, R, R_bar]
    assert adj.op.infer_shape(None, adj, [(8, 8)] * 3 + [(8, 3)] * 2) == [(8, 8), (8, 8), (8, 3), (8, 8)]


def test_pt_level_functions_return_what_the_reference_returns(surface):
    A, B, C, D = _m(10), _m(10), _m(10), _m(10, 4)
    T, R, success = surface.gensys_pt(A, B, C, D, tol=1e-8)  # gensys.py:679-683
    assert T.type.shape == (10, 10) and R.type.shape == (10, 4) and success.type.shape == ()
    assert isinstance(R.owner.op, surface.HipSelection) and R.owner.inputs == [B, C, D, T]
    T2, R2 = surface.cycle_reduction_pt(A, B, C, D, max_iter=50, tol=1e-6)  # cycle_reduction.py:216-219
    assert isinstance(T2.owner.op, surface.HipCycleReduction) and (T2.owner.op.max_iter, T2.owner.op.tol) == (50, 1e-6)
    assert R2.type.shape == (10, 4)
    T3, R3, n_steps = surface.scan_cycle_reduction(A, B, C, D, max_iter=20, tol=1e-7)  # :297-325
    assert T3.type.shape == (10, 10) and R3.type.shape == (10, 4) and n_steps.type.shape == ()
    ok, nf, nu = surface.check_bk_condition_pt(A, B, C, D, lead_var_idx=[1, 4, 7])  # perturbation.py:586-625
    assert ok.type.dtype == "bool" and nu.type.shape == ()


def test_fused_logp_op_shapes(surface):
    nb, n, k, p, T_len = 32, 10, 3, 2, 50
    args = [_Var("A", "float64", (nb, n, n)), _Var("B", "float64", (nb, n, n)), _Var("C", "float64", (nb, n, n)),
            _Var("D", "float64", (nb, n, k)), _Var("q", "float64", (nb, k)), _Var("Z", "float64", (p, n)),
            _Var("y", "float64", (T_len, p)), _Var("d", "float64", (p,)), _Var("H", "float64", (p,))]
    node = surface.HipSolveKalmanLogp(solver="gensys").make_node(*args)
    assert [o.type.shape for o in node.outputs] == [(nb,), (nb,)] and node.outputs[1].type.dtype == "int32"
    g = surface.HipSolveKalmanLogpGrad().make_node(*args)
    assert [o.type.shape for o in g.outputs] == [(nb, n, n)] * 3 + [(nb, n, k), (nb, k), (nb, p), (nb, p)]


def test_numba_wrappers_are_plain_python_when_handed_an_identity_decorator():
    """make_numba_*: with ``njit = identity`` and a fake C function the wrapper logic itself is checked (argument
    order of the C ABI, NaN / success=False on a non-zero return code)."""
    calls = []

    class _P:  # stands for arr.ctypes
        pass

    def fake_cr(a, b, c, batch, n, max_iter, tol, t, status, n_iter):
        calls.append((batch, n, max_iter, tol))
        return 2

    f = ops.make_numba_cycle_reduction(lambda fn: fn, fake_cr, 77, 1e-5, np.float32)
    T = f(np.eye(3), np.eye(3), np.eye(3))
    assert calls == [(1, 3, 77, 1e-5)] and T.dtype == np.float32 and np.isnan(T).all()

    def fake_gs(a, b, c, d, batch, n, k, tol, nl, t, r, eu, status):
        calls.append((batch, n, k, tol, nl, type(d).__name__, type(r).__name__))
        return 0

    g = ops.make_numba_gensys(lambda fn: fn, fake_gs, 1e-8)
    T, success = g(np.eye(4), np.eye(4), np.eye(4), np.ones((4, 2)))
    # D and R are real buffers (numba may refuse to type an integer 0 as a pointer argument)
    assert calls[-1] == (1, 4, 2, 1e-8, 0, "_ctypes", "_ctypes") and T.shape == (4, 4) and not success  # eu stayed [0,0,0]


def test_gensys_output_dtype_follows_floatx(surface, monkeypatch):
    """GensysWrapper.make_node declares ``pt.tensor("T", shape=...)`` without a dtype, i.e. config.floatX (gensys.py:646);
    HipGensys does the same (float64 when pytensor is absent) and perform() casts the float64 result to the declared dtype."""
    op = surface.HipGensys()
    node = op.make_node(_m(5), _m(5), _m(5), _m(5, 2))
    assert node.outputs[0].type.dtype == "float64" and node.outputs[1].type.dtype == "bool"
    monkeypatch.setattr(surface, "_floatx", lambda: "float32")
    node32 = op.make_node(_m(5), _m(5), _m(5), _m(5, 2))
    assert node32.outputs[0].type.dtype == "float32"


def test_live_check_tool_self_skips_without_pytensor():
    """tools/pytensor_live_check.py is the one-command proof of the Op layer under a REAL pytensor: where pytensor is
    absent -- here -- it must say so and exit 77 without touching anything; where it exists, every step it can run must pass."""
    import os
    import subprocess
    import sys

    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "pytensor_live_check.py")
    res = subprocess.run([sys.executable, tool, "--cpu-graph-only"], capture_output=True, text=True, timeout=600, check=False)
    assert res.returncode in (0, 77), res.stdout + res.stderr
    if res.returncode == 77:
        assert "not installed" in res.stdout
