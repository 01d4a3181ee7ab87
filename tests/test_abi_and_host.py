"""CPU: the C-ABI library loads and exports every symbol include/dsge_hip.h declares; host-side
argument handling; the product path refuses to run without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

from geconpy_amd import _lib, batched
from geconpy_amd import workloads as wl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "dsge_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dsge_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        from geconpy_amd.build import build_library

        build_library()
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in dsge_hip.h but not exported"
        assert name in _lib.PROTOTYPES, f"{name} has no ctypes prototype"
    assert sorted(_lib.PROTOTYPES) == declared
    assert lib.dsge_abi_version() == _lib.ABI_VERSION


def test_header_constants_match_python():
    text = open(os.path.join(ROOT, "include", "dsge_hip.h")).read()
    consts = dict(re.findall(r"#define\s+(DSGE_[A-Z_0-9]+)\s+(-?\d+)", text))
    assert int(consts["DSGE_ABI_VERSION"]) == _lib.ABI_VERSION
    assert int(consts["DSGE_MAX_N"]) == _lib.MAX_N
    assert int(consts["DSGE_MAX_N_CR"]) == _lib.MAX_N_CR
    assert int(consts["DSGE_MAX_N_BIG"]) == _lib.MAX_N_BIG
    assert int(consts["DSGE_MAX_P"]) == _lib.MAX_P
    for name in ("ST_NOT_CONVERGED", "ST_NAN", "ST_LYAP_FAIL", "ST_FILTER_NONFINITE", "Q_DIAG_SHARED", "Q_DIAG_BATCHED",
                 "Q_FULL_SHARED", "Q_FULL_BATCHED", "SOLVER_CYCLE_REDUCTION", "SOLVER_GENSYS",
                 "SOLVER_BACKWARD_DIRECT"):
        assert int(consts["DSGE_" + name]) == getattr(_lib, name)


@pytest.mark.skipif(_lib.device_count() > 0, reason="a GPU is present")
def test_no_cpu_fallback():
    """Without a gfx950 device every compute entry point must fail loudly."""
    A = np.zeros((2, 4, 4))
    with pytest.raises(_lib.DsgeHipError, match="no HIP device|no gfx950"):
        batched.cycle_reduction_batched(A, A, A)
    with pytest.raises(_lib.DsgeHipError):
        batched.solve_kalman_logp_batched(A, A, A, np.zeros((2, 4, 1)), np.ones(1), np.eye(1, 4), np.zeros((3, 1)))
    from geconpy_amd.engine import LogpEngine

    with pytest.raises(_lib.DsgeHipError):
        LogpEngine(0)


def test_malformed_calls_are_rejected_before_touching_the_gpu():
    lib = _lib.load()
    # n out of range / null pointers / bad k: DSGE_ERR_INVALID (1), never a crash
    assert lib.dsge_cycle_reduction_batched_host(None, None, None, 1, 4, 10, 1e-8, None, None, None) == 1
    a = np.zeros((1, 4, 4))
    st = np.zeros(1, dtype=np.int32)
    p = lambda x: x.ctypes.data  # noqa: E731
    assert lib.dsge_cycle_reduction_batched_host(p(a), p(a), p(a), 1, 0, 10, 1e-8, p(a), p(st), None) == 1
    assert lib.dsge_cycle_reduction_batched_host(p(a), p(a), p(a), 1, _lib.MAX_N_BIG + 1, 10, 1e-8, p(a), p(st), None) == 1
    assert lib.dsge_cycle_reduction_batched_host(p(a), p(a), p(a), -1, 4, 10, 1e-8, p(a), p(st), None) == 1
    assert b"range" in lib.dsge_last_error() or b"batch" in lib.dsge_last_error()
    assert lib.dsge_selection_batched_host(p(a), p(a), p(a), p(a), p(a), 1, 4, 5, p(a), None) == 1  # k > n


def test_shape_checks_in_numpy_front_end():
    A = np.zeros((2, 4, 4))
    with pytest.raises(ValueError):
        batched.cycle_reduction_batched(A, A[:, :3], A)
    with pytest.raises(ValueError):
        batched.kalman_logp_batched(A, np.zeros((2, 4, 1)), np.ones(1), np.zeros((2, 3)), np.zeros((5, 1)))
    with pytest.raises(ValueError):  # ambiguous Q when batch == k
        batched._q_mode(np.ones((2, 2)), 2, 2)
    assert batched._q_mode(np.ones((5, 2)), 5, 2)[1] == _lib.Q_DIAG_BATCHED
    assert batched._q_mode(np.ones((2, 2)), 5, 2)[1] == _lib.Q_FULL_SHARED


def test_filter_type_other_than_standard_is_refused_loudly():
    """build.py:577 / statespace.py:69 hand ``filter_type`` to the Kalman filter: the device computes the "standard" one, and a
    request for another variant must not silently get it (no GPU needed: the check runs before anything is staged)."""
    from geconpy_amd import pytensor_ops

    A = np.zeros((1, 3, 3))
    for ft in ("univariate", "steady_state", "single", "cholesky"):
        with pytest.raises(NotImplementedError):
            batched.solve_kalman_logp_batched(A, A, A, np.zeros((1, 3, 1)), np.ones(1), np.zeros((1, 3)), np.zeros((4, 1)), filter_type=ft)
        with pytest.raises(NotImplementedError):
            batched.kalman_logp_batched(A, np.zeros((1, 3, 1)), np.ones(1), np.zeros((1, 3)), np.zeros((4, 1)), filter_type=ft)
        with pytest.raises(NotImplementedError):
            pytensor_ops.HipSolveKalmanLogp(filter_type=ft)
    with pytest.raises(ValueError):
        batched.check_filter_type("kalman")
    assert pytensor_ops.HipSolveKalmanLogp().filter_type == "standard"


def test_structure_hints():
    b = wl.sw_shaped_batch(3)
    om = wl.sw_shaped_observation_model()
    assert batched.state_hint(b["A"]) == 18
    assert batched.state_hint(b["T_star"]) == 18
    assert batched.selector_hint(om["Z"]) == 1
    Z2 = om["Z"].copy()
    Z2[0, 5] = 0.5
    assert batched.selector_hint(Z2) == 0
    Z3 = om["Z"].copy()
    Z3[1] = Z3[0]
    assert batched.selector_hint(Z3) == 0  # two rows on the same column


def test_shard_bounds_cover_the_batch_exactly():
    for batch in (1, 7, 64, 65536, 65537):
        for world in (1, 2, 3, 8):
            b = [wl.shard_bounds(batch, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == batch
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))


def test_draw_seeding_is_shard_independent():
    """Draw i is the same system whichever shard generates it (bit-exact draw indexing)."""
    full = wl.sw_shaped_batch(6)
    part = wl.sw_shaped_batch(3, first_draw=3)
    for x in "ABCD":
        assert np.array_equal(full[x][3:], part[x])
    assert np.array_equal(full["sigma"][3:], part["sigma"])


def test_options_struct_defaults_and_scope():
    """dsge_options (per-call switches): init fills the compiled-in defaults, unknown fields are refused, the
    push/pop scope is balanced per thread -- no GPU needed."""
    import ctypes

    lib = _lib.load()
    o = _lib.make_options()
    assert o.struct_size == ctypes.sizeof(_lib.Options)
    assert (o.cr_compact, o.cr_fused_selection, o.cr_deflation, o.cr_two_waves) == (1, 1, 1, 1)
    assert (o.kalman_order, o.kalman_tiny, o.kalman_block, o.kalman_mfma, o.pipeline_chunks, o.gensys_split) == (1, 1, 0, 2, 0, 1)
    assert o.n_static_hint == -1 and o.kalman_steady_tol == 1e-14
    assert (o.kalman_nt_products, o.cr_fused_deflation, o.cr_four_waves) == (1, 1, 1)
    o2 = _lib.make_options({"kalman_steady_tol": 0.0}, n_static_hint=10)
    assert o2.kalman_steady_tol == 0.0 and o2.n_static_hint == 10 and o2.cr_compact == 1
    with pytest.raises(ValueError):
        _lib.make_options(no_such_switch=1)
    assert lib.dsge_options_pop() != 0  # nothing pushed on this thread
    with _lib.options_scope({"cr_compact": 0}):
        with _lib.options_scope({"kalman_order": 2}):
            pass
    assert lib.dsge_options_pop() != 0
    bad = _lib.make_options()
    bad.kalman_steady_tol = 1.0
    assert lib.dsge_options_push(ctypes.addressof(bad)) != 0
    bad.kalman_steady_tol = 1e-14
    bad.struct_size = 8
    assert lib.dsge_options_push(ctypes.addressof(bad)) != 0
    # ABI 8: no process-wide setters any more -- the defaults are compiled in
    assert not any(name.startswith("dsge_set_") and name != "dsge_set_device" for name in _lib.PROTOTYPES)
    assert not hasattr(lib, "dsge_set_kalman_steady_tol") and not hasattr(lib, "dsge_set_cr_deflation")
    # ... and the conventions of the filter step are fields of the same struct
    assert (o.ll_constant, o.mask_d, o.joseph, o.jitter_F, o.jitter_P) == (0, 0, 1, -1.0, -1.0)
    o3 = _lib.make_options(_lib.filter_conventions(ll_constant="one", jitter_on_P=False, mask_d=True, joseph=False))
    assert (o3.ll_constant, o3.mask_d, o3.joseph, o3.jitter_F, o3.jitter_P) == (2, 1, 0, -1.0, 0.0)
    assert _lib.make_options(ll_constant="observed").ll_constant == 1
    bad = _lib.make_options()
    bad.ll_constant = 3
    assert lib.dsge_options_push(ctypes.addressof(bad)) != 0
    assert lib.dsge_forget_measured_shapes() == 0  # (host-side records only: no GPU needed)
