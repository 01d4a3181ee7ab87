"""ctypes binding of libdsge_hip.so (C ABI declared in include/dsge_hip.h).

The library is the product path; there is NO fallback.  ``load()`` raises when the shared
object is missing and every compute call raises ``DsgeHipError`` when no gfx950 device is
present (the library itself refuses with DSGE_ERR_HIP).
"""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libdsge_hip.so")

ABI_VERSION = 9
ERR_INVALID, ERR_HIP, ERR_TOO_LARGE = 1, 2, 3
MAX_N = 64
MAX_N_CR = 64
MAX_N_GENSYS = 64
MAX_N_BIG = 96  # cycle reduction, selection and the fused solve + Kalman logp with a cycle-reduction solver (csrc/dsge_big.hpp)
MAX_P = 16

ST_OK = 0
ST_NOT_CONVERGED = 1
ST_NAN = 2
ST_LYAP_FAIL = 4
ST_FILTER_NONFINITE = 8
ST_GENSYS_QZ_FAIL = 16
ST_GENSYS_TOO_BIG = 32
ST_GRAD_UNSUPPORTED = 64

Q_DIAG_SHARED, Q_DIAG_BATCHED, Q_FULL_SHARED, Q_FULL_BATCHED = 0, 1, 2, 3
SOLVER_CYCLE_REDUCTION, SOLVER_GENSYS, SOLVER_BACKWARD_DIRECT, SOLVER_SCAN_CYCLE_REDUCTION = 0, 1, 2, 3
SOLVER_FLAG_ZERO_T_ON_FAILURE = 0x100  # include/dsge_hip.h: DSGE_SOLVER_FLAG_ZERO_T_ON_FAILURE
SOLVER_CODES = {
    "cycle_reduction": SOLVER_CYCLE_REDUCTION,
    "gensys": SOLVER_GENSYS,
    "backward_direct": SOLVER_BACKWARD_DIRECT,
    "scan_cycle_reduction": SOLVER_SCAN_CYCLE_REDUCTION,
}


class DsgeHipError(RuntimeError):
    code = None  # the library's return code (DSGE_ERR_*), when the error came from a call


class DsgeTooLargeError(DsgeHipError):
    """DSGE_ERR_TOO_LARGE: a well-formed call whose problem exceeds the entry point's on-chip capacity."""


class DsgeNoVerdictError(DsgeHipError):
    """``eu = [-3, -3, 0]``: gensys on a model with 65 .. 96 variables (solved by spectral division, csrc/dsge_big.hpp) whose draw
    is NOT certified regular -- not converged, a root within 2e-4 of the unit circle, a scale the reference's absolute tolerances
    would notice -- and there is no ordered QZ at that size to issue the reference's verdict ([1, 0, k], [0, 1, 0], [-2, -2, 0],
    or [1, 1, 0] for a regular draw close to the unit circle).  The reference never returns -3; the batched entry points report
    it per draw (status NOT_CONVERGED | GENSYS_TOO_BIG, T = R = 0), the single-model wrappers raise this."""


EU_NO_VERDICT = -3  # eu[0] = eu[1] = -3: see DsgeNoVerdictError


class GensysForward(C.Structure):
    """``dsge_gensys_forward`` of include/dsge_hip.h (addresses; 0 = not wanted)."""

    _fields_ = [("f_mat", C.c_void_p), ("f_wt", C.c_void_p), ("y_wt", C.c_void_p), ("loose", C.c_void_p),
                ("n_unstable", C.c_void_p), ("pi_raw", C.c_int32)]


_dp = C.c_void_p  # double* / int32* / stream: passed as raw addresses (host or device)
_i = C.c_int
_f = C.c_double

# name -> argtypes; every symbol include/dsge_hip.h declares must appear here
PROTOTYPES = {
    "dsge_abi_version": [],
    "dsge_last_error": [],
    "dsge_device_count": [],
    "dsge_set_device": [_i],
    "dsge_stream_synchronize": [_dp],
    "dsge_cycle_reduction_batched": [_dp, _dp, _dp, _i, _i, _i, _f, _dp, _dp, _dp, _dp],
    "dsge_cycle_reduction_batched_host": [_dp, _dp, _dp, _i, _i, _i, _f, _dp, _dp, _dp],
    "dsge_scan_cycle_reduction_batched": [_dp, _dp, _dp, _i, _i, _i, _f, _dp, _dp, _dp, _dp],
    "dsge_scan_cycle_reduction_batched_host": [_dp, _dp, _dp, _i, _i, _i, _f, _dp, _dp, _dp],
    "dsge_gensys_batched": [_dp, _dp, _dp, _dp, _i, _i, _i, _f, _i, _dp, _dp, _dp, _dp, _dp],
    "dsge_gensys_batched_host": [_dp, _dp, _dp, _dp, _i, _i, _i, _f, _i, _dp, _dp, _dp, _dp],
    "dsge_gensys_pencil_batched": [_dp, _dp, _dp, _dp, _dp, _i, _i, _i, _i, _f, _dp, _dp, _dp, _dp, _dp, _dp, _dp],
    "dsge_gensys_pencil_batched_host": [_dp, _dp, _dp, _dp, _dp, _i, _i, _i, _i, _f, _dp, _dp, _dp, _dp, _dp, _dp],
    "dsge_gensys_pencil_full_batched": [_dp, _dp, _dp, _dp, _dp, _i, _i, _i, _i, _f, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp],
    "dsge_gensys_pencil_full_batched_host": [_dp, _dp, _dp, _dp, _dp, _i, _i, _i, _i, _f, _dp, _dp, _dp, _dp, _dp, _dp, _dp],
    "dsge_bk_eigenvalues_batched": [_dp, _dp, _dp, _i, _i, _f, _dp, _dp, _dp, _dp, _dp, _dp, _dp],
    "dsge_bk_eigenvalues_batched_host": [_dp, _dp, _dp, _i, _i, _f, _dp, _dp, _dp, _dp, _dp, _dp],
    "dsge_debug_cr_phases": [_i, _dp],
    "dsge_debug_kalman_steady_steps": [_dp],
    "dsge_debug_kalman_timeline": [_dp],
    "dsge_debug_kalman_phases": [_i, _dp],
    "dsge_debug_big_phases": [_i, _dp],
    "dsge_debug_gensys_window_phases": [_i, _dp],
    "dsge_debug_gensys_phases": [_dp, _dp, _dp, _i, _i, _f, _i, _dp, _dp, _dp, _dp],
    "dsge_debug_gensys_stage_ms": [_i, _dp],
    "dsge_selection_batched": [_dp, _dp, _dp, _dp, _dp, _i, _i, _i, _dp, _dp, _dp],
    "dsge_selection_batched_host": [_dp, _dp, _dp, _dp, _dp, _i, _i, _i, _dp, _dp],
    "dsge_policy_adjoints_batched": [_dp, _dp, _dp, _dp, _i, _i, _dp, _dp, _dp, _dp, _dp],
    "dsge_policy_adjoints_batched_host": [_dp, _dp, _dp, _dp, _i, _i, _dp, _dp, _dp, _dp],
    "dsge_debug_adjoint_refine": [_i],
    "dsge_debug_second_order_phases": [_i, _dp],
    "dsge_selection_adjoints_batched": [_dp, _dp, _dp, _dp, _dp, _i, _i, _i, _dp, _dp, _dp, _dp, _dp],
    "dsge_selection_adjoints_batched_host": [_dp, _dp, _dp, _dp, _dp, _i, _i, _i, _dp, _dp, _dp, _dp],
    "dsge_policy_norms_batched": [_dp, _dp, _dp, _dp, _dp, _dp, _dp, _i, _i, _i, _dp, _dp, _dp],
    "dsge_policy_norms_batched_host": [_dp, _dp, _dp, _dp, _dp, _dp, _dp, _i, _i, _i, _dp, _dp],
    "dsge_backward_direct_batched": [_dp, _dp, _dp, _i, _i, _i, _dp, _dp, _dp],
    "dsge_backward_direct_batched_host": [_dp, _dp, _dp, _i, _i, _i, _dp, _dp],
    "dsge_lyapunov_batched": [_dp, _dp, _dp, _i, _i, _i, _i, _dp, _dp, _dp, _dp],
    "dsge_lyapunov_batched_host": [_dp, _dp, _dp, _i, _i, _i, _i, _dp, _dp, _dp],
    "dsge_autocorrelation_batched": [_dp, _dp, _dp, _i, _dp, _dp, _i, _i, _i, _i, _i, _i, _i, _dp, _dp, _dp, _dp],
    "dsge_autocorrelation_batched_host": [_dp, _dp, _dp, _i, _dp, _dp, _i, _i, _i, _i, _i, _i, _i, _dp, _dp, _dp],
    "dsge_kalman_logp_batched": [_dp, _dp, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _i, _i, _i, _i, _f, _f, _i, _i, _dp, _dp, _dp],
    "dsge_kalman_logp_batched_host": [_dp, _dp, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _i, _i, _i, _i, _f, _f, _i, _i, _dp, _dp],
    "dsge_kalman_filter_outputs_batched": [_dp, _dp, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _i, _i, _i, _i, _f, _f, _dp, _dp,
                                           _dp, _dp, _dp, _i, _dp, _dp],
    "dsge_kalman_filter_outputs_batched_host": [_dp, _dp, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _i, _i, _i, _i, _f, _f, _dp,
                                                _dp, _dp, _dp, _dp, _i, _dp],
    "dsge_solve_kalman_logp_batched": [_dp, _dp, _dp, _dp, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _i, _i, _i, _i, _i, _f, _i, _f, _f, _i, _i, _i, _dp, _dp, _dp, _dp, _dp, _dp, _dp],
    "dsge_solve_kalman_logp_batched_host": [_dp, _dp, _dp, _dp, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _i, _i, _i, _i, _i, _f, _i, _f, _f, _i, _i, _i, _dp, _dp, _dp, _dp, _dp, _dp],
    "dsge_solve_kalman_logp_augmented_batched": [_dp, _dp, _dp, _dp, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _i, _i, _i, _i,
                                                 _i, _f, _i, _f, _f, _i, _dp, _i, _dp, _dp, _i, _i, _i, _dp, _dp, _dp, _dp,
                                                 _dp, _dp],
    "dsge_solve_kalman_logp_augmented_batched_host": [_dp, _dp, _dp, _dp, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _i, _i,
                                                      _i, _i, _i, _f, _i, _f, _f, _i, _dp, _i, _dp, _dp, _i, _i, _i, _dp,
                                                      _dp, _dp, _dp, _dp],
    "dsge_solve_kalman_logp_grad_batched": [_dp, _dp, _dp, _dp, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _i, _i, _i, _i, _i,
                                            _f, _i, _f, _f, _i, _i, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp],
    "dsge_solve_kalman_logp_grad_batched_host": [_dp, _dp, _dp, _dp, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _i, _i, _i, _i,
                                                 _i, _f, _i, _f, _f, _i, _i, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp],
    "dsge_solve_kalman_logp_grad_dense_z_batched": [_dp, _dp, _dp, _dp, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _i, _i, _i, _i,
                                                    _i, _f, _i, _f, _f, _i, _i, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp,
                                                    _dp],
    "dsge_solve_kalman_logp_grad_dense_z_batched_host": [_dp, _dp, _dp, _dp, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _i, _i,
                                                         _i, _i, _i, _f, _i, _f, _f, _i, _i, _dp, _dp, _dp, _dp, _dp, _dp, _dp,
                                                         _dp, _dp, _dp],
    "dsge_options_init": [_dp],
    "dsge_options_push": [_dp],
    "dsge_options_pop": [],
    "dsge_forget_measured_shapes": [],
    "dsge_solve_kalman_logp_batched_opt": [_dp, _dp, _dp, _dp, _dp, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _i, _i, _i, _i, _i, _f, _i, _f, _f, _i, _i, _i, _dp, _dp, _dp, _dp, _dp, _dp, _dp],
    "dsge_solve_kalman_logp_batched_host_opt": [_dp, _dp, _dp, _dp, _dp, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _i, _i, _i, _i, _i, _f, _i, _f, _f, _i, _i, _i, _dp, _dp, _dp, _dp, _dp, _dp],
    "dsge_solve_kalman_logp_grad_batched_opt": [_dp, _dp, _dp, _dp, _dp, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _i, _i, _i, _i, _i,
                                                _f, _i, _f, _f, _i, _i, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp],
    "dsge_solve_kalman_logp_grad_batched_host_opt": [_dp, _dp, _dp, _dp, _dp, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _i, _i, _i, _i,
                                                     _i, _f, _i, _f, _f, _i, _i, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp],
    "dsge_second_order_logp_batched": [_dp, _dp, _dp, _dp, _dp, _i, _dp, _dp, _i, _dp, _dp, _dp, _dp, _i, _i, _i, _i, _i, _i, _f,
                                       _i, _f, _f, _dp, _i, _dp, _i, _dp, _i, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp],
    "dsge_second_order_logp_batched_host": [_dp, _dp, _dp, _dp, _dp, _i, _dp, _dp, _i, _dp, _dp, _dp, _dp, _i, _i, _i, _i, _i, _i,
                                            _f, _i, _f, _f, _dp, _i, _dp, _i, _dp, _i, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp],
    "dsge_profile_pipeline": [_dp, _dp, _dp, _dp, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _dp, _i, _i, _i, _i, _i, _i, _f, _i, _f, _f, _i, _i, _i, _dp, _dp, _i, _dp, _dp],
}



class Options(C.Structure):
    """``dsge_options`` of include/dsge_hip.h: the kernel-variant switches and the filter conventions of ONE call (per call /
    per host thread; there is no process-wide mutable state since ABI 8)."""

    _fields_ = [
        ("struct_size", C.c_uint32),
        ("cr_compact", C.c_int32),
        ("cr_fused_selection", C.c_int32),
        ("cr_deflation", C.c_int32),
        ("cr_two_waves", C.c_int32),
        ("n_static_hint", C.c_int32),
        ("kalman_order", C.c_int32),
        ("kalman_tiny", C.c_int32),
        ("kalman_block", C.c_int32),
        ("kalman_mfma", C.c_int32),
        ("pipeline_chunks", C.c_int32),
        ("gensys_split", C.c_int32),
        ("kalman_steady_tol", C.c_double),
        ("kalman_nt_products", C.c_int32),
        ("cr_fused_deflation", C.c_int32),
        ("cr_four_waves", C.c_int32),
        ("gensys_real_stage", C.c_int32),
        ("gensys_pairs", C.c_int32),
        ("gensys_shape_cache", C.c_int32),
        ("kalman_narrow", C.c_int32),
        ("gensys_direct_blocks", C.c_int32),
        # ABI 8: conventions of the filter step (third party: pymc_extras)
        ("ll_constant", C.c_int32),
        ("mask_d", C.c_int32),
        ("joseph", C.c_int32),
        ("kalman_head_draws", C.c_int32),
        ("jitter_F", C.c_double),
        ("jitter_P", C.c_double),
        ("gensys_doubling", C.c_int32),
        ("kalman_grad_split", C.c_int32),
        ("reserved_", C.c_int32 * 2),
    ]


LL_CONSTANT = {"p": 0, "observed": 1, "one": 2}  # DSGE_LL_CONST_* (include/dsge_hip.h)


def filter_conventions(ll_constant="p", jitter_on_F=True, jitter_on_P=True, mask_d=False, joseph=True):
    """``dsge_options`` fields for one combination of the third-party conventions of the "standard" filter step, named as
    ``oracle.FilterConventions`` names them (so that the combination tests/test_oracle_kalman.py::test_pymc_extras_pin reports
    for a real pymc_extras install is pasted here verbatim): ``options=filter_conventions(ll_constant="one")``.  A jitter that
    is "on" is the call's ``jitter`` argument (``jitter_F = -1``), "off" is 0."""
    return {"ll_constant": LL_CONSTANT[ll_constant], "jitter_F": -1.0 if jitter_on_F else 0.0,
            "jitter_P": -1.0 if jitter_on_P else 0.0, "mask_d": int(bool(mask_d)), "joseph": int(bool(joseph))}


def make_options(options=None, **fields):
    """A ``dsge_options`` holding the compiled-in defaults with ``fields`` (or the dict ``options``) applied;
    an ``Options`` instance is passed through.  ``ll_constant`` also takes the names "p" / "observed" / "one"."""
    if isinstance(options, Options) and not fields:
        return options
    o = Options()
    check(load().dsge_options_init(C.addressof(o)))
    if isinstance(options, Options):
        C.memmove(C.addressof(o), C.addressof(options), C.sizeof(Options))
    elif options:
        fields = {**options, **fields}
    for name, value in fields.items():
        if name not in {f[0] for f in Options._fields_} or name in ("struct_size", "reserved_"):
            raise ValueError(f"unknown option {name!r}")
        if name == "ll_constant" and isinstance(value, str):
            value = LL_CONSTANT[value]
        setattr(o, name, value)
    return o


def opt_ptr(options):
    """(address or None, keep-alive object) for the ``*_opt`` entry points."""
    if options is None:
        return None, None
    o = make_options(options)
    return C.addressof(o), o


class options_scope:
    """``with options_scope(opts): ...`` -- every library call this THREAD makes inside uses ``opts``
    (dsge_options_push / dsge_options_pop); ``None`` is a no-op."""

    def __init__(self, options):
        self.options = None if options is None else make_options(options)

    def __enter__(self):
        if self.options is not None:
            check(load().dsge_options_push(C.addressof(self.options)))
        return self.options

    def __exit__(self, *exc):
        if self.options is not None:
            check(load().dsge_options_pop())
        return False


_lib = None


def load():
    """Load (once) and return the ctypes handle.  Loading does not touch the GPU."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DsgeHipError(
            f"{LIB_PATH} is missing: build it with `python -m geconpy_amd.build` "
            "(there is no CPU fallback for the HIP engine)"
        )
    # One HIP runtime per process: the PyTorch-ROCm wheel bundles its own libamdhip64 (soname
    # libamdhip64.so.7, but linked by the name "libamdhip64.so").  If this library were loaded
    # first it would pull /opt/rocm's copy and a later `import torch` would load a SECOND
    # runtime, after which one of the two sees no device.  Importing torch first makes our
    # NEEDED libamdhip64.so.7 resolve to the runtime torch already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.argtypes = argtypes
        fn.restype = {"dsge_last_error": C.c_char_p}.get(name, C.c_int)
    if lib.dsge_abi_version() != ABI_VERSION:
        raise DsgeHipError("libdsge_hip ABI version mismatch")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().dsge_last_error()
        cls = DsgeTooLargeError if rc == ERR_TOO_LARGE else DsgeHipError
        err = cls(f"libdsge_hip call failed (code {rc}): {msg.decode() if msg else '?'}")
        err.code = rc
        raise err


def device_count():
    return load().dsge_device_count()
