"""Per-thread kernel-variant switches for the measurement scripts: ``set_option(name, value)`` keeps ONE pushed dsge_options
record on the calling thread that carries every override made so far (ABI 8 has no process-wide setters)."""
import ctypes

from geconpy_amd import _lib

_overrides = {}
_pushed = [False]


def set_option(name, value):
    lib = _lib.load()
    _overrides[name] = value
    if _pushed[0]:
        _lib.check(lib.dsge_options_pop())
        _pushed[0] = False
    defaults = _lib.make_options()
    for key in [k for k, v in _overrides.items() if getattr(defaults, k) == v]:
        del _overrides[key]
    if _overrides:
        rec = _lib.make_options(dict(_overrides))
        _lib.check(lib.dsge_options_push(ctypes.addressof(rec)))
        _pushed[0] = True
    if name == "cr_deflation":
        _lib.check(lib.dsge_forget_measured_shapes())
