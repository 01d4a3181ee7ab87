"""Pin the Kalman-filter conventions against the REAL third-party filter -- run on any machine that has pymc_extras
(>= 0.12, the version gEconpy requires: pyproject.toml:43-45) and pytensor; neither exists in the build container, so this
script has never run here ("parity unpinned"):

    python tests/golden/make_pymc_extras_golden.py          # writes tests/golden/pymc_extras_kalman.npz
    python -m pytest tests/test_oracle_kalman.py -k pymc_extras_pin

It evaluates the graph gEconpy builds (``PyMCStateSpace.build_statespace_graph`` -> ``StandardFilter``; call sites
gEconpy/model/statespace.py:1143-1157 with ``missing_fill_value`` and ``cov_jitter = JITTER_DEFAULT``) on the matrices of
SW-shaped draws 0 and 752 and of the RBC config -- per-step log-likelihoods, filtered / predicted states and covariances --
for four data sets each: complete; scattered NaN entries; whole missing periods + ``missing_fill_value`` markers; and with an
observation intercept ``d != 0`` under missing entries (is ``d`` masked?).  Each with the default ``cov_jitter`` and with 0.
Everything it stores is NUMBERS (inputs and the filter's outputs); no third-party source is read or copied.

The filter is driven through ``pymc_extras.statespace.filters.StandardFilter().build_graph(...)`` -- the method
``build_statespace_graph`` calls -- so no ``PyMCStateSpace`` subclass, priors or PyMC model are needed.  The signature is
looked up at run time (``inspect``) because it changed between releases (``missing_fill_value`` / ``cov_jitter`` keywords);
exit code 2 = pymc_extras / pytensor not importable (nothing written), 3 = the installed API is not the one described here
(the message says what was found: adapt ``run_filter``).
"""
from __future__ import annotations

import inspect
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def stationary_cov(T, RQR):
    import scipy.linalg as sla

    return sla.solve_discrete_lyapunov(T, RQR)


def cases():
    """(name, dict(T, R, Q, Z, H, d, y)) -- built from this repo's workload generators and its device-independent oracle for T, R."""
    import oracle
    from geconpy_amd import workloads as wl

    out = []
    om = wl.sw_shaped_observation_model()
    for draw in (0, 752):
        b = wl.sw_shaped_batch(1, first_draw=draw)
        res = oracle.solve_kalman_logp(b["A"][0], b["B"][0], b["C"][0], b["D"][0], np.diag(b["sigma"][0] ** 2), om["Z"], om["y"],
                                       H=np.diag(om["Hdiag"]), tol=1e-12, max_iter=1000)
        out.append((f"sw{draw}", dict(T=res["T"], R=res["R"], Q=np.diag(b["sigma"][0] ** 2), Z=om["Z"], H=np.diag(om["Hdiag"]),
                                      y=om["y"][:60].copy())))
    rb, rom = wl.rbc_batch(1)
    res = oracle.solve_kalman_logp(rb["A"][0], rb["B"][0], rb["C"][0], rb["D"][0], np.diag(rb["sigma"][0] ** 2), rom["Z"], rom["y"],
                                   H=np.diag(rom["Hdiag"]), tol=1e-12, max_iter=1000)
    out.append(("rbc", dict(T=res["T"], R=res["R"], Q=np.diag(rb["sigma"][0] ** 2), Z=rom["Z"], H=np.diag(rom["Hdiag"]),
                            y=rom["y"][:60].copy())))
    return out


def data_variants(y, fill):
    rng = np.random.default_rng(20261003)
    T_len, p = y.shape
    v = {"complete": y.copy()}
    a = y.copy()
    a[rng.random(a.shape) < 0.15] = np.nan
    v["scattered_nan"] = a
    b = y.copy()
    b[5:8, :] = np.nan  # whole periods missing: ll_t = 0 there?
    b[20, 0] = fill     # the missing_fill_value marker instead of NaN
    if p > 1:
        b[33, 1:] = fill
    v["periods_and_fill"] = b
    return v


def run_filter(m, y, jitter, fill, d):
    """-> dict of numpy outputs of the installed StandardFilter."""
    import pytensor
    import pytensor.tensor as pt
    from pymc_extras.statespace.filters import StandardFilter

    kf = StandardFilter()
    sig = inspect.signature(kf.build_graph)
    names = list(sig.parameters)
    need = ["data", "a0", "P0", "c", "d", "T", "Z", "R", "H", "Q"]
    if names[: len(need)] != need:
        print("unexpected StandardFilter.build_graph signature:", sig, file=sys.stderr)
        raise SystemExit(3)
    k_states = m["T"].shape[0]
    P0 = stationary_cov(m["T"], m["R"] @ m["Q"] @ m["R"].T)
    yy = y.copy()
    # gEconpy hands the data to PyMCStateSpace, which replaces NaN by missing_fill_value before the filter sees it
    # (statespace.py:1143): do the same so that both markers take the filter's own missing-data path
    yy[np.isnan(yy)] = fill
    args = [pt.as_tensor_variable(np.asarray(x, dtype="float64")) for x in
            (yy, np.zeros(k_states), P0, np.zeros(k_states), d, m["T"], m["Z"], m["R"], m["H"], m["Q"])]
    kw = {}
    if "missing_fill_value" in names:
        kw["missing_fill_value"] = fill
    if "cov_jitter" in names:
        kw["cov_jitter"] = jitter
    elif jitter != 1e-8:
        print("this StandardFilter.build_graph has no cov_jitter keyword: only the default jitter can be pinned", file=sys.stderr)
        return None
    outs = kf.build_graph(*args, **kw)
    vals = pytensor.function([], list(outs))()
    # documented order: filtered_states, predicted_states, observed_states, filtered_covs, predicted_covs, observed_covs, ll_obs
    res = {f"out{i}": np.asarray(v) for i, v in enumerate(vals)}
    ll = [v for v in vals if np.asarray(v).ndim == 1 and np.asarray(v).shape[0] == y.shape[0]]
    if len(ll) != 1:
        print("could not identify the per-observation log-likelihood among the outputs:", [np.asarray(v).shape for v in vals],
              file=sys.stderr)
        raise SystemExit(3)
    res["ll"] = np.asarray(ll[0])
    return res


def main():
    try:
        import pymc_extras
        import pytensor
    except ImportError as exc:
        print(f"pymc_extras / pytensor not importable ({exc}): nothing written -- the filter conventions stay UNPINNED", file=sys.stderr)
        return 2
    fill = -9999.0
    out = {"versions": np.array([f"pymc_extras {pymc_extras.__version__}", f"pytensor {pytensor.__version__}"])}
    for name, m in cases():
        p = m["Z"].shape[0]
        for key in ("T", "R", "Q", "Z", "H"):
            out[f"{name}_{key}"] = m[key]
        for vname, y in data_variants(m["y"], fill).items():
            for dname, d in (("d0", np.zeros(p)), ("d1", 0.01 * (1.0 + np.arange(p)))):
                if dname == "d1" and vname == "complete":
                    continue
                for jname, jitter in (("jit", 1e-8), ("nojit", 0.0)):
                    r = run_filter(m, y, jitter, fill, d)
                    if r is None:
                        continue
                    tag = f"{name}_{vname}_{dname}_{jname}"
                    out[f"{tag}_y"] = y
                    out[f"{tag}_d"] = d
                    out[f"{tag}_jitter"] = np.array(jitter)
                    for k, v in r.items():
                        out[f"{tag}_{k}"] = v
                    print(tag, "logp", float(r["ll"].sum()))
    np.savez_compressed(os.path.join(HERE, "pymc_extras_kalman.npz"), **out)
    print("wrote", os.path.join(HERE, "pymc_extras_kalman.npz"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
