"""Container-only generator: cross-check vectors for the Kalman log-likelihood from statsmodels.

    /opt/conda/bin/python3.9 tests/golden/make_statsmodels_golden.py        (from the repo root)

The reference's filter lives in pymc_extras (not installed).  The build container does carry an
Anaconda tree with statsmodels 0.12.2 (``/opt/conda/lib/python3.9``), whose ``KalmanFilter`` is an independent,
compiled implementation of the same standard recursion with a stationary initial covariance.  With ``jitter = 0`` and
complete data the two conventions coincide (pymc_extras adds ``jitter * I`` to F and P+, and masks missing rows
instead of dropping them), so statsmodels pins the recursion, the stationary initialisation and the log-likelihood
formula of oracle/statespace.py; the jitter and missing-data conventions stay restated from upstream.

Inputs are generated with numpy only (no repo import: this interpreter has an older numpy/scipy); the SW-shaped and
RBC systems are solved by the caller and passed in through tests/golden/_sm_inputs.npz, written by
``python tests/golden/make_statsmodels_golden.py --inputs`` under the repo's own interpreter.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
INPUTS = os.path.join(HERE, "_sm_inputs.npz")
OUT = os.path.join(HERE, "statsmodels_kalman.npz")


def write_inputs():
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import oracle
    from geconpy_amd import workloads as wl

    cases = {}
    rng = np.random.default_rng(42)
    # (a) small dense system, non-selector Z, intercept
    m, k, p = 5, 2, 2
    T = rng.standard_normal((m, m))
    T *= 0.85 / np.abs(np.linalg.eigvals(T)).max()
    cases["small"] = dict(T=T, R=rng.standard_normal((m, k)), Q=np.diag([0.5, 1.2]), Z=rng.standard_normal((p, m)),
                          H=np.diag([0.1, 0.2]), d=np.array([0.1, -0.2]), y=rng.standard_normal((80, p)))
    # (b) RBC at the calibrated point, observed Y and C
    A, B, C, D = wl.rbc_linearized_jacobians(**wl.RBC_CALIBRATION)
    Tm, ok, _ = oracle.cycle_reduction_core(A, B, C, 200, 1e-13)
    Rm = oracle.compute_selection_matrix(B, C, D, Tm)
    Z = np.zeros((2, 8))
    Z[0, wl.RBC_VARIABLES.index("Y")] = 1.0
    Z[1, wl.RBC_VARIABLES.index("C")] = 1.0
    cases["rbc"] = dict(T=Tm, R=Rm, Q=np.array([[wl.RBC_CALIBRATION["sigma_A"] ** 2]]), Z=Z, H=np.diag([1e-4, 1e-4]),
                        d=np.zeros(2), y=rng.normal(0, 0.05, (100, 2)))
    # (c) two SW-shaped draws (n = 40, 7 observables, T = 200: BASELINE configs[2])
    b = wl.sw_shaped_batch(2)
    om = wl.sw_shaped_observation_model()
    for i in range(2):
        Tm, ok, _ = oracle.cycle_reduction_core(b["A"][i], b["B"][i], b["C"][i], 200, 1e-13)
        Rm = oracle.compute_selection_matrix(b["B"][i], b["C"][i], b["D"][i], Tm)
        cases[f"sw{i}"] = dict(T=Tm, R=Rm, Q=np.diag(b["sigma"][i] ** 2), Z=om["Z"], H=np.diag(om["Hdiag"]), d=np.zeros(7),
                               y=om["y"])
    flat = {f"{name}_{key}": val for name, c in cases.items() for key, val in c.items()}
    np.savez_compressed(INPUTS, names=np.array(list(cases)), **flat)
    print("wrote", INPUTS)


def run_statsmodels():
    class _MachAr:  # numpy >= 1.24 dropped np.MachAr, which statsmodels 0.12 still imports
        def __init__(self):
            fi = np.finfo(float)
            self.eps, self.tiny, self.huge, self.xmin, self.xmax = fi.eps, fi.tiny, fi.max, fi.tiny, fi.max
            self.precision, self.resolution = fi.precision, fi.resolution

    np.MachAr = _MachAr
    import warnings

    warnings.filterwarnings("ignore")
    import statsmodels
    from statsmodels.tsa.statespace.kalman_filter import KalmanFilter

    g = np.load(INPUTS)
    out = {"statsmodels_version": np.array(statsmodels.__version__), "names": g["names"]}
    for name in g["names"]:
        c = {key: g[f"{name}_{key}"] for key in ("T", "R", "Q", "Z", "H", "d", "y")}
        m, k, p = c["T"].shape[0], c["R"].shape[1], c["Z"].shape[0]
        kf = KalmanFilter(k_endog=p, k_states=m, k_posdef=k)
        kf.bind(np.ascontiguousarray(c["y"]))
        kf["design"], kf["obs_intercept"], kf["obs_cov"] = c["Z"], c["d"], c["H"]
        kf["transition"], kf["selection"], kf["state_cov"] = c["T"], c["R"], c["Q"]
        kf.initialize_stationary()
        res = kf.filter()
        for key, val in c.items():
            out[f"{name}_{key}"] = val
        out[f"{name}_loglike"] = np.array(float(res.llf_obs.sum()))
        out[f"{name}_llf_obs"] = np.asarray(res.llf_obs)
        out[f"{name}_P0"] = np.asarray(res.predicted_state_cov[:, :, 0])
        print(name, m, p, float(res.llf_obs.sum()))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    if "--inputs" in sys.argv:
        write_inputs()
    else:
        run_statsmodels()
