"""Container-only generator: 40-digit (mpmath) evaluation of the Kalman log-likelihood on the cases of
tests/golden/statsmodels_kalman.npz -- jitter = 0, complete data, stationary P0 by doubling.  An arbitrary-precision
run of the textbook recursion is the ground truth both the oracle (numpy float64) and statsmodels are compared with;
it also covers the pymc_extras convention with jitter 1e-8 (``*_loglike_mp_jitter``), i.e. the exact arithmetic value
of what oracle/statespace.py restates.

    python tests/golden/make_mp_golden.py            (pure Python; the n = 40 cases take several minutes)
"""
import os
import sys

import numpy as np
from mpmath import det, log, matrix, mp, pi

mp.dps = 40
HERE = os.path.dirname(os.path.abspath(__file__))


def loglike_mp(c, jitter):
    M = lambda a: matrix(np.atleast_2d(a).tolist())  # noqa: E731
    T, R, Q, Z, H = M(c["T"]), M(c["R"]), M(c["Q"]), M(c["Z"]), M(c["H"])
    m, p = T.rows, Z.rows
    d = matrix(c["d"].tolist())
    G = R * Q * R.T
    G = (G + G.T) / 2
    P, A = G.copy(), T.copy()
    for _ in range(80):  # P0 = sum_k T^k G T'^k by doubling
        P = P + A * P * A.T
        A = A * A
    jit = mp.mpf(jitter)
    Ip, Im = mp.eye(p), mp.eye(m)
    a = matrix(m, 1)
    ll = mp.mpf(0)
    for t in range(c["y"].shape[0]):
        y = matrix(c["y"][t].tolist())
        v = y - (d + Z * a)
        PZt = P * Z.T
        F = Z * PZt + H + jit * Ip
        Fi = F ** -1
        K = PZt * Fi
        ll += -mp.mpf(1) / 2 * (p * log(2 * pi) + log(det(F)) + (v.T * Fi * v)[0])
        IKZ = Im - K * Z
        Pf = IKZ * P * IKZ.T + K * H * K.T + jit * Im   # Joseph form, as restated in oracle/statespace.py
        a = T * (a + K * v)
        P = T * Pf * T.T + G
        P = (P + P.T) / 2
    return ll


if __name__ == "__main__":
    g = np.load(os.path.join(HERE, "statsmodels_kalman.npz"))
    names = sys.argv[1:] or [str(n) for n in g["names"]]
    out_path = os.path.join(HERE, "mp_kalman.npz")
    out = dict(np.load(out_path)) if os.path.exists(out_path) else {}
    for name in names:
        c = {key: g[f"{name}_{key}"] for key in ("T", "R", "Q", "Z", "H", "d", "y")}
        for label, jit in (("", 0.0), ("_jitter", 1e-8)):
            val = loglike_mp(c, jit)
            out[f"{name}_loglike_mp{label}"] = np.array(float(val))
            out[f"{name}_loglike_mp{label}_str"] = np.array(mp.nstr(val, 30))
            print(name, label or "(jitter 0)", mp.nstr(val, 25), flush=True)
        np.savez_compressed(out_path, **out)
