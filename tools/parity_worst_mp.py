"""Container: 40-digit evaluation of the whole path (cycle reduction with the reference's stopping rule, R, P0, filter
with the restated pymc_extras conventions) on the draws where device and oracle differ most (tools/parity_worst.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np
from mpmath import mp, matrix, mpf
from make_mp_golden import loglike_mp  # sets mp.dps = 40
from geconpy_amd import workloads as wl

def norm1(M):
    return max(sum(abs(M[i, j]) for i in range(M.rows)) for j in range(M.cols))

def cr_mp(A, B, C, tol, max_iter=1000):
    A0, A1, A2, Ah = A.copy(), B.copy(), C.copy(), B.copy()
    for it in range(max_iter):
        A1i = A1 ** -1
        X0 = A1i * A0; X2 = A1i * A2
        m00, m02, m20, m22 = A0 * X0, A0 * X2, A2 * X0, A2 * X2
        A1 = A1 - m02 - m20; Ah = Ah - m20; A0 = -m00; A2 = -m22
        if norm1(A0) < tol and norm1(A2) < tol:
            return -(Ah ** -1) * A, it + 1
    raise RuntimeError("no convergence")

if __name__ == "__main__":
    w = np.load("gpurun_out/parity_worst.npz")
    om = wl.sw_shaped_observation_model()
    for i, gpu, orc in list(zip(w["idx"], w["gpu"], w["oracle"]))[: int(sys.argv[1]) if len(sys.argv) > 1 else 2]:
        b = wl.sw_shaped_batch(1, first_draw=int(i))
        M = lambda a: matrix(a.tolist())
        A, B, C, D = (M(b[x][0]) for x in "ABCD")
        T, it = cr_mp(A, B, C, mpf("1e-8"))
        R = -((C * T + B) ** -1) * D
        Tn = np.array(T.tolist(), dtype=object); Rn = np.array(R.tolist(), dtype=object)
        c = dict(T=Tn, R=Rn, Q=np.diag(b["sigma"][0] ** 2), Z=om["Z"], H=np.diag(om["Hdiag"]), d=np.zeros(7), y=om["y"])
        val = loglike_mp(c, 1e-8)
        print(f"draw {i}: CR iterations {it}; 40-digit logp {mp.nstr(val, 20)}; device rel err {abs(float(gpu) - float(val)) / abs(float(val)):.2e}; "
              f"oracle rel err {abs(float(orc) - float(val)) / abs(float(val)):.2e}", flush=True)
