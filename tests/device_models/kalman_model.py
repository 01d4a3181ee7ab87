"""Executable model of the device Kalman update (kalman_sel_kernel / kalman_kernel):
downdate form  P+ = P - K (P Zm' + jitter K)' + jitter I  with K = P Zm' F^-1, log det F from the
Gauss-Jordan pivots accumulated as a product, instead of the reference's Joseph form."""
import numpy as np


def kalman_downdate_logp(y, T, R, Q, Z, Hdiag, d, P0, jitter=1e-8, fill=-9999.0):
    m, p = T.shape[0], Z.shape[0]
    a = np.zeros(m)
    P = P0.copy()
    RQR = R @ Q @ R.T
    RQR = 0.5 * (RQR + RQR.T)
    quad = 0.0
    mant, expo, n_steps = 1.0, 0, 0
    for t in range(y.shape[0]):
        yt = y[t]
        miss = np.isnan(yt) | (yt == fill)
        w = (~miss).astype(float)
        Zm = w[:, None] * Z
        v = np.where(miss, 0.0, yt) - (d + Zm @ a)
        PZt = P @ Zm.T
        F = Zm @ PZt + np.diag(w * Hdiag) + jitter * np.eye(p)
        # Gauss-Jordan inverse without pivoting; pivots give det F
        Fi = F.copy()
        step_m, step_e = 1.0, 0
        for j in range(p):
            piv = Fi[j, j]
            mm, ee = np.frexp(piv)
            step_m *= mm
            step_e += ee
            inv = 1.0 / piv
            col = Fi[:, j].copy()
            row = Fi[j, :].copy()
            Fi = Fi - np.outer(col * inv, row)
            Fi[j, :] = row * inv
            Fi[:, j] = -col * inv
            Fi[j, j] = inv
        K = PZt @ Fi
        if not miss.all():
            quad += v @ Fi @ v
            mant, e2 = np.frexp(mant * step_m)
            expo += e2 + step_e
            n_steps += 1
        a = a + K @ v
        P = P - K @ (PZt + jitter * K).T + jitter * np.eye(m)
        a = T @ a
        X = T @ (P @ T.T)
        P = 0.5 * (X + X.T) + RQR
    logdet = np.log(mant) + expo * np.log(2.0)
    return -0.5 * (n_steps * p * np.log(2 * np.pi) + logdet + quad)
