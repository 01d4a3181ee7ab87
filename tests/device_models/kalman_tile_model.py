"""Executable model of kalman_mf_kernel (csrc/dsge_kalman_mf.hpp): what the tile-layout filter computes, step for step, in numpy.

What differs from the reference's recursion (oracle.kalman_filter_logp) and is modelled here:
  * the exact reduction to the retained variables U = S u O (S: variables with a non-zero column of T, O: observed non-states),
    states first, and products restricted to the state block:  W = P+[S,S] Tc',  P' = Tc W + Q  with Tc = T[U, S];
  * only the UPPER triangle of P and Q is carried (the upper 4 x 4 tiles on the device; the lower triangle is the mirror image):
    the downdate is the product  P+ = P + K (-V)' + jit_P I,  V = P Zm' + jit_V K,  evaluated for i <= j only -- no
    symmetrisation step anywhere;
  * the p x p inverse by Gauss-Jordan without pivoting whose reciprocals are a hardware seed (4.6e-8 relative,
    tools/latency_probe/rcp_probe.hip) plus ONE Newton step; det F from the pivots themselves;
  * the steady-state switch: once max |P+_t - P+_{t-1}| over the state block <= tol * max diag(P_t), the gain, F^-1 and det F are
    frozen and only the mean recursion runs while the missing-data mask stays the same.
"""
import numpy as np


def _mirror(U):
    """Full symmetric matrix from its upper triangle."""
    return np.triu(U) + np.triu(U, 1).T


def _rcp_one_newton(x, seed_error):
    inv = (1.0 / x) * (1.0 + seed_error)
    return inv * (2.0 - x * inv)


def retained_variables(T, Z):
    """positions -> original variables: states (non-zero columns of T) first, then the observed non-states."""
    m = T.shape[0]
    is_state = np.any(T != 0.0, axis=0)
    observed = np.any(Z != 0.0, axis=0)
    states = [j for j in range(m) if is_state[j]]
    extra = [j for j in range(m) if observed[j] and not is_state[j]]
    return np.array(states + extra, dtype=int), len(states)


def kalman_tile_logp(y, T, RQR, Z, Hdiag, d, P0, jit_F=1e-8, jit_P=1e-8, jit_V=1e-8, fill=-9999.0, steady_tol=1e-14,
                     seed_error=4.6e-8, return_steady_step=False):
    """log-likelihood of the selector-Z model by the device's recursion.  P0: the stationary covariance (full model)."""
    p = Z.shape[0]
    perm, s = retained_variables(T, Z)
    m = len(perm)
    assert all(np.count_nonzero(Z[o]) == 1 for o in range(p)), "selector design matrix"
    zpos = np.array([int(np.where(perm == np.flatnonzero(Z[o])[0])[0][0]) for o in range(p)])
    zval = np.array([Z[o, np.flatnonzero(Z[o])[0]] for o in range(p)])
    Tc = T[np.ix_(perm, perm[:s])]  # m x s
    Q = np.triu(0.5 * (RQR + RQR.T)[np.ix_(perm, perm)])
    P = np.triu(P0[np.ix_(perm, perm)])
    iu = np.triu_indices(m)
    a = np.zeros(m)
    quad, mant, expo, n_steps, n_entries = 0.0, 1.0, 0, 0, 0
    steady_step = -1
    P_plus_old = None
    frozen = None
    t = 0
    n_t = y.shape[0]
    while t < n_t:
        yt = y[t]
        obs = ~(np.isnan(yt) | (yt == fill))
        c = np.where(obs, zval, 0.0)
        Pfull = _mirror(P)
        PZt = Pfull[:, zpos] * zval[None, :]  # unmasked panel
        v = np.where(obs, yt, 0.0) - (d + c * a[zpos])
        F = (c[:, None] * PZt[zpos, :]) * obs[None, :] + np.diag(np.where(obs, Hdiag, 0.0) + jit_F)
        Fi = F.copy()
        sm, se = 1.0, 0
        inv_own = np.ones(p)
        for j in range(p):
            piv = Fi[j, j]
            inv = _rcp_one_newton(piv, seed_error)
            rowj = Fi[j, :].copy()
            ci = Fi[:, j] * inv
            ci[j] = 0.0
            Fi = Fi - np.outer(ci, rowj)
            Fi[:, j] = -ci
            Fi[j, j] = 1.0
            Fi[j, :] = np.where(np.arange(p) == j, 1.0, rowj)  # (the pivot row stays unscaled until the end: inv_own)
            inv_own[j] = inv
            mm, ee = np.frexp(piv)
            sm *= mm
            se += int(ee)
        Finv = Fi * inv_own[:, None]
        if obs.any():
            quad += v @ Finv @ v
            mant, e2 = np.frexp(mant * sm)
            expo += int(e2) + se
            n_steps += 1
            n_entries += int(obs.sum())
        K = (PZt @ Fi.T) * inv_own[None, :] * obs[None, :]
        V = PZt * obs[None, :] + jit_V * K
        af = a + K @ v
        # (e) downdate as a product, upper triangle only
        Pn = np.zeros_like(P)
        Pn[iu] = (P - K @ V.T + jit_P * np.eye(m))[iu]
        pm = np.max(np.abs(np.diag(P)))
        steady = False
        if steady_tol > 0.0 and t > 0 and P_plus_old is not None:
            nb = min(4 * ((s + 3) // 4), m)  # the padded state block the device keeps as a full square
            dmax = np.max(np.abs((Pn - P_plus_old)[:nb, :nb][np.triu_indices(nb)]))
            steady = bool(dmax <= steady_tol * pm)
        P_plus_old = Pn.copy()
        # (f) prediction through the state block
        Pss = _mirror(Pn)[:s, :s]
        W = Pss @ Tc.T  # s x m
        X = Tc @ W
        P = np.zeros_like(P)
        P[iu] = (X + Q)[iu]
        a = Tc @ af[:s]
        t += 1
        if not steady:
            continue
        if steady_step < 0:
            steady_step = t
        while t < n_t:
            ys = y[t]
            obs_s = ~(np.isnan(ys) | (ys == fill))
            if not np.array_equal(obs_s, obs):
                break
            v = np.where(obs_s, ys, 0.0) - (d + c * a[zpos])
            if obs.any():
                quad += v @ Finv @ v
                mant, e2 = np.frexp(mant * sm)
                expo += int(e2) + se
                n_steps += 1
                n_entries += int(obs.sum())
            a = Tc @ (a + K @ v)[:s]
            t += 1
    logdet = np.log(mant) + expo * np.log(2.0)
    lp = -0.5 * (n_steps * p * np.log(2 * np.pi) + logdet + quad)
    return (lp, steady_step) if return_steady_step else lp
