"""Seeded synthetic workloads for the solve + Kalman-logp hot path (host side, numpy).

These generate the *inputs* of the path (Jacobian blocks A,B,C,D per parameter draw,
shock/measurement covariances, design matrix and one shared data panel); they contain
no solver.  Shapes and recipes are the ones fixed in SURVEY.md §8(d):

* ``rbc_linearized_jacobians``: closed-form A,B,C,D of the reference's
  ``tests/_resources/test_gcns/rbc_linearized.gcn`` (8 linear equations :22-49, steady
  state :6-20, calibration :56-65), variable order ``[A, C, I, K, L, R, W, Y]``.
* ``sw_shaped_system`` / ``sw_shaped_batch``: Smets-Wouters-*shaped* systems
  (n=40, 18 states, 12 forward-looking variables, 7 shocks, 7 observables).  The reference
  ships no Smets-Wouters model; the generator builds systems whose
  unique stable solution ``T*`` is known by construction.
"""
from __future__ import annotations

import numpy as np

RBC_VARIABLES = ("A", "C", "I", "K", "L", "R", "W", "Y")
RBC_CALIBRATION = dict(sigma=2.0, phi=1.5, alpha=0.35, beta=0.985, delta=0.025, rho_A=0.95, sigma_A=0.01)

SW_SHAPE = dict(n=40, n_state=18, n_lead=12, k=7, p=7, T_len=200)
SW_SEED0 = 20260630


def rbc_steady_state(sigma, phi, alpha, beta, delta):
    """rbc_linearized.gcn:6-20."""
    R = 1.0 / beta - (1.0 - delta)
    W = (1 - alpha) ** (1 / (1 - alpha)) * (alpha / R) ** (alpha / (1 - alpha))
    Y = (R / (R - delta * alpha)) ** (sigma / (sigma + phi)) * ((1 - alpha) ** (-phi) * W ** (1 + phi)) ** (
        1 / (sigma + phi)
    )
    K = alpha * Y / R
    I = delta * K  # noqa: E741
    C = Y - I
    L = (1 - alpha) * Y / W
    return dict(A=1.0, R=R, W=W, Y=Y, K=K, I=I, C=C, L=L)


def rbc_linearized_jacobians(sigma, phi, alpha, beta, delta, rho_A, sigma_A=None):
    """A = dF/dy_{t-1}, B = dF/dy_t, C = dF/dy_{t+1}, D = dF/de for F = LHS - RHS of
    rbc_linearized.gcn:22-49.  Arguments may be scalars or equal-length 1-D arrays (a
    batch of draws); returns arrays with a leading batch axis in the latter case."""
    del sigma_A
    th = np.broadcast_arrays(*(np.asarray(x, dtype=np.float64) for x in (sigma, phi, alpha, beta, delta, rho_A)))
    scalar = th[0].ndim == 0
    sigma, phi, alpha, beta, delta, rho_A = (np.atleast_1d(x) for x in th)
    nb = sigma.shape[0]
    ss = rbc_steady_state(sigma, phi, alpha, beta, delta)
    iA, iC, iI, iK, iL, iR, iW, iY = range(8)
    A = np.zeros((nb, 8, 8))
    B = np.zeros((nb, 8, 8))
    C = np.zeros((nb, 8, 8))
    D = np.zeros((nb, 8, 1))
    # 1. W = sigma C + phi L
    B[:, 0, iW] = 1.0
    B[:, 0, iC] = -sigma
    B[:, 0, iL] = -phi
    # 2. sigma/beta (C[1] - C) = R_ss R[1]
    C[:, 1, iC] = sigma / beta
    B[:, 1, iC] = -sigma / beta
    C[:, 1, iR] = -ss["R"]
    # 3. K = (1-delta) K[-1] + delta I
    B[:, 2, iK] = 1.0
    A[:, 2, iK] = -(1.0 - delta)
    B[:, 2, iI] = -delta
    # 4. Y = A + alpha K[-1] + (1-alpha) L
    B[:, 3, iY] = 1.0
    B[:, 3, iA] = -1.0
    A[:, 3, iK] = -alpha
    B[:, 3, iL] = -(1.0 - alpha)
    # 5. R = Y - K[-1]
    B[:, 4, iR] = 1.0
    B[:, 4, iY] = -1.0
    A[:, 4, iK] = 1.0
    # 6. W = Y - L
    B[:, 5, iW] = 1.0
    B[:, 5, iY] = -1.0
    B[:, 5, iL] = 1.0
    # 7. Y_ss Y = C_ss C + I_ss I
    B[:, 6, iY] = ss["Y"]
    B[:, 6, iC] = -ss["C"]
    B[:, 6, iI] = -ss["I"]
    # 8. A = rho_A A[-1] + eps
    B[:, 7, iA] = 1.0
    A[:, 7, iA] = -rho_A
    D[:, 7, 0] = -1.0
    if scalar:
        return A[0], B[0], C[0], D[0]
    return A, B, C, D


def rbc_prior_draws(batch, seed=1):
    """Seeded parameter draws approximating the GCN priors (rbc_linearized.gcn:56-65;
    the ``maxent`` Gamma priors need ``preliz``, absent here, so their [lower, upper]
    mass intervals are sampled uniformly)."""
    rng = np.random.default_rng(seed)
    return dict(
        sigma=rng.uniform(1.5, 3.0, batch),
        phi=rng.uniform(1.0, 5.0, batch),
        alpha=rng.beta(5, 9, batch),
        beta=np.clip(rng.beta(10, 1, batch), 0.5, 0.9995),
        delta=np.clip(rng.beta(1, 10, batch), 1e-3, 0.5),
        rho_A=np.clip(rng.beta(1, 5, batch), 0.0, 0.995),
        sigma_A=rng.uniform(0.001, 0.1, batch),
    )


def _rescale_spectral_radius(M, target):
    rho = np.max(np.abs(np.linalg.eigvals(M)))
    return M * (target / rho)


def sw_shaped_system(seed, n=40, n_state=18, n_lead=12, k=7):
    """One SW-shaped system.  Returns A,B,C,D,T_star with
    ``A + B T* + C T*^2 = 0`` exactly in real arithmetic, ``rho(T*) < 1`` and exactly
    ``n_lead`` unstable pencil roots (the reciprocals of eig(G))."""
    rng = np.random.default_rng(seed)
    S = _rescale_spectral_radius(rng.standard_normal((n_state, n_state)), 0.95 * rng.uniform(0.5, 1.0))
    T_star = np.zeros((n, n))
    T_star[:n_state, :n_state] = S
    T_star[n_state:, :n_state] = 0.3 * rng.standard_normal((n - n_state, n_state))
    G = np.zeros((n, n))
    G[:, n - n_lead :] = rng.standard_normal((n, n_lead))
    G = _rescale_spectral_radius(G, rng.uniform(0.3, 0.8))
    M = np.eye(n) + 0.2 * rng.standard_normal((n, n))
    C = M @ G
    B = M - C @ T_star
    A = -M @ T_star
    E = np.zeros((n, k))
    E[:k, :k] = -np.eye(k)
    D = M @ E
    return A, B, C, D, T_star


def sw_shaped_batch(batch, first_draw=0, seed0=SW_SEED0, **shape):
    """``batch`` systems, draw i seeded ``default_rng(seed0 + i)`` (bit-exact draw
    indexing: draw i is the same system on every rank/shard).  Also returns per-draw
    shock standard deviations sigma ~ U(0.005, 0.02) (k per draw)."""
    sh = dict(SW_SHAPE)
    sh.update(shape)
    n, k = sh["n"], sh["k"]
    A = np.empty((batch, n, n))
    B = np.empty((batch, n, n))
    C = np.empty((batch, n, n))
    D = np.empty((batch, n, k))
    Tst = np.empty((batch, n, n))
    sig = np.empty((batch, k))
    for b in range(batch):
        i = first_draw + b
        A[b], B[b], C[b], D[b], Tst[b] = sw_shaped_system(seed0 + i, n, sh["n_state"], sh["n_lead"], k)
        sig[b] = np.random.default_rng((seed0 + i, 1)).uniform(0.005, 0.02, k)
    return dict(A=A, B=B, C=C, D=D, T_star=Tst, sigma=sig)


SW_OBSERVED_JUMPS = (19, 22, 25, 27, 30, 33, 37)  # seven NON-state variables (18..39), four of them forward-looking (28..39)


def sw_shaped_observation_model(seed0=SW_SEED0, observed=None, **shape):
    """Shared pieces: Z selects variables 0..p-1 (``observed``: any other p variables, e.g. ``SW_OBSERVED_JUMPS`` --
    a Smets-Wouters data set observes growth rates, inflation, hours: jump variables, ``_make_design_matrix`` allows any,
    statespace.py:260-332), H = diag(1e-4), and one data panel ``y`` (T_len x p) simulated from draw 0's solution
    ``x_t = T* x_{t-1} + R e_t`` (R = [I_k; 0] by construction) plus measurement noise."""
    sh = dict(SW_SHAPE)
    sh.update(shape)
    n, k, p, T_len = sh["n"], sh["k"], sh["p"], sh["T_len"]
    obs = np.arange(p) if observed is None else np.asarray(observed, dtype=np.int64)
    if obs.shape != (p,) or obs.min() < 0 or obs.max() >= n or len(set(obs.tolist())) != p:
        raise ValueError(f"observed must name {p} distinct variables in 0..{n - 1}")
    d0 = sw_shaped_batch(1, 0, seed0, **shape)
    T_star = d0["T_star"][0]
    sig = d0["sigma"][0]
    rng = np.random.default_rng((seed0, 2))
    Z = np.zeros((p, n))
    Z[np.arange(p), obs] = 1.0
    Hdiag = np.full(p, 1e-4)
    x = np.zeros(n)
    y = np.empty((T_len, p))
    for t in range(T_len):
        e = rng.standard_normal(k) * sig
        x = T_star @ x
        x[:k] += e
        y[t] = Z @ x + rng.standard_normal(p) * np.sqrt(Hdiag)
    return dict(Z=Z, Hdiag=Hdiag, y=y)


def shard_bounds(batch, world_size, rank):
    """Contiguous draw shard of ``rank``: ``[lo, hi)`` with the ragged tail on the last
    rank."""
    per = batch // world_size
    lo = rank * per
    hi = batch if rank == world_size - 1 else lo + per
    return lo, hi


def rbc_batch(batch, first_draw=0, seed=1, T_len=200):
    """BASELINE.json configs[1]: the RBC model at ``batch`` seeded prior draws (closed-form Jacobians),
    observed series Y, T_len periods of data default_rng(0).normal(0, 0.05).  Same dict layout as
    ``sw_shaped_batch`` + observation model; draw i of the GLOBAL batch `first_draw + batch` (the prior draws of a batch depend on its total size)."""
    th = rbc_prior_draws(first_draw + batch, seed=seed)
    th = {k_: v[first_draw:] for k_, v in th.items()}
    A, B, C, D = rbc_linearized_jacobians(**th)
    Z = np.zeros((1, 8))
    Z[0, RBC_VARIABLES.index("Y")] = 1.0
    y = np.random.default_rng(0).normal(0, 0.05, (T_len, 1))
    return dict(A=A, B=B, C=C, D=D, sigma=th["sigma_A"][:, None]), dict(Z=Z, Hdiag=np.zeros(1), y=y)


FULL_NK_SEED0 = 20260777


def full_nk_batch(batch, first_draw=0, T_len=200, rel=1e-3):
    """SURVEY 8d, non-synthetic sanity configuration: the reference's ``full_nk`` golden system (24 variables, 4 shocks,
    14 forward-looking variables, pencil N = 38; tests/_resources/expected_matrices.py via tests/golden/reference_goldens.npz)
    replicated with seeded relative perturbations of the non-zero entries of A, B, C (``default_rng(FULL_NK_SEED0 + i)``
    for draw i).  Observed: the first three variables with measurement error 1e-4 (4 shocks, 3 series), ``T_len`` periods
    of ``default_rng(0).normal(0, 0.01)`` data.  Same layout as ``rbc_batch``."""
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                        "reference_goldens.npz")
    g = np.load(path)
    A0, B0, C0, D0 = (np.asarray(g[f"full_nk_{x}"], dtype=np.float64) for x in "ABCD")
    n, k = D0.shape
    A = np.empty((batch, n, n))
    B = np.empty((batch, n, n))
    C = np.empty((batch, n, n))
    for j in range(batch):
        rng = np.random.default_rng(FULL_NK_SEED0 + first_draw + j)
        A[j] = A0 * (1.0 + rel * rng.standard_normal(A0.shape))
        B[j] = B0 * (1.0 + rel * rng.standard_normal(B0.shape))
        C[j] = C0 * (1.0 + rel * rng.standard_normal(C0.shape))
    D = np.broadcast_to(D0, (batch, n, k)).copy()
    p = 3
    Z = np.zeros((p, n))
    Z[np.arange(p), np.arange(p)] = 1.0
    y = np.random.default_rng(0).normal(0, 0.01, (T_len, p))
    return dict(A=A, B=B, C=C, D=D, sigma=np.full((batch, k), 0.01)), dict(Z=Z, Hdiag=np.full(p, 1e-4), y=y)


def sw_theta_draws(batch, first_draw=0, seed0=SW_SEED0, n_state=18, n_lead=12, k=7):
    """Parameter draws of ``jacobian_codegen.sw_shaped_program`` as a (batch, 37) array: column scalings a, c ~ U(-1, 1),
    shock standard deviations sigma ~ U(0.005, 0.02); draw i is seeded ``default_rng((seed0 + i, 3))`` whichever shard
    generates it."""
    th = np.empty((batch, n_state + n_lead + k))
    for b in range(batch):
        rng = np.random.default_rng((seed0 + first_draw + b, 3))
        th[b, : n_state + n_lead] = rng.uniform(-1.0, 1.0, n_state + n_lead)
        th[b, n_state + n_lead :] = rng.uniform(0.005, 0.02, k)
    return th


def sw_theta_jacobians(theta, seed=None):
    """Host (numpy) twin of ``sw_shaped_program``: the same affine map theta -> A, B, C, D, q -- the inputs of the path for
    the checker, no solver in it."""
    from .jacobian_codegen import SW_THETA_SCALE

    theta = np.atleast_2d(np.asarray(theta, dtype=np.float64))
    A0, B0, C0, D0, _ = sw_shaped_system(SW_SEED0 if seed is None else seed)
    S = np.flatnonzero((A0 != 0).any(axis=0))
    Lc = np.flatnonzero((C0 != 0).any(axis=0))
    nb = theta.shape[0]
    A = np.repeat(A0[None], nb, axis=0)
    C = np.repeat(C0[None], nb, axis=0)
    A[:, :, S] = A0[None][:, :, S] * (1.0 + SW_THETA_SCALE * theta[:, None, : len(S)])
    C[:, :, Lc] = C0[None][:, :, Lc] * (1.0 + SW_THETA_SCALE * theta[:, None, len(S) : len(S) + len(Lc)])
    B = np.repeat(B0[None], nb, axis=0)
    D = np.repeat(D0[None], nb, axis=0)
    return A, B, C, D, theta[:, len(S) + len(Lc) :] ** 2


# ---- BASELINE.json configs[4]: second-order perturbation + pruned-state-space filter on the SW-shaped systems -------------

SW2_SEED = 20260931
SW2_NNZ_PER_EQUATION = 12


def second_order_hessian_pattern(A0, C0, k, nnz_per_eq=SW2_NNZ_PER_EQUATION, seed=SW2_SEED):
    """Sparsity pattern of the model Hessian d2F_i / dz_a dz_b, z = [y-; y; y+; u], as the device takes it: an int32 (nnz, 3)
    array of (equation, z_a, z_b) with z_a <= z_b, sorted by equation, shared by every draw of a model.  The reference builds
    no second derivatives (gEconpy/model/perturbation.py:97-98), so the pattern is synthetic: ``nnz_per_eq`` seeded pairs per
    equation among the z entries the equation can depend on -- y- only through the state variables (non-zero columns of A),
    y+ only through the forward-looking ones (non-zero columns of C)."""
    n = A0.shape[0]
    S = np.flatnonzero((A0 != 0).any(axis=0))
    L = np.flatnonzero((C0 != 0).any(axis=0))
    allowed = np.concatenate([S, n + np.arange(n), 2 * n + L, 3 * n + np.arange(k)])
    rng = np.random.default_rng(seed)
    idx = []
    for i in range(n):
        seen = set()
        while len(seen) < nnz_per_eq:
            a, b = sorted(int(x) for x in rng.choice(allowed, 2))
            seen.add((a, b))
        idx += [(i, a, b) for a, b in sorted(seen)]
    return np.asarray(idx, dtype=np.int32)


def sw_second_order_batch(batch, first_draw=0, seed0=SW_SEED0, **shape):
    """``sw_shaped_batch`` + per-draw Hessian values on the shared pattern of ``second_order_hessian_pattern`` (draw i seeded
    ``default_rng((seed0 + i, 4))``): entries ~ N(0, 1).  -> the dict of ``sw_shaped_batch`` with ``hess_idx`` (nnz, 3) int32
    and ``hess_val`` (batch, nnz)."""
    b = sw_shaped_batch(batch, first_draw, seed0, **shape)
    # (the pattern is a property of the model, not of a draw: an empty batch takes it from the first draw of the family)
    pat = b if batch else sw_shaped_batch(1, first_draw, seed0, **shape)
    idx = second_order_hessian_pattern(pat["A"][0], pat["C"][0], pat["D"].shape[2])
    val = np.empty((batch, len(idx)))
    for j in range(batch):
        val[j] = np.random.default_rng((seed0 + first_draw + j, 4)).standard_normal(len(idx))
    b["hess_idx"] = idx
    b["hess_val"] = val
    return b
