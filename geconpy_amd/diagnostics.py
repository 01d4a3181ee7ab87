"""Batched counterpart of the per-draw pipeline behind ``solvability_check``
(gEconpy/model/statistics/perturbation_diagnostics.py:105-161, :362-490).

The reference loops over draws (serially or in a fork pool whose ``imap_unordered`` scrambles the
order, :484-489); per draw it solves the steady state and linearises on the host, then
``_solve_perturbation`` -> Blanchard-Kahn check -> gEcon residual norms.  Here the part after the
linearisation runs for ALL draws in three launches and the result arrays are in input order.

The steady-state / linearisation stages stay with the caller: pass a boolean ``upstream_failed``
(or ``None``) to have those draws labelled ``"steady_state"`` as the reference would.
"""
from __future__ import annotations

import numpy as np

from . import batched


def _gecon_partition(T, R, tol):
    """statespace_to_gEcon_representation (gEconpy/model/perturbation.py:322-382): state variables =
    columns of T with an entry >= tol; entries below tol are flushed to zero."""
    PP = np.where(np.abs(T) < tol, 0.0, T)
    QQ = np.where(np.abs(R) < tol, 0.0, R)
    mask = np.abs(T).max(axis=1) >= tol  # (batch, n): column-wise max over rows
    return PP, QQ, mask


def solvability_check_batched(A, B, C, D, solver="cycle_reduction", tol=1e-8, max_iter=1000, norm_tol=1e-8,
                              backward_looking=False, upstream_failed=None):
    """Returns dict(failure_step (object array: None or the name of the failing stage),
    norm_deterministic, norm_stochastic, T, R) for a batch of linearised systems.

    Stages and labels as in ``_check_one_draw`` (:105-161): "perturbation" (solver failure),
    "blanchard-kahn" (root count: taken from the gensys existence/uniqueness codes, eu[0] = 0 too many
    unstable roots, eu[1] = 0 too few), "deterministic_norm", "stochastic_norm"; norms are NaN where not
    reached.
    """
    A, B, C = (np.ascontiguousarray(x, dtype=np.float64) for x in (A, B, C))
    D = np.ascontiguousarray(D, dtype=np.float64)
    nb, n, _ = A.shape
    failure = np.full(nb, None, dtype=object)
    det = np.full(nb, np.nan)
    sto = np.full(nb, np.nan)
    if upstream_failed is not None:
        failure[np.asarray(upstream_failed, dtype=bool)] = "steady_state"

    eff = "backward_direct" if backward_looking else solver
    if eff == "cycle_reduction":
        T, status, _ = batched.cycle_reduction_batched(A, B, C, max_iter=max_iter, tol=tol)
        ok = status == 0
        R = batched.selection_batched(B, C, D, T)
    elif eff == "gensys":
        g = batched.gensys_batched(A, B, C, D, tol=tol)
        T, R, ok = g["T"], g["R"], g["success"]
    elif eff == "backward_direct":
        T, R = batched.backward_direct_batched(A, B, D)
        ok = np.isfinite(T).all(axis=(1, 2))
    else:
        raise ValueError(f"Unknown solver {solver!r}")
    reached = failure == None  # noqa: E711
    failure[reached & ~ok] = "perturbation"

    # Blanchard-Kahn: root counting by the device QZ (skipped when gensys already decided it)
    reached = failure == None  # noqa: E711
    if eff == "cycle_reduction" and reached.any():
        eu = batched.gensys_batched(A, B, C, None, tol=tol)["eu"]
        bk_ok = (eu[:, 0] == 1) & (eu[:, 1] == 1)
        failure[reached & ~bk_ok] = "blanchard-kahn"

    reached = failure == None  # noqa: E711
    if reached.any():
        PP, QQ, mask = _gecon_partition(T, R, tol)
        # draws that share a state mask go through one launch (the mask is structural: normally one group)
        keys = {}
        for i in np.flatnonzero(reached):
            keys.setdefault(mask[i].tobytes(), []).append(i)
        for key, idx in keys.items():
            idx = np.asarray(idx)
            m = np.frombuffer(key, dtype=bool)
            d_, s_ = batched.policy_norms_batched(A[idx], B[idx], C[idx], D[idx], PP[idx], QQ[idx], m)
            det[idx] = d_
            sto[idx] = s_
        bad_det = reached & ~(det <= norm_tol)
        failure[bad_det] = "deterministic_norm"
        bad_sto = reached & ~bad_det & ~(sto <= norm_tol)
        failure[bad_sto] = "stochastic_norm"
    return dict(failure_step=failure, norm_deterministic=det, norm_stochastic=sto, T=T, R=R)


def check_bk_condition_batched(A, B, C, D=None, tol=1e-8, return_value="bool"):
    """Batched ``check_bk_condition`` (gEconpy/model/perturbation.py:448-565): per draw, the number of eigenvalues of the
    Sims pencil with modulus > 1 must equal the number of forward-looking variables.

    ``return_value="bool"`` -> boolean array; ``"dataframe"`` -> list of per-draw pandas DataFrames with the reference's
    columns ``Modulus``, ``Real``, ``Imaginary`` (ascending modulus); ``None`` -> nothing (the counts are in
    ``batched.bk_eigenvalues_batched``).  ``D`` is accepted for signature parity and ignored, as in the reference."""
    if return_value not in ("dataframe", "bool", None):
        raise ValueError(f'Unknown return type "{return_value}"')
    out = batched.bk_eigenvalues_batched(A, B, C, tol=tol)
    if return_value is None:
        return None
    if return_value == "bool":
        return out["satisfied"]
    import pandas as pd

    frames = []
    for i in range(out["real"].shape[0]):
        m = int(out["n_eig"][i])
        re, im = out["real"][i, :m], out["imag"][i, :m]
        frames.append(pd.DataFrame({"Modulus": np.hypot(re, im), "Real": re, "Imaginary": im}))
    return frames


def sample_parameters(priors, n_samples, seed=None, method="lhs", hdi_prob=0.99):
    """Parameter draws for ``prior_solvability_check_batched`` -> (names, array (n_samples, len(priors))).

    ``priors``: dict name -> frozen ``scipy.stats`` distribution (anything with ``rvs`` / ``ppf``; the reference's priors
    are preliz distributions, which expose the same two methods) or a ``(low, high)`` pair.  ``method`` as in
    ``prior_solvability_check`` (gEconpy/model/statistics/perturbation_diagnostics.py:526-579): ``"random"`` -- Monte
    Carlo through ``rvs``; ``"lhs" | "sobol" | "halton"`` -- uniform quasi-Monte-Carlo over the central ``hdi_prob``
    interval of every prior (the reference uses the HDI; for the unimodal priors of DSGE models the two coincide up to
    skewness); ``"sobol_ppf" | "halton_ppf" | "lhs_ppf"`` -- quasi-Monte-Carlo through the inverse CDF."""
    from scipy.stats import qmc

    names = list(priors)
    d = len(names)
    rng = np.random.default_rng(seed)
    if method == "random":
        cols = []
        for nm in names:
            pr = priors[nm]
            cols.append(rng.uniform(pr[0], pr[1], n_samples) if isinstance(pr, tuple) else
                        np.asarray(pr.rvs(n_samples, random_state=rng), dtype=np.float64))
        return names, np.stack(cols, axis=1)
    base = method.removesuffix("_ppf")
    engines = {"lhs": qmc.LatinHypercube, "sobol": qmc.Sobol, "halton": qmc.Halton}
    if base not in engines:
        raise ValueError(f"unknown sampling method {method!r}")
    u = engines[base](d=d, seed=rng).random(n_samples)
    out = np.empty_like(u)
    tail = 0.5 * (1.0 - hdi_prob)
    for j, nm in enumerate(names):
        pr = priors[nm]
        if isinstance(pr, tuple):
            out[:, j] = pr[0] + (pr[1] - pr[0]) * u[:, j]
        elif method.endswith("_ppf"):
            out[:, j] = pr.ppf(np.clip(u[:, j], 1e-12, 1.0 - 1e-12))
        else:
            lo, hi = pr.ppf(tail), pr.ppf(1.0 - tail)
            out[:, j] = lo + (hi - lo) * u[:, j]
    return names, out


def prior_solvability_check_batched(program, n_samples, priors, *, seed=None, method="lhs", hdi_prob=0.99, defaults=None,
                                    device=0, **kwargs):
    """Batched ``prior_solvability_check`` (gEconpy/model/statistics/perturbation_diagnostics.py:526-579) for a model
    given as a ``JacobianProgram`` (the generated theta -> A, B, C, D kernel): sample ``n_samples`` parameter
    vectors from ``priors`` (``sample_parameters``), evaluate ALL Jacobians in one launch on the device, and push the batch
    through ``solvability_check_batched``.  Parameters without a prior take ``defaults[name]``.  A draw whose Jacobians
    are not finite is labelled ``"steady_state"`` -- in a generated program the steady state is part of the same closed
    form, so that is where a draw outside the model's domain fails (the reference labels the host stage that raises,
    :126-133).  Returns a pandas DataFrame like the reference's: one row per draw, the sampled parameters followed by
    ``failure_step``, ``norm_deterministic``, ``norm_stochastic`` -- in input order (the reference's pool returns
    completion order, :484-489).  ``**kwargs`` go to ``solvability_check_batched`` (solver, tol, max_iter, norm_tol)."""
    import pandas as pd

    from .engine import LogpEngine

    pnames = [str(p) for p in program.params]
    unknown = set(priors) - set(pnames)
    if unknown:
        raise ValueError(f"priors for names that are not parameters of the program: {sorted(unknown)}")
    if not priors:
        raise ValueError("no priors given (use solvability_check_batched with a hand-made sample instead)")
    names, draws = sample_parameters(priors, n_samples, seed=seed, method=method, hdi_prob=hdi_prob)
    defaults = defaults or {}
    missing = [p for p in pnames if p not in priors and p not in defaults]
    if missing:
        raise ValueError(f"no prior and no default for {missing}")
    theta = np.empty((n_samples, len(pnames)))
    for j, p in enumerate(pnames):
        theta[:, j] = draws[:, names.index(p)] if p in priors else float(defaults[p])
    eng = LogpEngine(device)
    A, B, C, D, _q = eng.jacobians_from_theta(program, eng.to_device(theta))
    eng.torch.cuda.synchronize()
    A, B, C, D = (x.cpu().numpy() for x in (A, B, C, D))
    bad = ~(np.isfinite(A).all(axis=(1, 2)) & np.isfinite(B).all(axis=(1, 2)) & np.isfinite(C).all(axis=(1, 2)) &
            np.isfinite(D).all(axis=(1, 2)))
    for M in (A, B, C, D):  # the kernels never see NaN / Inf inputs from a draw that is already labelled
        M[bad] = 0.0
    A[bad] = B[bad] = np.eye(A.shape[1])
    res = solvability_check_batched(A, B, C, D, upstream_failed=bad, **kwargs)
    df = pd.DataFrame(draws, columns=names)
    df["failure_step"] = res["failure_step"]
    df["norm_deterministic"] = res["norm_deterministic"]
    df["norm_stochastic"] = res["norm_stochastic"]
    return df
