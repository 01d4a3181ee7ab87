"""Host-side mirror of the constant bookkeeping of ``DSGEStateSpace`` that the fused, augmented
evaluation needs (gEconpy/model/statespace.py): which deterministic chains are appended to the state
vector and what the design matrix looks like.  Everything here is index arithmetic on model metadata --
it runs once per model, not per draw; the per-draw work (un-permutation, building ``T_aug``/``R_aug``,
Lyapunov, filter) happens on the device in ``dsge_solve_kalman_logp_augmented_batched``.

Only the selector / cumulator branch of ``_make_design_matrix`` (:282-296) is constant; observation
equations make ``Z`` parameter dependent (:298-332) -- then build ``Z`` per draw and pass it batched.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import _lib, batched

CUMULATOR_AGGREGATIONS = ("sum", "mean")  # statespace.py:48


@dataclass
class StateAugmentation:
    """Layout of the augmented state vector: model variables, cumulator chains, observation-lag chains."""

    state_names: list            # model variables, in the order of T's rows/columns after un-permutation
    cumulator_variables: list    # temporally aggregated observed variables (statespace.py:562-571)
    aggregation_period: int
    obs_lag_depths: dict = field(default_factory=dict)   # variable -> number of lag slots (:1043-1049)
    link_rows: np.ndarray = None  # int32: T_aug[link_rows[i], link_cols[i]] = 1
    link_cols: np.ndarray = None
    obs_lag_starts: dict = field(default_factory=dict)

    @property
    def n(self):
        return len(self.state_names)

    @property
    def n_cumulator_states(self):  # :558-559
        return len(self.cumulator_variables) * (self.aggregation_period - 1)

    @property
    def n_obs_lag_states(self):  # :586-587
        return sum(self.obs_lag_depths.values())

    @property
    def m(self):
        return self.n + self.n_cumulator_states + self.n_obs_lag_states

    @property
    def augmented_state_names(self):  # :573-591
        names = list(self.state_names)
        names += [f"{v}_cumulator_lag{lag}" for v in self.cumulator_variables for lag in range(1, self.aggregation_period)]
        names += [f"{v}_obs_lag{k}" for v, depth in self.obs_lag_depths.items() for k in range(1, depth + 1)]
        return names

    def obs_lag_column(self, var_name, lag):  # :593-596 (lag < 0)
        return self.obs_lag_starts[var_name] + (-lag - 1)


def build_augmentation(state_names, temporal_aggregation=None, aggregation_period=4, obs_lag_depths=None,
                       obs_equation_names=()):
    """The unit entries of the constant blocks ``[F C]`` of ``_augment_transition`` (statespace.py:598-650) and
    ``_append_obs_lag_block`` (:652-694), as (row, col) links into the m x m augmented transition."""
    state_names = list(state_names)
    temporal_aggregation = dict(temporal_aggregation or {})
    obs_lag_depths = dict(obs_lag_depths or {})
    cum_vars = [v for v, method in temporal_aggregation.items()
                if method in CUMULATOR_AGGREGATIONS and v not in obs_equation_names]  # :562-571
    n = len(state_names)
    n_cum_lags = aggregation_period - 1
    rows, cols = [], []
    for agg_pos, var in enumerate(cum_vars):  # F[agg_pos * n_cum_lags, orig_idx] = 1 (:640-643); shift companion (:631-634)
        base = n + agg_pos * n_cum_lags
        if n_cum_lags >= 1:
            rows.append(base)
            cols.append(state_names.index(var))
        for j in range(1, n_cum_lags):
            rows.append(base + j)
            cols.append(base + j - 1)
    k_prev = n + len(cum_vars) * n_cum_lags
    starts, offset = {}, k_prev
    for var, depth in obs_lag_depths.items():  # consecutive slots in insertion order (:1072-1076)
        starts[var] = offset
        rows.append(offset)
        cols.append(state_names.index(var))  # F_lag[block_start, orig_idx] = 1 (:684-685)
        for j in range(1, depth):
            rows.append(offset + j)
            cols.append(offset + j - 1)      # C_lag[block_start + j, block_start + j - 1] = 1 (:686-687)
        offset += depth
    return StateAugmentation(state_names, cum_vars, aggregation_period, obs_lag_depths,
                             np.asarray(rows, dtype=np.int32), np.asarray(cols, dtype=np.int32), starts)


def make_design_matrix(aug, observed_states, temporal_aggregation=None):
    """Constant selector design matrix of ``_make_design_matrix`` (statespace.py:282-296): unit weight on the observed
    variable's column, or weight (1 for "sum", 1/s for "mean") on the column AND its cumulator slots."""
    temporal_aggregation = dict(temporal_aggregation or {})
    n_cum_lags = aug.aggregation_period - 1
    Z = np.zeros((len(observed_states), aug.m))
    for i, name in enumerate(observed_states):
        orig_idx = aug.state_names.index(name)
        method = temporal_aggregation.get(name)
        if method in CUMULATOR_AGGREGATIONS:
            agg_pos = aug.cumulator_variables.index(name)
            cum_start = aug.n + agg_pos * n_cum_lags
            weight = 1.0 / aug.aggregation_period if method == "mean" else 1.0
            Z[i, orig_idx] = weight
            Z[i, cum_start:cum_start + n_cum_lags] = weight
        else:
            Z[i, orig_idx] = 1.0
    return Z


def solve_kalman_logp_augmented_batched(A, B, C, D, Q, Z, y, aug, inv_var_order=None, d=None, Hdiag=None, q_mode=None,
                                        solver="cycle_reduction", tol=1e-6, max_iter=50, jitter=batched.JITTER_DEFAULT,
                                        missing_fill_value=batched.MISSING_FILL, return_statespace=False, options=None):
    """One fused evaluation per draw with the un-permutation and augmentation of
    ``DSGEStateSpace.make_symbolic_graph`` done on the device.  ``Z`` is (p, m) (``make_design_matrix``) or
    (batch, p, m); ``inv_var_order`` the (n,) permutation of statespace.py:217-220 (None = identity).
    Returns dict(logp, status, resid[, T_aug, R_aug])."""
    A, B, C = batched._check_abc(A, B, C)
    D = batched._f64(D, 3)
    y = batched._f64(y, 2)
    nb, n, _ = A.shape
    k = D.shape[2]
    T_len, p = y.shape
    m = aug.m
    if aug.n != n:
        raise ValueError("augmentation was built for a different number of model variables")
    Q, code = batched._resolve_q(Q, q_mode, nb, k)
    Z, zb, d, db, Hdiag, hb = batched._obs_args(Z, d, Hdiag, nb, p, m)
    inv = None if inv_var_order is None else np.ascontiguousarray(inv_var_order, dtype=np.int32)
    if inv is not None and sorted(inv.tolist()) != list(range(n)):
        raise ValueError("inv_var_order must be a permutation of range(n)")
    lr = np.ascontiguousarray(aug.link_rows, dtype=np.int32)
    lc = np.ascontiguousarray(aug.link_cols, dtype=np.int32)
    # structure hints of the AUGMENTED system: a column of T_aug is non-zero iff it is a state column of T (non-zero
    # column of A, un-permuted) or the source of a link
    a_cols = np.any(A.reshape(-1, n) != 0, axis=0)
    if inv is not None:
        a_cols = a_cols[inv]
    aug_cols = np.zeros(m, dtype=bool)
    aug_cols[:n] = a_cols
    aug_cols[lc] = True
    n_state_hint = int(aug_cols.sum())
    logp = np.empty(nb)
    status = np.empty(nb, dtype=np.int32)
    resid = np.empty(nb)
    Ta = np.empty((nb, m, m)) if return_statespace else None
    Ra = np.empty((nb, m, k)) if return_statespace else None
    nl = batched.lead_hint(C, tol) if solver == "gensys" else 0
    with _lib.options_scope(options):
        _lib.check(
            _lib.load().dsge_solve_kalman_logp_augmented_batched_host(
                batched._ptr(A), batched._ptr(B), batched._ptr(C), batched._ptr(D), batched._ptr(Q), code, batched._ptr(Z), zb,
                batched._ptr(d), db, batched._ptr(Hdiag), hb, batched._ptr(y), nb, n, k, p, T_len, _lib.SOLVER_CODES[solver],
                float(tol), int(max_iter), float(jitter), float(missing_fill_value), m, batched._ptr(inv), len(lr),
                batched._ptr(lr), batched._ptr(lc), n_state_hint, batched.selector_hint(Z), nl, batched._ptr(logp),
                batched._ptr(status), batched._ptr(Ta), batched._ptr(Ra), batched._ptr(resid)
            )
        )
    out = dict(logp=logp, status=status, resid=resid)
    if return_statespace:
        out.update(T_aug=Ta, R_aug=Ra)
    return out
