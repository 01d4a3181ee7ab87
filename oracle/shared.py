"""Oracle restatement of the solver-shared algebra (TEST INFRASTRUCTURE ONLY).

  * ``compute_selection_matrix`` <- gEconpy/solvers/shared.py:74-75
  * ``policy_residual``          <- gEconpy/model/statespace.py:213
  * ``solve_policy_function_with_backward_direct`` <- gEconpy/solvers/backward_looking.py:102-134
  * ``policy_function_adjoints``  <- ``o1_policy_function_adjoints`` gEconpy/solvers/shared.py:12-71
"""
from __future__ import annotations

import numpy as np


def compute_selection_matrix(B, C, D, T):
    """R = -(C T + B)^-1 D."""
    return -np.linalg.solve(C @ T + B, D)


def policy_residual(A, B, C, T):
    """sum((A + B T + C T T)^2), evaluated in solver order."""
    return float(np.square(A + B @ T + C @ T @ T).sum())


def solve_policy_function_with_backward_direct(A, B, C, D):
    """T = (-B)^-1 A, R = -B^-1 D for models without leads (C == 0)."""
    del C
    return np.linalg.solve(-B, A), -np.linalg.solve(B, D)


def policy_function_adjoints(A, B, C, T, T_bar, jitter=1e-16):
    """Reverse-mode sensitivities through ``A + B T + C T T = 0`` (shared.py:12-71).

    Solves ``(kron(T, C') + kron(I, T' C') + kron(I, B') + jitter I) vec(S) = -vec(T_bar)`` with
    column-major ``vec`` (the reference ravels ``T_bar.T`` and reshapes-then-transposes, :53,64) and
    returns ``[A_bar, B_bar, C_bar] = [S, S T', S T' T']`` (:67-69).
    """
    del A
    n = T.shape[0]
    eye = np.eye(n)
    K = np.kron(T, C.T) + np.kron(eye, T.T @ C.T) + np.kron(eye, B.T)
    K = K + jitter * np.eye(n * n)
    vec_S = np.linalg.solve(K, -T_bar.T.ravel())
    S = vec_S.reshape((n, n)).T
    return S, S @ T.T, S @ T.T @ T.T
