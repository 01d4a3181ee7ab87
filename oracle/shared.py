"""Oracle restatement of the solver-shared algebra (TEST INFRASTRUCTURE ONLY).

  * ``compute_selection_matrix`` <- gEconpy/solvers/shared.py:74-75
  * ``policy_residual``          <- gEconpy/model/statespace.py:213
  * ``solve_policy_function_with_backward_direct`` <- gEconpy/solvers/backward_looking.py:102-134
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
