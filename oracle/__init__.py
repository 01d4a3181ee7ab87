"""CPU oracle for the first-order-perturbation + Kalman log-likelihood hot path.

TEST INFRASTRUCTURE ONLY.  This package is a float64 numpy/scipy restatement of the
reference algorithms (gEconpy @ /root/reference; every function cites the file:line it
follows).  It exists so that the HIP product path in ``geconpy_amd`` can be checked.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it; the product package never does, and it fails loudly when
its HIP library is missing instead of falling back to this code.

Pinning status
--------------
* gensys / cycle reduction / selection matrix / backward-direct: PINNED.  Checked in
  the build container against the reference's own function bodies (executed by AST
  extraction, ``tests/golden/_ref_extract.py``) on the reference's golden A,B,C,D
  fixtures (``tests/_resources/expected_matrices.py``); outputs frozen under
  ``tests/golden/*.npz`` by ``tests/golden/make_golden.py``.
* Discrete Lyapunov ``P0`` and the Kalman filter log-likelihood: the RECURSION is pinned against
  statsmodels 0.12.2 (independent compiled filter found in the build container's Anaconda tree;
  ``tests/golden/make_statsmodels_golden.py`` -> ``tests/golden/statsmodels_kalman.npz``; with
  ``jitter = 0`` and complete data the conventions coincide: 4e-11 relative from a common P0).
  The pymc_extras CONVENTIONS layered on it stay restated and unpinned (jitter 1e-8 on F and P+,
  fill value -9999, masked rows kept with the full ``p``, ``d`` unmasked): the reference delegates
  the filter to ``pymc_extras>=0.12.0`` ``StandardFilter`` and the Lyapunov solve to
  ``pytensor>=3.0.4`` (``pyproject.toml:43-45``), absent from ``/root/reference`` and from this
  image; its own tests assert only finiteness and one self-consistency equality at this boundary
  (tests/model/test_statespace.py:100-115, 583-630), which ``tests/test_oracle_kalman.py`` reproduces.
"""
from .cycle_reduction import (  # noqa: F401
    cycle_reduction_core,
    cycle_reduction_numpy,
    scan_cycle_reduction,
    solve_policy_function_with_cycle_reduction,
)
from .gensys_qz import (  # noqa: F401
    check_bk_condition,
    compute_bk_eigenvalues,
    gensys,
    gensys_core,
    gensys_setup,
    gensys_T_success,
    solve_policy_function_with_gensys,
)
from .shared import (  # noqa: F401
    compute_selection_matrix,
    policy_function_adjoints,
    policy_residual,
    solve_policy_function_with_backward_direct,
)
from .statespace import (  # noqa: F401
    DEFAULT_CONVENTIONS,
    FilterConventions,
    JITTER_DEFAULT,
    MISSING_FILL,
    autocorrelation_matrices,
    kalman_filter_logp,
    solve_discrete_lyapunov,
    solve_kalman_logp,
)
