"""GPU: randomized configurations of the fused evaluation against the oracle (tools/fuzz_fused.py: 6..64 variables, 1..12
shocks, 1..8 observables, 1..40 periods; selector and dense design matrices, diagonal and full shock covariances, cycle
reduction and gensys, missing observations, policy outputs).  The first run of this fuzz found the round-1 bug of
test_more_shocks_than_the_reduced_tile_is_wide."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("seed", [101, 202, 303])
def test_fused_evaluation_fuzz(seed):
    import fuzz_fused

    assert fuzz_fused.run(seed, 60, verbose=False) == 0


@pytest.mark.parametrize("seed", [7, 8])
def test_gradient_fuzz(seed):
    """Directional derivatives of the device gradient against extrapolated central differences of the oracle over random
    configurations (6..56 variables, selector Z, batches of 1 / 2 / 5): 1e-6 relative (observed: <= 1e-9)."""
    import fuzz_grad

    assert fuzz_grad.run(seed, 40, verbose=False, rtol=1e-6) == 0


@pytest.mark.parametrize("seed", [17, 18])
def test_gradient_fuzz_full_covariance_and_dense_design(seed):
    """The same with the shock covariance (diagonal / full symmetric: Q_bar) and the design matrix (selector / dense
    observation equations through the state augmentation: Z_bar) drawn at random per trial (statespace.py:247-251, 298-332)."""
    import fuzz_grad

    assert fuzz_grad.run(seed, 40, verbose=False, rtol=1e-6, modes=True) == 0


@pytest.mark.parametrize("seed", [1, 2])
def test_cycle_reduction_fuzz(seed):
    """Cycle reduction alone, 3..64 variables, three tolerances, default kernels / one wavefront for 49..64 / dense kernel:
    status and ITERATION COUNTS equal to the oracle's, T within 1e-9 (fixed bar, no conditioning allowance)."""
    import fuzz_cr

    assert fuzz_cr.run(seed, 60, verbose=False) == 0


@pytest.mark.parametrize("seed", [4, 5])
def test_standalone_filter_fuzz(seed):
    """dsge_kalman_logp_batched (T, R, Q given) at random sizes up to 64 states / 16 observables: column-sparse and dense
    transitions, selector / dense / batched design matrices, diagonal and full Q, missing data."""
    import fuzz_kalman

    assert fuzz_kalman.run(seed, 50, verbose=False) == 0


@pytest.mark.parametrize("seed", [1, 2])
def test_mixed_structure_batches_fuzz(seed):
    """Batches in which a quarter of the draws violate the structure hints (more states, fewer static variables): every
    second pass of the kernel cascades runs, every sampled draw matches the oracle."""
    import fuzz_mixed

    assert fuzz_mixed.run(seed, 20, verbose=False) == 0


@pytest.mark.parametrize("seed", [1, 2])
def test_standalone_pullbacks_fuzz(seed):
    """dsge_policy_adjoints_batched against the oracle's Kronecker solve and dsge_selection_adjoints_batched against the
    closed form, 3..56 variables.  Policy adjoints 1e-9 (the doubling series loses digits on rare non-normal systems; those
    are refined once in the kernel's second pass, see tools/fuzz_adjoints.py), selection pullback 1e-9."""
    import fuzz_adjoints

    assert fuzz_adjoints.run(seed, 30, verbose=False) == 0


@pytest.mark.parametrize("seed", [1, 2])
def test_gensys_and_bk_fuzz(seed):
    """gensys alone at random sizes (3..48 variables, pencils up to 60) against the LAPACK-based oracle: T, success, eu, and
    the Blanchard-Kahn counts; a fifth of the batches carries an explosive draw."""
    import fuzz_gensys

    assert fuzz_gensys.run(seed, 40, verbose=False) == 0


@pytest.mark.parametrize("seed", [1, 2])
def test_raw_pencil_gensys_fuzz(seed):
    """dsge_gensys_pencil_batched on the pencils of random models, plain and under a random equivalence transformation
    (dense pencils, same solution): G1, impact, eu against the oracle's gensys."""
    import fuzz_pencil

    assert fuzz_pencil.run(seed, 30, verbose=False) == 0


def test_chunked_host_path_fuzz():
    """The `*_host` twin of the fused evaluation (two chunks from 512 draws, four from 2048, on two streams with leased
    arenas) against the device-resident single pipeline at random sizes: same status, logp bit-identical (1e-10 where the
    structure hints select another gensys kernel)."""
    import fuzz_hostpath

    assert fuzz_hostpath.run(3, 8, verbose=False) == 0


def test_autocorrelation_and_lyapunov_fuzz():
    """dsge_autocorrelation_batched (latent and observed, correlation and covariance, lag steps) and dsge_lyapunov_batched at
    random sizes up to 64 states against the oracle: 1e-9."""
    import fuzz_acf

    assert fuzz_acf.run(1, 40, verbose=False) == 0


@pytest.mark.parametrize("seed", [1, 2])
def test_generated_theta_kernels_fuzz(seed):
    """Random parameter programs (jacobian_codegen): theta -> A, B, C, D, q / Z, d and both pullbacks against sympy's
    lambdify of the same expressions and derivatives; the parameter names include the generated kernel's own identifiers
    (theta, A, q, draw, x0 ...) and names that are no C identifiers (rho^A, sigma.e)."""
    import fuzz_theta

    assert fuzz_theta.run(seed, 6, verbose=False) == 0


@pytest.mark.parametrize("seed", [1, 2])
def test_second_order_fuzz(seed):
    """The second-order path at random sizes / structures (tools/fuzz_second_order.py: 4..34 variables, pruned states up to
    208, observed non-states, missing observations, samples that reach the steady-state switch): coefficients to 1e-9 and the
    pruned quasi-likelihood to 1e-8 against oracle/second_order.py (parity unpinned against the reference, which raises)."""
    import fuzz_second_order

    assert fuzz_second_order.run(seed, 10, verbose=False) == 0
