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
