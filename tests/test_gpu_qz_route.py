"""The ordered QZ for every draw (dsge_options.gensys_doubling = 0) keeps its whole test coverage: since round 5 the library's
default solves gensys by spectral division and hands only the draws without a certificate to the QZ kernels, so the suite's
gensys tests -- written against the QZ path: window launches, pair kernels, capacity records, rescue pass, the reference's
golden systems and failure cases (tests/model/test_perturbation.py, tests/solvers/test_gensys.py of the reference), the fuzz
suite at its fixed bars -- are run a second time here in a child process whose library default is the QZ path
(environment variable DSGE_GENSYS_DOUBLING = 0, read once at load: csrc/dsge_api.hip)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.skipif(os.environ.get("DSGE_GENSYS_DOUBLING") is not None, reason="already running under an explicit default")
def test_gensys_tests_with_the_ordered_qz_for_every_draw():
    env = dict(os.environ, DSGE_GENSYS_DOUBLING="0")
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-q", "-x", "-m", "gpu", "-k", "gensys or fuzz or bk or solvab",
           "-p", "no:cacheprovider"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = "\n".join(res.stdout.splitlines()[-25:])
    assert res.returncode == 0, tail + "\n" + res.stderr[-2000:]
    assert " passed" in tail and "failed" not in tail, tail


def test_environment_default_is_what_the_library_reports():
    """dsge_options_init reports gensys_doubling = 1 without the variable and 0 / 2 under it (child processes: the variable is read
    when the library is loaded)."""
    code = "from geconpy_amd import _lib; print(_lib.make_options().gensys_doubling)"
    for val, want in ((None, "1"), ("0", "0"), ("2", "2"), ("7", "1")):
        env = {k: v for k, v in os.environ.items() if k != "DSGE_GENSYS_DOUBLING"}
        if val is not None:
            env["DSGE_GENSYS_DOUBLING"] = val
        out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-1500:]
        assert out.stdout.strip().splitlines()[-1] == want, (val, out.stdout)
