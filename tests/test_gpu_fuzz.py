"""A short run of every mode of tests/fuzz_parity.py (randomised differential tests of the build, the multi-rank
build, readmap + merge + propagate_from_end, the walk and the frontier kernels against the oracle).  Long runs are
made with the tool itself; this keeps the modes alive."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", ["", "FUZZ_SHARDED", "FUZZ_READMAP", "FUZZ_WALK", "FUZZ_DISCOVER"])
def test_fuzz_mode(mode):
    env = dict(os.environ)
    if mode:
        env[mode] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "6", "3"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "random builds equal to the oracle" in r.stdout


@pytest.mark.parametrize("mode", ["", "FUZZ_POPULATION"])
def test_fuzz_population_tracer(mode):
    """tests/fuzz_population_tracer.py: the push tracer against the oracle's restatement, and the sharded merge + marked
    readmap build against the single-GPU merge + migration (long runs: DESIGN.md section 2)."""
    env = dict(os.environ)
    if mode:
        env[mode] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_population_tracer.py"), "8", "7"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "cases equal" in r.stdout
