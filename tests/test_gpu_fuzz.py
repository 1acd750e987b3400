"""Randomised parity sweep (tests/fuzz_parity.py) as a test: random model, chain length 0..5 with random member
directions and sequence position, board sizes 1..257, image-index subsets, NULL Jacobian patterns, rotation scales
covering every small-angle branch and |rot| > pi -- every block within the 1e-10 bar of the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_randomised_parity_sweep(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "80", str(seed)], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    print(r.stdout[-1500:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "worst" in r.stdout
