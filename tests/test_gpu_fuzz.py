"""Randomised GPU-vs-oracle parity sweep (scripts/fuzz_parity.py): random meshes (hex, polyhedral, decomposed), jitter up
to near-inversion, random parameters, constraints and layer patches; every case must be bit-compatible."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_cases(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "10", str(seed)], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count(" ok ") == 10
