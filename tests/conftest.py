import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the record of a run shows where its time went (the GPU suite has a wall-clock limit at the driver)
    if getattr(config.option, "durations", None) is None:
        config.option.durations = 10


def rel_linf(a, b):
    """max_i |a_i - b_i|_inf / max_i |b_i|_inf  (BASELINE.md section 3)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (denom if denom > 0 else 1.0))


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle_ffi
    oracle_ffi.build()
    return oracle_ffi
