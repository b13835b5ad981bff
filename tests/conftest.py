import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    """Fixtures generated from the reference's own Python (tests/golden/make_golden.py)."""
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_golden.npz"))


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.load()
    return O
