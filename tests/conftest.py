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


def assert_close_but_few(a, b, rtol=0.0, atol=0.0, frac=1e-3, cap=None, err_msg=""):
    """allclose for all but a fraction `frac` of the elements (frac > 0: at least one element may differ), the outliers bounded by
    `cap` (absolute).  For comparisons of LATER commands of a trace between two implementations of the update whose means
    agree to ~1e-6, not to the bit: a rollout in contact can turn that into another contact history (unilateral contacts
    and Coulomb friction are not continuous), which moves a handful of the K rollouts -- and their weights -- by far more
    than the rounding that caused it.  First commands, where the inputs are identical, are compared exactly."""
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, f"{err_msg}: shapes {a.shape} vs {b.shape}"
    if a.size == 0:
        return
    bad = ~np.isclose(a, b, rtol=rtol, atol=atol)
    allowed = 0 if frac <= 0.0 else max(1, int(frac * bad.size))      # (frac = 0: an exact comparison, no outlier allowed)
    assert bad.sum() <= allowed, f"{err_msg}: {int(bad.sum())} of {bad.size} elements differ (allowed {allowed}), max |d| = {np.abs(a - b).max():.3g}"
    if cap is not None:
        assert np.abs(a - b).max() <= cap, f"{err_msg}: an outlier differs by {np.abs(a - b).max():.3g} > {cap}"
