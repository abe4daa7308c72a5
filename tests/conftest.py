import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'cc_reference.npz'))


@pytest.fixture(scope='session')
def ctx():
    """GPU context.  Fails loudly (no skip, no CPU fallback) when the HIP
    library or the GPU is missing."""
    import xcorr_amd as X
    return X.Context(0)
