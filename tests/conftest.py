import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_ops():
    return np.load(os.path.join(GOLDEN, 'ops.npz'))


@pytest.fixture(scope='session')
def golden_nets():
    return np.load(os.path.join(GOLDEN, 'nets.npz'))


@pytest.fixture(scope='session')
def golden_d2s():
    return np.load(os.path.join(GOLDEN, 'd2s_maps.npz'))


@pytest.fixture(scope='session', autouse=True)
def _build_oracle():
    """The oracle C library is test infrastructure: build it on demand."""
    so = os.path.join(ROOT, 'oracle', 'libsrx_oracle.so')
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')])
    yield
