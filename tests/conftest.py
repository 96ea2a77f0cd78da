import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope='session')
def golden_lik():
    return load_golden('likelihoods.npz')


@pytest.fixture(scope='session')
def golden_sweeps():
    return load_golden('sweeps.npz')


@pytest.fixture(scope='session')
def golden_fits():
    return load_golden('fit_traces.npz')


@pytest.fixture(scope='session')
def monks():
    return load_golden('monks.npz')


@pytest.fixture(scope='session')
def golden_init():
    return load_golden('init.npz')
