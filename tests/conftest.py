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


class _Merged:
    """several .npz files with disjoint keys read as one"""

    def __init__(self, *names):
        self.files = [load_golden(n) for n in names]

    def __getitem__(self, key):
        for f in self.files:
            if key in f.files:
                return f[key]
        raise KeyError(key)


# tags of the likelihood fixtures: a, b (n_features 2), c (3) and, from `make_golden.py wide`, e (5), f (8), g (6)
LIK_TAGS = ['a', 'b', 'c', 'e', 'f', 'g']


@pytest.fixture(scope='session')
def golden_lik():
    return _Merged('likelihoods.npz', 'wide_likelihoods.npz')


# the sweep and fit fixtures exist at n_features = 2 and at 5 (wide_*.npz: same generators, same keys)
@pytest.fixture(scope='session', params=['sweeps.npz', 'wide_sweeps.npz'], ids=['d2', 'd5'])
def golden_sweeps(request):
    return load_golden(request.param)


@pytest.fixture(scope='session', params=['fit_traces.npz', 'wide_fit_traces.npz'], ids=['d2', 'd5'])
def golden_fits(request):
    return load_golden(request.param)


@pytest.fixture(scope='session')
def monks():
    return load_golden('monks.npz')


# initialisation cases (tag, directed, n_features): init.npz and, from `make_golden.py wide`, wide_init.npz
INIT_CASES = [('u', False, 2), ('d', True, 2), ('u3', False, 3), ('u5', False, 5), ('d6', True, 6), ('u8', False, 8)]


@pytest.fixture(scope='session')
def golden_init():
    return _Merged('init.npz', 'wide_init.npz')
