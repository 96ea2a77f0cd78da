"""Wall-clock of the non-headline BASELINE.json configs through the estimator
facades (these are parity-test cases, not bench lines; the numbers go into
DESIGN.md).  Starting values are supplied so only the Gibbs loop is timed.

    python profiles/run_configs.py [c1] [c3] [c4]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import dynetlsm_amd as da                                  # noqa: E402
from dynetlsm_amd.synthetic import synthetic_lsm_network   # noqa: E402


def c1():
    Y = np.load(os.path.join(ROOT, 'tests', 'golden', 'monks.npz'))['Y_undirected']
    m = da.DynamicNetworkLSM(n_iter=500, tune=None, burn=None, random_state=42)
    t0 = time.perf_counter()
    m.fit(Y)
    dt = time.perf_counter() - t0
    return dict(config='C1 LSM monks T=3 N=18, 500 it', seconds=round(dt, 3),
                loop_seconds=round(m.loop_seconds_, 4),
                it_per_s=round(499 / m.loop_seconds_, 1))


def c3(n_iter=30):
    net = synthetic_lsm_network(10, 2000, 2, density=0.03, seed=0)
    rng = np.random.RandomState(0)
    K = 20
    init = dict(X=net['X_init'], intercept=[net['intercept']])
    m = da.DynamicNetworkHDPLPCM(n_iter=n_iter, tune=None, burn=None, n_components=K,
                                 random_state=1)
    t0 = time.perf_counter()
    m.fit(net['Y'], init=init)
    dt = time.perf_counter() - t0
    return dict(config='C3 HDP-LPCM T=10 N=2000 K=20, %d it' % n_iter,
                seconds=round(dt, 3), loop_seconds=round(m.loop_seconds_, 3),
                it_per_s=round((n_iter - 1) / m.loop_seconds_, 2),
                n_clusters_used=int(len(np.unique(m.z_))))


def c4(n_iter=100):
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from test_gpu_full_size import _sparse_directed
    T, N = 5, 10000
    X, radii, degree, in_edges, out_edges = _sparse_directed(T, N, 20, 0)
    # the facade takes a dense network; build it from the edge lists (4 GB float64)
    Y = np.zeros((T, N, N))
    for t in range(T):
        for i in range(N):
            Y[t, i, out_edges[t, i, :degree[t, i, 1]]] = 1.0
    m = da.DynamicNetworkLSM(n_iter=n_iter, tune=None, burn=None, is_directed=True,
                             n_control=100, n_resample_control=100, tau_sq=1e-4,
                             sigma_sq=1e-5, step_size_X=0.002, random_state=3)
    t0 = time.perf_counter()
    m.fit(Y, init=dict(X=X, intercept=[1.0, 0.5], radii=radii))
    dt = time.perf_counter() - t0
    return dict(config='C4 directed case-control T=5 N=10000 n_control=100, %d it' % n_iter,
                seconds=round(dt, 3), loop_seconds=round(m.loop_seconds_, 3),
                it_per_s=round((n_iter - 1) / m.loop_seconds_, 2))


def d2(n_iter=40):
    """not a BASELINE config: the exact directed model at C2's size (radii / intercept MH
    host driven, sweep on the device), both sweep algorithms"""
    net = synthetic_lsm_network(10, 2000, 2, density=0.03, seed=0, directed=True)
    N = 2000
    radii = np.full(N, 1.0 / N)
    out = {}
    for algo in (3, 4):
        m = da.DynamicNetworkLSM(n_iter=n_iter, tune=None, burn=None, is_directed=True,
                                 tau_sq=2.0, sigma_sq=0.1, step_size_X=0.0005, random_state=3,
                                 sweep_algo=algo)
        m.fit(net['Y'], init=dict(X=net['X_init'] / N, intercept=[1.0, 1.0], radii=radii))
        out['algo%d_it_per_s' % algo] = round((n_iter - 1) / m.loop_seconds_, 2)
    return dict(config='directed exact T=10 N=2000, %d it' % n_iter, **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['c1', 'c3', 'c4']
    for w in which:
        print(json.dumps(globals()[w]()), flush=True)
