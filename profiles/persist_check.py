"""Persistent sweep (algo 7) against the launch-per-batch form (algo 4): the two run the same
items and the same fixed-point solves, so positions and sampler state must agree BIT FOR BIT,
sweep after sweep, at every size - a stale read of a handed-off byte shows up as a difference.

    python profiles/persist_check.py [n_sweeps_at_C2]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dynetlsm_amd import Chain, SamplerGrid          # noqa: E402


def net(seed, T, N, D, density=0.2, directed=False):
    rng = np.random.RandomState(seed)
    X = rng.randn(T, N, D)
    Y = (rng.rand(T, N, N) < density).astype(np.float64)
    for t in range(T):
        np.fill_diagonal(Y[t], 0.0)
    if not directed:
        Y = np.triu(Y, 1)
        Y = Y + Y.transpose(0, 2, 1)
    return X, Y, rng.dirichlet(np.ones(N))


def run(algo, T, N, D, model, prior, n_sweeps, seed=5, tune=5, X=None, Y=None):
    if X is None:
        X, Y, radii = net(seed, T, N, D, directed=model == 'directed')
    else:
        radii = np.random.RandomState(seed).dirichlet(np.ones(N))
    rng = np.random.RandomState(seed + 1)
    K = 3
    mu = rng.randn(K, D); sigma = rng.uniform(0.5, 1.5, K)
    z = rng.randint(0, K, size=(T, N)).astype(np.int64)
    g = SamplerGrid(T, N, 0.2, tune=tune, tune_interval=2)
    out = []
    with Chain(T, N, D, model, seed=0xC0FFEE, chain_id=1) as c:
        c.upload_network(Y); c.set_positions(X)
        c.set_intercepts([0.5] if model == 'undirected' else [0.3, 0.7])
        if model != 'undirected':
            c.set_radii(radii)
        if prior == 'mix':
            c.set_prior_mixture(mu, sigma, 0.8, z)
        else:
            c.set_prior_random_walk(2.0, 0.1)
        c.set_samplers(g)
        t0 = time.time()
        for it in range(1, n_sweeps + 1):
            c.sweep_positions(it, algo=algo)
            if it <= 3 or it == n_sweeps or it % 16 == 0:
                out.append(c.get_positions().copy())
        dt = time.time() - t0
        c.get_samplers(g)
    return out, g, dt


def same(a, b):
    for x, y in zip(a[0], b[0]):
        if not np.array_equal(x, y):
            return False, float(np.abs(x - y).max())
    ok = (np.array_equal(a[1].step_size, b[1].step_size) and
          np.array_equal(a[1].n_accepted, b[1].n_accepted) and
          np.array_equal(a[1].n_steps, b[1].n_steps))
    return ok, 0.0


def main():
    nC2 = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    cases = [(4, 300, 2, 'undirected', 'rw', 4), (3, 1100, 2, 'undirected', 'mix', 3),
             (1, 770, 2, 'undirected', 'rw', 3), (5, 129, 1, 'undirected', 'mix', 3),
             (4, 260, 2, 'directed', 'rw', 3), (2, 257, 4, 'undirected', 'rw', 3),
             (3, 128, 3, 'directed', 'mix', 3), (37, 140, 2, 'undirected', 'mix', 2),
             (4, 10, 2, 'undirected', 'rw', 3)]
    bad = 0
    for T, N, D, model, prior, ns in cases:
        a = run(4, T, N, D, model, prior, ns)
        b = run(7, T, N, D, model, prior, ns)
        ok, err = same(a, b)
        print('T=%d N=%d D=%d %s %s: %s (max diff %.3g) algo4 %.1f ms algo7 %.1f ms' %
              (T, N, D, model, prior, 'bitwise equal' if ok else 'DIFFERENT', err,
               1e3 * a[2], 1e3 * b[2]), flush=True)
        bad += not ok
    # C2: many sweeps, no tuning (steady acceptance), bitwise
    X, Y, _ = net(3, 10, 2000, 2, density=0.03)
    a = run(4, 10, 2000, 2, 'undirected', 'rw', nC2, tune=None, X=X, Y=Y)
    b = run(7, 10, 2000, 2, 'undirected', 'rw', nC2, tune=None, X=X, Y=Y)
    ok, err = same(a, b)
    print('C2 size, %d sweeps: %s (max diff %.3g); host-synchronised sweeps %.3f / %.3f ms' %
          (nC2, 'bitwise equal' if ok else 'DIFFERENT', err, 1e3 * a[2] / nC2, 1e3 * b[2] / nC2), flush=True)
    bad += not ok
    # the failure path: no poll budget -> the launch gives up, the host reports it, nothing hangs
    os.environ['DLSM_PERSIST_BUDGET'] = '0'
    try:
        run(7, 4, 600, 2, 'undirected', 'rw', 1)
        print('budget 0: NO ERROR REPORTED')
        bad += 1
    except Exception as e:          # noqa: BLE001
        print('budget 0: reported:', str(e)[:100])
    del os.environ['DLSM_PERSIST_BUDGET']
    print('FAILED' if bad else 'ALL OK')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
