"""Where an HDP-LPCM iteration (config C3) spends its time: device calls vs host updates.
    python profiles/hdp_phases.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import dynetlsm_amd as da                                  # noqa: E402
from dynetlsm_amd import hdp_updates as hu                 # noqa: E402
from dynetlsm_amd.synthetic import synthetic_lsm_network   # noqa: E402

T, N, D, K = 10, 2000, 2, 20
net = synthetic_lsm_network(T, N, D, density=0.03, seed=0)
rng = np.random.RandomState(0)
X = net['X_init'].copy()
z = rng.randint(0, K, size=(T, N)).astype(np.int64)
mu = rng.randn(K, D); sigma = np.ones(K); beta = np.ones(K) / K
weights = np.ones((T, K, K)) / K; lmbda = np.array([0.8])
hp = hu.HDPHyper(K, gamma=1.0, alpha_init=1.0, alpha=1.0, kappa=4.0, mean_variance_prior=2.0,
                 a=2.0, lambda_prior=0.9, lambda_variance_prior=0.01, gamma_prior_shape=1,
                 gamma_prior_rate=1, alpha_init_shape=1, alpha_init_rate=1,
                 alpha_kappa_shape=1, alpha_kappa_rate=1)
hp.b = 1.0
c = da.Chain(T, N, D, 'undirected', seed=1)
c.upload_network(net['Y']); c.set_positions(X); c.set_intercepts([net['intercept']])
c.set_samplers(da.SamplerGrid(T, N, 0.1, tune=None))
ph = {}


def tick(name, t0):
    c.synchronize()
    ph[name] = ph.get(name, 0.0) + time.perf_counter() - t0


n_it = 40
for it in range(1, n_it + 1):
    t0 = time.perf_counter(); c.set_prior_mixture(mu, sigma, lmbda, z if it == 1 else None); tick('set_prior', t0)
    t0 = time.perf_counter(); c.sweep_positions(it, 0); tick('sweep', t0)
    t0 = time.perf_counter(); c.center(); tick('center', t0)
    t0 = time.perf_counter(); c.loglik_full([[0.1], [0.2]]); tick('loglik x1', t0)
    t0 = time.perf_counter(); z, n, nk = c.sample_labels(it, weights); tick('labels', t0)
    t0 = time.perf_counter(); X = c.get_positions(); tick('get_positions', t0)
    t0 = time.perf_counter()
    mu, sigma, weights = mu.copy(), sigma.copy(), weights.copy()
    sums = hu.DeviceLabelSums(c)
    beta, lmbda = hu.gibbs_updates(sums, n, nk, mu, sigma, beta, weights, lmbda, hp, rng)
    tick('gibbs_updates (host draws + 3 device sums)', t0)
    t0 = time.perf_counter()
    hu.log_posterior_terms(sums, np.array([0.1]), np.array([0.1]), 2.0, mu, sigma, weights, beta,
                           lmbda, hp)
    tick('log_posterior (host + 1 device sum)', t0)
tot = sum(ph.values())
for k, v in ph.items():
    print('%-44s %7.3f ms' % (k, 1e3 * v / n_it))
print('%-44s %7.3f ms' % ('total', 1e3 * tot / n_it))
