"""Duration of the label block update's launches (k_sample_labels + k_label_counts) at C3's
size, by the kernel-attached events.
    python profiles/labels_timing.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dynetlsm_amd as da                                  # noqa: E402
from dynetlsm_amd.synthetic import synthetic_lsm_network   # noqa: E402

T, N, D, K = 10, 2000, 2, 20
net = synthetic_lsm_network(T, N, D, density=0.03, seed=0)
rng = np.random.RandomState(0)
z = rng.randint(0, K, size=(T, N)).astype(np.int64)
mu = rng.randn(K, D)
sigma = np.ones(K)
weights = np.ones((T, K, K)) / K
c = da.Chain(T, N, D, 'undirected', seed=1)
c.upload_network(net['Y'])
c.set_positions(net['X_init'])
c.set_intercepts([net['intercept']])
c.set_prior_mixture(mu, sigma, 0.8, z)
for it in range(5):
    c.sample_labels(it, weights)
c.synchronize()
c.profile_enable(True)
for it in range(50):
    c.sample_labels(it, weights)
ms, n = c.profile_read(3)
print('label update: %.1f us per call (%d calls)' % (1e3 * ms / n, n))
