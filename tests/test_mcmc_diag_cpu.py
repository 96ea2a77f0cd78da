"""the diagnostics the full-size posterior tests lean on, against processes with known answers"""
import numpy as np

from mcmc_diag import autocorr, effective_n, mcse, split_rhat, pooled_mean_and_se


def _ar1(n, phi, rng, mean=0.0):
    x = np.empty(n)
    x[0] = rng.randn() / np.sqrt(1 - phi * phi)
    e = rng.randn(n)
    for i in range(1, n):
        x[i] = phi * x[i - 1] + e[i]
    return x + mean


def test_effective_n_of_an_ar1_process():
    rng = np.random.RandomState(0)
    n, phi = 20000, 0.8
    x = _ar1(n, phi, rng)
    want = n * (1 - phi) / (1 + phi)
    assert abs(effective_n(x) / want - 1) < 0.15
    np.testing.assert_allclose(autocorr(x, 3), [phi, phi ** 2, phi ** 3], atol=0.03)
    # white noise: about n
    assert effective_n(rng.randn(n)) > 0.8 * n
    # the standard error covers the true mean
    hits = 0
    for s in range(40):
        y = _ar1(4000, 0.7, np.random.RandomState(100 + s), mean=3.0)
        hits += abs(y.mean() - 3.0) < 2 * mcse(y)
    assert hits >= 32               # ~95 % nominal


def test_split_rhat_separates_mixed_from_stuck_chains():
    rng = np.random.RandomState(1)
    good = np.stack([_ar1(3000, 0.5, rng) for _ in range(2)])
    assert split_rhat(good) < 1.02
    shifted = good.copy(); shifted[1] += 2.0
    assert split_rhat(shifted) > 1.3
    drifting = good.copy(); drifting += np.linspace(0, 3, 3000)
    assert split_rhat(drifting) > 1.1
    m, se = pooled_mean_and_se(good)
    assert abs(m) < 4 * se
