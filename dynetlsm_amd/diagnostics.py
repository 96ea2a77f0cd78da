"""Convergence diagnostics of the stored traces, as the reference reports them after the
HDP-LPCM loop (hdp_lpcm.py:1165-1176 -> trace_utils.py): Geweke's z-score with the variance
of each window's mean taken from the spectral density at frequency zero of an AR(p) fit
(Yule-Walker, order by AIC), and the effective sample size from the autocorrelations.

Host-side numpy on traces of a few thousand scalars; nothing here touches the device.
The reference delegates the Yule-Walker solve to statsmodels (``method='adjusted'``), which is
not installed here: the solve below follows that estimator's published definition
(autocovariances with the n - k denominator, Toeplitz system), parity unpinned.
"""
from math import ceil, floor

import numpy as np
from scipy.linalg import solve_toeplitz
from scipy.stats import norm

__all__ = ['yule_walker_adjusted', 'spectrum0_ar', 'geweke_z', 'geweke_diag', 'effective_n']


def _autocov_adjusted(x, max_lag):
    """r[k] = sum_t x_t x_{t+k} / (n - k), k = 0 .. max_lag, of the de-meaned series"""
    n = x.shape[0]
    x = x - x.mean()
    full = np.correlate(x, x, mode='full')[n - 1:n + max_lag]
    return full / (n - np.arange(max_lag + 1))


def yule_walker_adjusted(x, order, acov=None):
    """AR(order) coefficients and innovation standard deviation from the Yule-Walker equations
    on the adjusted autocovariances (statsmodels.regression.linear_model.yule_walker with
    demean=True, method='adjusted', as trace_utils.py:71 calls it)."""
    r = _autocov_adjusted(np.asarray(x, dtype=np.float64), order) if acov is None else acov
    rho = solve_toeplitz(r[:order], r[1:order + 1])
    sigma_sq = r[0] - np.dot(r[1:order + 1], rho)
    return rho, np.sqrt(sigma_sq)


def spectrum0_ar(x, max_order='auto'):
    """trace_utils.py:57-79: (f(0) / n, chosen order) of the AR(p) fit minimising
    2 n log(sigma) + 2 (p + 1) over p = 1 .. max_order (auto: floor(10 log10 n))."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    if np.allclose(np.var(x), 0.0):
        return 0., 0.
    if max_order == 'auto':
        max_order = floor(10 * np.log10(n))
    acov = _autocov_adjusted(x, max_order)
    best = None
    for p in range(1, max_order + 1):
        coefs, sigma = yule_walker_adjusted(x, p, acov=acov)
        aic = 2 * n * np.log(sigma) + 2 * (p + 1)
        f0 = sigma ** 2 / (1 - np.sum(coefs)) ** 2
        if best is None or aic < best[0]:          # first minimum, as np.argmin
            best = (aic, f0, p)
    return best[1] / n, float(best[2])


def geweke_z(x, first=0.1, last=0.5):
    """trace_utils.py:82-99: (mean of the first 10 % - mean of the last 50 %) over the root of
    the two spectral variance estimates"""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    head = x[:ceil(first * n)]
    tail = x[n - floor(last * n):]
    v1, _ = spectrum0_ar(head)
    v2, _ = spectrum0_ar(tail)
    return (head.mean() - tail.mean()) / np.sqrt(v1 + v2)


def geweke_diag(x, first=0.1, last=0.5, n_burn=None):
    """trace_utils.py:102-115: (z-score, two-sided p-value) of the trace after burn-in"""
    x = np.asarray(x, dtype=np.float64)
    if n_burn is not None:
        x = x[n_burn:]
    z = geweke_z(x, first=first, last=last)
    return z, 2 * (1 - norm.cdf(np.abs(z)))


def effective_n(x, maxlags=100):
    """trace_utils.py:38-45: n / (1 + 2 sum of the first ``maxlags`` autocorrelations)"""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    if maxlags >= n or maxlags < 1:
        raise ValueError('maxlags must be strictly positive < %d' % n)
    d = x - x.mean()
    corr = np.correlate(d, d, mode='full')[n:n + maxlags] / np.dot(d, d)
    return n / (1 + 2 * np.sum(corr))
