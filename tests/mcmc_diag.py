"""Chain diagnostics for the posterior-parity tests: effective sample size from the
autocorrelations (the estimator of the reference's trace_utils.py:11-45: normalised
autocorrelations up to lag 100, n / (1 + 2 sum rho)), Monte Carlo standard errors from it and
the split potential-scale-reduction factor over several chains."""
import numpy as np


def autocorr(x, maxlags=100):
    """trace_utils.py:11-36 (xcorr of x with itself, mean removed, normed) for lags 1..maxlags"""
    x = np.asarray(x, dtype=np.float64)
    x = x - x.mean()
    n = x.shape[0]
    maxlags = min(maxlags, n - 1)
    full = np.correlate(x, x, mode='full')
    return full[n:n + maxlags] / np.dot(x, x)


def effective_n(x, maxlags=100):
    """trace_utils.py:39-45; floored at 1 and capped at n (a negative sum of a short noisy trace
    would claim more than n draws)"""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    if np.var(x) == 0.0:
        return float(n)
    ess = n / (1.0 + 2.0 * autocorr(x, maxlags).sum())
    return float(min(max(ess, 1.0), n))


def mcse(x, maxlags=100):
    """Monte Carlo standard error of the mean of x"""
    x = np.asarray(x, dtype=np.float64)
    return float(x.std(ddof=1) / np.sqrt(effective_n(x, maxlags)))


def split_rhat(chains):
    """split-R-hat of Gelman et al. (BDA3 11.4): every chain cut in two halves, between- against
    within-sequence variance.  chains: (m, n)"""
    c = np.asarray(chains, dtype=np.float64)
    half = c.shape[1] // 2
    s = np.concatenate([c[:, :half], c[:, half:2 * half]], axis=0)
    n = s.shape[1]
    W = s.var(axis=1, ddof=1).mean()
    B = n * s.mean(axis=1).var(ddof=1)
    if W == 0.0:
        return 1.0
    return float(np.sqrt(((n - 1.0) / n * W + B / n) / W))


def rhat(chains):
    """the potential-scale-reduction factor WITHOUT splitting (Gelman & Rubin 1992): between- against
    within-chain variance of whole chains - it asks whether the chains sit in the same place; the split form
    above additionally asks every chain to be stationary over its own length"""
    c = np.asarray(chains, dtype=np.float64)
    n = c.shape[1]
    W = c.var(axis=1, ddof=1).mean()
    B = n * c.mean(axis=1).var(ddof=1)
    if W == 0.0:
        return 1.0
    return float(np.sqrt(((n - 1.0) / n * W + B / n) / W))


def pooled_mean_and_se(chains, maxlags=100):
    """mean over all chains and its Monte Carlo standard error (independent chains: the
    per-chain errors add in quadrature)"""
    c = np.asarray(chains, dtype=np.float64)
    se = np.sqrt(sum(mcse(x, maxlags) ** 2 for x in c)) / c.shape[0]
    return float(c.mean()), float(se)
