"""CPU restatement of the forecast accumulations.

TEST INFRASTRUCTURE ONLY.  Pinned by tests/golden/forecast.npz (the reference's
forecast properties and forecast.pyx:marginal_forecast run on a synthetic stored trace).
"""
import numpy as np

__all__ = ['mean_probas', 'marginal_forecast']


def _dist(X):
    d = X[:, None, :] - X[None, :, :]
    return np.sqrt((d * d).sum(-1))


def mean_probas(Xs, intercepts, zero_diag=False):
    """(1/S) sum_s expit(b_s - dist(Xs[s])) (hdp_lpcm.py:583-587, :621-624)"""
    S = Xs.shape[0]
    out = np.zeros((Xs.shape[1], Xs.shape[1]))
    for s in range(S):
        out += 1.0 / (1.0 + np.exp(-(intercepts[s] - _dist(Xs[s])))) / S
    if zero_diag:
        np.fill_diagonal(out, 0.0)
    return out


def marginal_forecast(x, x_prev, z, trans_weights, mus, sigmas, intercepts, lmbdas):
    """forecast.pyx:79-128 with renormalize=False, pair loops vectorised"""
    S, N = x_prev.shape[:2]
    D = x.shape[1]
    d = _dist(x)
    num = np.zeros((N, N))
    den = np.zeros((N, N))
    for s in range(S):
        w = np.zeros(N)
        for i in range(N):
            for k in range(sigmas.shape[1]):
                m = lmbdas[s] * mus[s, k] + (1 - lmbdas[s]) * x_prev[s, i]
                pdf = np.exp(-0.5 * D * np.log(2 * np.pi * sigmas[s, k]) -
                             0.5 * np.sum((x[i] - m) ** 2) / sigmas[s, k])
                w[i] += trans_weights[s, z[s, i], k] * pdf
        ww = np.outer(w, w)
        num += ww / (1.0 + np.exp(-(intercepts[s] - d))) / S
        den += ww / S
    np.fill_diagonal(num, 0.0)
    np.fill_diagonal(den, 1.0)
    return num / den
