"""Synthetic dynamic networks for the benchmark configurations.

Own generator (numpy on the host): a Gaussian random walk of latent positions
and Bernoulli dyads with logit  b - ||X_ti - X_tj||  (the model of
DynamicNetworkLSM); the reference's generator is O(T N^2) Python-side.
"""
import numpy as np

__all__ = ['synthetic_lsm_network']


def _expit(x):
    return 1.0 / (1.0 + np.exp(-x))


def _pairwise(X):
    sq = (X * X).sum(1)
    d2 = sq[:, None] + sq[None, :] - 2.0 * X.dot(X.T)
    np.maximum(d2, 0.0, out=d2)
    return np.sqrt(d2)


def synthetic_lsm_network(T=10, N=2000, D=2, density=0.03, seed=0, directed=False,
                          x0_scale=1.5, walk_scale=0.3, init_noise=0.1):
    """Returns dict(Y, X_true, intercept, X_init).

    X[0] ~ N(0, x0_scale^2 I), X[t] = X[t-1] + N(0, walk_scale^2 I), centred;
    the intercept is solved so that the expected density at t = 0 is
    ``density``; Y_tij ~ Bernoulli(expit(b - d_tij)), symmetrised unless
    ``directed``, zero diagonal, float64 as fit(Y) expects.  ``X_init`` = truth +
    N(0, init_noise^2) is where the timing runs start the chain (the
    initialisation pipeline is outside the hot path)."""
    rng = np.random.RandomState(seed)
    X = np.zeros((T, N, D))
    X[0] = x0_scale * rng.randn(N, D)
    for t in range(1, T):
        X[t] = X[t - 1] + walk_scale * rng.randn(N, D)
    X -= X.mean(axis=(0, 1))
    d0 = _pairwise(X[0])
    iu = np.triu_indices(N, 1)
    lo, hi = -20.0, 20.0
    for _ in range(60):
        b = 0.5 * (lo + hi)
        if _expit(b - d0[iu]).mean() < density:
            lo = b
        else:
            hi = b
    b = 0.5 * (lo + hi)
    Y = np.zeros((T, N, N))
    for t in range(T):
        P = _expit(b - _pairwise(X[t]))
        U = rng.rand(N, N)
        A = (U < P).astype(np.float64)
        np.fill_diagonal(A, 0.0)
        if not directed:
            A = np.triu(A, 1)
            A = A + A.T
        Y[t] = A
    X_init = X + init_noise * rng.randn(T, N, D)
    return dict(Y=Y, X_true=X, intercept=float(b), X_init=X_init)
