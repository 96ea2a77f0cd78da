"""Initial imputation of missing dyads (coded -1).

What the reference does with missing dyads (imputer.py:11-81; lsm.py:345-359, :525-545;
hdp_lpcm.py:669-706, :1025-1049): it fills them in once before the chain - at random with the
slice's observed density for ``strategy='random'`` - then, every iteration, draws
Bernoulli(expit(eta)) for them; those draws are assigned to a temporary
(``Y_new[idx][mask] = ...``), so the network the chain conditions on never changes; the
HDP-LPCM averages the draws after burn-in into ``missings_`` (the held-out edge
probabilities).  The engine follows that: the packed network is the imputed one,
``missings_`` is accumulated by the estimator on the host.

``impute_missing_dyads`` reproduces the reference's filled-in network bit for bit (same
per-slice statistic, same ``RandomState(123).choice`` calls in slice order);
``SimpleNetworkImputer`` is the reference's class name over it.
"""
import numpy as np

__all__ = ['impute_missing_dyads', 'SimpleNetworkImputer']


def _slice_fill_value(Yt, observed, strategy):
    if observed.all():
        return 0.0
    if strategy == 'random':           # observed density, over all ordered pairs
        n = Yt.shape[0]
        return Yt[observed].sum() / (n * (n - 1))
    values, freq = np.unique(Yt[observed], return_counts=True)
    return values[np.argmax(freq)]     # most frequent observed value


def impute_missing_dyads(Y, strategy='most_frequent', missing_value=-1, random_state=123):
    """Returns (Y_imputed, per-slice fill statistics); ``Y`` (T, N, N) is not modified.

    'random': the upper triangle of every slice is completed with Bernoulli(density) draws
    and mirrored - a directed slice comes out symmetrised, as in the reference;
    'most_frequent': every missing entry gets the slice's most frequent observed value."""
    if strategy not in ('most_frequent', 'random'):
        raise ValueError("Can only use these strategies: {0}  got strategy='{1}".format(
            {'most_frequent', 'random'}, strategy))
    out = np.array(Y, dtype=np.float64)
    T, N = out.shape[:2]
    rng = (random_state if isinstance(random_state, np.random.RandomState)
           else np.random.RandomState(random_state))
    stats = np.array([_slice_fill_value(out[t], out[t] != missing_value, strategy)
                      for t in range(T)])
    upper = np.triu_indices(N, k=1)
    for t, p in enumerate(stats):
        if strategy == 'most_frequent':
            out[t][out[t] == missing_value] = p
            continue
        half = out[t][upper]
        holes = half == missing_value
        half[holes] = rng.choice([0, 1], p=[1 - p, p], size=int(holes.sum()))
        filled = np.zeros((N, N))
        filled[upper] = half
        out[t] = filled + filled.T
    return out, stats


class SimpleNetworkImputer(object):
    """the reference's estimator-style wrapper (imputer.py:11-81)"""

    def __init__(self, missing_value=-1, strategy='most_frequent', random_state=123, copy=True):
        self.missing_value, self.strategy = missing_value, strategy
        self.random_state, self.copy = random_state, copy

    def fit_transform(self, Y):
        out, self.statistics_ = impute_missing_dyads(Y, self.strategy, self.missing_value,
                                                     self.random_state)
        return out
