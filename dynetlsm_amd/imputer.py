"""Initial imputation of missing dyads (coded -1), restating imputer.py:11-81.

What the reference does with missing dyads (lsm.py:345-359, :525-545; hdp_lpcm.py:669-706,
:1025-1049): impute them once before the chain - at random with the slice's observed
density for ``strategy='random'`` - then, every iteration, draw Bernoulli(expit(eta)) for
them; those draws are assigned to a temporary (``Y_new[idx][mask] = ...``), so the network the
chain conditions on never changes; the HDP-LPCM averages the draws after burn-in into
``missings_`` (the held-out edge probabilities).  The engine follows that: the packed
network is the imputed one, ``missings_`` is accumulated by the estimator on the host.
"""
import numpy as np

__all__ = ['SimpleNetworkImputer']


class SimpleNetworkImputer(object):
    def __init__(self, missing_value=-1, strategy='most_frequent', random_state=123, copy=True):
        self.missing_value = missing_value
        self.strategy = strategy
        self.copy = copy
        self.random_state = random_state

    def fit(self, Y):
        if self.strategy not in ('most_frequent', 'random'):
            raise ValueError("Can only use these strategies: {0}  got strategy='{1}".format(
                {'most_frequent', 'random'}, self.strategy))
        Y = np.array(Y, dtype=np.float64, copy=self.copy)
        T, N = Y.shape[:2]
        self.statistics_ = np.empty(T)
        for t in range(T):
            nan_mask = Y[t] == self.missing_value
            if not np.any(nan_mask):
                self.statistics_[t] = 0.0
            elif self.strategy == 'most_frequent':
                vals, counts = np.unique(Y[t][~nan_mask].ravel(), return_counts=True)
                self.statistics_[t] = vals[np.argmax(counts)]
            else:
                self.statistics_[t] = Y[t][~nan_mask].sum() / (N * (N - 1))
        return self

    def transform(self, Y):
        Y = np.array(Y, dtype=np.float64, copy=self.copy)
        if Y.shape[0] != self.statistics_.shape[0]:
            raise ValueError("Y has %d time steps, expected %d"
                             % (Y.shape[0], self.statistics_.shape[0]))
        rng = (self.random_state if isinstance(self.random_state, np.random.RandomState)
               else np.random.RandomState(self.random_state))
        for t in range(Y.shape[0]):
            if self.strategy == 'random':
                # the upper triangle is imputed and mirrored (imputer.py:67-77): for a directed
                # network this also symmetrises the slice, as in the reference
                iu = np.triu_indices(Y.shape[1], k=1)
                y_vec = Y[t][iu]
                nan_mask = y_vec == self.missing_value
                y_vec[nan_mask] = rng.choice([0, 1], p=[1 - self.statistics_[t], self.statistics_[t]],
                                             size=np.sum(nan_mask))
                Y[t][iu] = y_vec
                Y[t][np.tril_indices(Y.shape[1], k=-1)] = 0
                Y[t] += Y[t].T
            else:
                Y[t][Y[t] == self.missing_value] = self.statistics_[t]
        return Y

    def fit_transform(self, Y):
        return self.fit(Y).transform(Y)
