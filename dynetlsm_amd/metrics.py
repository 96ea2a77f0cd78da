"""Evaluation helpers of the reference (metrics.py): in-sample / out-of-sample AUC of the
edge probabilities and the variation of information between two partitions.  Host side:
O(T N^2 log) once per fitted model (scikit-learn's ``roc_auc_score``, as the reference)."""
import numpy as np

__all__ = ['network_auc', 'out_of_sample_auc', 'variation_of_information', 'FittedQuantities']


def _dyads(Y, is_directed):
    T, N, _ = Y.shape
    mask = ~np.eye(N, dtype=bool) if is_directed else np.triu(np.ones((N, N), dtype=bool), 1)
    return np.nonzero(np.broadcast_to(mask, (T, N, N)))


def network_auc(Y_true, Y_pred, is_directed=False, nan_mask=None):
    """metrics.py:10-24: AUC over the off-diagonal (directed) / upper-triangular dyads"""
    from sklearn.metrics import roc_auc_score
    idx = _dyads(Y_true, is_directed)
    y_fit, y_true = Y_pred[idx], Y_true[idx]
    if nan_mask is not None:
        y_fit, y_true = y_fit[~nan_mask], y_true[~nan_mask]
    return roc_auc_score(y_true, y_fit)


def out_of_sample_auc(y_true, y_pred, test_indices):
    """metrics.py:27-29"""
    from sklearn.metrics import roc_auc_score
    return roc_auc_score(y_true[_dyads(y_true, False)][test_indices], y_pred)


def variation_of_information(labels_true, labels_pred):
    """metrics.py:56-61"""
    from sklearn.metrics import mutual_info_score
    from sklearn.metrics.cluster import entropy
    return (entropy(labels_true) + entropy(labels_pred) -
            2 * mutual_info_score(labels_true, labels_pred))


class FittedQuantities(object):
    """``distances_``, ``probas_`` and ``auc_`` of a fitted estimator (lsm.py:283-320,
    hdp_lpcm.py:466-495, :632-639)."""

    @property
    def distances_(self):
        if not hasattr(self, 'X_'):
            raise ValueError('Model not fit.')
        X = self.X_
        sq = (X * X).sum(-1)
        d2 = sq[:, :, None] + sq[:, None, :] - 2 * np.einsum('tid,tjd->tij', X, X)
        return np.sqrt(np.maximum(d2, 0.0))

    @property
    def probas_(self):
        if not hasattr(self, 'X_'):
            raise ValueError('Model not fit.')
        d = self.distances_
        if self.is_directed:
            eta = (self.intercept_[0] * (1 - d / self.radii_[None, None, :]) +
                   self.intercept_[1] * (1 - d / self.radii_[None, :, None]))
        else:
            eta = np.ravel(self.intercept_)[0] - d
        p = 1 / (1 + np.exp(-eta))
        idx = np.arange(d.shape[1])
        p[:, idx, idx] = 0.0
        return p

    @property
    def auc_(self):
        """In-sample AUC of the selected model."""
        if not hasattr(self, 'X_'):
            raise ValueError('Model not fit.')
        return network_auc(self.Y_fit_, self.probas_, is_directed=self.is_directed,
                           nan_mask=getattr(self, 'nan_mask_', None))
