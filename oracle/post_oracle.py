"""CPU restatement of the post-loop processing of DynamicNetworkHDPLPCM.

TEST INFRASTRUCTURE ONLY.  Pinned by tests/golden/post.npz (the reference's
label_utils / model_selection functions run on a synthetic stored trace,
tests/golden/make_golden.py post).
"""
import numpy as np

__all__ = ['posterior_cooccurrence', 'posterior_expected_vi', 'time_averaged_expected_vi',
           'minimize_expected_vi', 'cluster_counts', 'cluster_counts_t',
           'latent_marginal_loglikelihood']


def posterior_cooccurrence(zs, n_burn, K):
    """label_utils.py:40-62 for every t: (T, N, N) fraction of the kept samples in which
    nodes i and j share a label"""
    S, T, N = zs.shape
    out = np.zeros((T, N, N))
    eye = np.eye(K)
    for t in range(T):
        for z in zs[n_burn:, t]:
            ind = eye[z]
            out[t] += ind.dot(ind.T)
        out[t] /= (S - n_burn)
    return out


def posterior_expected_vi(labels, cooc):
    """model_selection/posterior_vi.py:10-20 (the non-vectorised definition)"""
    n = labels.shape[0]
    vi = 0.0
    for i in range(n):
        ind = labels == labels[i]
        vi += np.log2(np.sum(ind))
        vi -= 2 * np.log2(np.sum(ind * cooc[i, :]))
        vi += np.log2(np.sum(cooc[i, :]))
    return vi / n


def time_averaged_expected_vi(labels, cooc):
    """posterior_vi.py:45-52"""
    return sum(posterior_expected_vi(labels[t], cooc[t]) for t in range(labels.shape[0])) / \
        labels.shape[0]


def minimize_expected_vi(zs, n_burn, cooc, loglik_of_sample):
    """posterior_vi.py:55-82: argmin over the kept samples; ties go to the sample with the
    highest network log-likelihood (first one on equality)"""
    ids = np.arange(n_burn, zs.shape[0])
    vis = np.array([time_averaged_expected_vi(zs[i], cooc) for i in ids])
    mins = np.where(vis == vis.min())[0]
    if mins.shape[0] > 1:
        best, best_ll = None, -np.inf
        for m in mins:
            ll = loglik_of_sample(ids[m])
            if ll > best_ll:
                best, best_ll = ids[m], ll
        return best, vis
    return ids[mins[0]], vis


def cluster_counts(zs, n_burn):
    """approx_bic.py:40-51"""
    return np.array([np.unique(z.ravel()).shape[0] for z in zs[n_burn:]])


def cluster_counts_t(zs, n_burn):
    """approx_bic.py:26-37"""
    return np.array([[np.unique(z[t]).shape[0] for z in zs[n_burn:]]
                     for t in range(zs.shape[1])])


def latent_marginal_loglikelihood(X, init_w, trans_w, mu, sigma, lmbda):
    """approx_bic.py:54-76: forward algorithm over the label chain of every node"""
    T, N, D = X.shape
    ll = 0.0
    for i in range(N):
        g = np.empty((T, sigma.shape[0]))
        for t in range(T):
            m = mu if t == 0 else lmbda * mu + (1 - lmbda) * X[t - 1, i]
            g[t] = np.exp(-0.5 * D * np.log(2 * np.pi * sigma) -
                          0.5 * np.sum((X[t, i] - m) ** 2, axis=1) / sigma)
        f = init_w * g[0]
        c = f.sum(); ll += np.log(c); f = f / c
        for t in range(1, T):
            f = g[t] * trans_w[t].T.dot(f)
            c = f.sum(); ll += np.log(c); f = f / c
    return ll
