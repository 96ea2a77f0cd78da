"""Post-loop processing of a fitted DynamicNetworkHDPLPCM, mirroring
hdp_lpcm.py:1085-1176, label_utils.py and model_selection/: model selection by BIC / MAP
size / minimum posterior expected VI, weight renormalisation, Procrustes alignment of the
stored samples, posterior group counts.

The two O(n_samples T N^2) parts - posterior co-occurrence matrices and the expected-VI
criterion of every kept sample - run on the device (``Chain.post_*``), as do the network
log-likelihoods needed for BIC and for VI ties; the rest is O(n_samples T N) bookkeeping
on the host, as in the reference.
"""
from types import SimpleNamespace

import numpy as np

__all__ = ['cluster_counts', 'cluster_counts_t', 'renormalize_weights',
           'latent_marginal_loglikelihood', 'select_bic', 'posterior_cooccurrences',
           'expected_vi', 'minimize_posterior_expected_vi', 'procrustes_align_samples',
           'posterior_group_counts', 'select_model', 'select_model_device',
           'posterior_group_counts_from']


def _presence(zs, K):
    """(S, T, K) bool: label k used at time t of sample s"""
    S, T, N = zs.shape
    flat = (np.arange(S * T)[:, None] * K + zs.reshape(S * T, N)).ravel()
    return (np.bincount(flat, minlength=S * T * K) > 0).reshape(S, T, K)


def cluster_counts(zs, n_burn):
    """approx_bic.py:40-51: number of labels in use (over all times) per kept sample"""
    K = int(zs.max()) + 1
    return _presence(zs[n_burn:], K).any(axis=1).sum(axis=1)


def cluster_counts_t(zs, n_burn):
    """approx_bic.py:26-37: (T, S_kept)"""
    K = int(zs.max()) + 1
    return _presence(zs[n_burn:], K).sum(axis=2).T


def renormalize_weights(model, sample_id):
    """label_utils.py:10-37"""
    T, N = model.zs_.shape[1:]
    active = np.unique(model.zs_[sample_id].ravel())
    beta = model.betas_[sample_id, active]
    beta = beta / beta.sum()
    weights = model.weights_[sample_id]
    init_w = weights[0, 0, active]
    init_w = init_w / init_w.sum()
    trans_w = np.zeros((T, active.shape[0], active.shape[0]))
    for t in range(1, T):
        trans_w[t] = weights[t, active][:, active]
        trans_w[t] /= np.sum(trans_w[t], axis=1).reshape(-1, 1)
    _, z = np.unique(model.zs_[sample_id].ravel(), return_inverse=True)
    return (z.reshape(T, N), beta, init_w, trans_w, model.mus_[sample_id, active],
            model.sigmas_[sample_id, active])


def latent_marginal_loglikelihood(X, init_w, trans_w, mu, sigma, lmbda):
    """approx_bic.py:54-76, all nodes at once: forward algorithm over the label chain"""
    T, N, D = X.shape
    lmbda = float(np.ravel(lmbda)[0])

    def gauss(t):
        m = mu[None] if t == 0 else lmbda * mu[None] + (1 - lmbda) * X[t - 1][:, None, :]
        ss = np.sum((X[t][:, None, :] - m) ** 2, axis=2)
        return np.exp(-0.5 * D * np.log(2 * np.pi * sigma)[None] - 0.5 * ss / sigma[None])
    f = init_w[None] * gauss(0)
    c = f.sum(axis=1)
    ll = np.log(c).sum()
    f = f / c[:, None]
    for t in range(1, T):
        f = gauss(t) * f.dot(trans_w[t])
        c = f.sum(axis=1)
        ll += np.log(c).sum()
        f = f / c[:, None]
    return ll


def _network_loglik(model, chain, sample_id):
    chain.set_positions(model.Xs_[sample_id])
    if model.is_directed:
        chain.set_radii(model.radiis_[sample_id])
    return chain.loglik_full([model.intercepts_[sample_id]])[0]


def select_bic(model, chain, n_burn):
    """approx_bic.py:79-162: per model size present in the kept samples, the MAP sample
    and its BIC; returns (bic[n_sizes, 4], models, counts)"""
    Y = model.Y_fit_
    T, N, _ = Y.shape
    counts = cluster_counts(model.zs_, n_burn)
    bic, models = [], []
    for k in np.unique(counts):
        lp = np.where(counts == k, model.logps_[n_burn:], -np.inf)
        map_id = int(np.argmax(lp)) + n_burn
        z, beta, init_w, trans_w, mu, sigma = renormalize_weights(model, map_id)
        X, intercept, lmbda = model.Xs_[map_id], model.intercepts_[map_id], model.lambdas_[map_id]
        radii = model.radiis_[map_id] if model.is_directed else None
        loglik_k = _network_loglik(model, chain, map_id)
        bic_k = -2 * loglik_k
        off_diag_sum = np.sum(Y) - np.einsum('ikk', Y).sum()
        if model.is_directed:
            bic_k += (2 + N) * np.log(off_diag_sum)
        else:
            bic_k += np.log(0.5 * off_diag_sum)
        bic_k -= 2 * latent_marginal_loglikelihood(X, init_w, trans_w, mu, sigma, lmbda)
        n_params = ((model.n_features + 1) * k + (k - 1) + (k - 1) + (T - 1) * k * (k - 1))
        bic_k += n_params * np.log(N * T)
        models.append(SimpleNamespace(beta=beta, init_weights=init_w, trans_weights=trans_w, X=X,
                                      mu=mu, sigma=sigma, lmbda=lmbda, z=model.zs_[map_id],
                                      intercept=intercept, radii=radii))
        bic.append([k, bic_k, loglik_k, map_id])
    return np.array(bic), models, counts


def posterior_cooccurrences(model, chain, n_burn, want_matrix=True):
    """hdp_lpcm.py:1180-1186 on the device"""
    return chain.post_cooccurrence(model.zs_[n_burn:], model.n_components, want_matrix)


def expected_vi(zs_kept, cooc_row_sums, vi_sums):
    """posterior_vi.py:23-52 assembled from the device's sample-dependent sums:
    per sample, the time average of
        (1/N) [ sum_k n_k log2 n_k - 2 sum_i log2(sum_j C_ij [z_j = z_i]) + sum_i log2 sum_j C_ij ]"""
    S, T, N = zs_kept.shape
    K = int(zs_kept.max()) + 1
    flat = (np.arange(S * T)[:, None] * K + zs_kept.reshape(S * T, N)).ravel()
    nk = np.bincount(flat, minlength=S * T * K).reshape(S, T, K).astype(np.float64)
    with np.errstate(divide='ignore', invalid='ignore'):
        t1 = np.where(nk > 0, nk * np.log2(nk), 0.0).sum(axis=2)          # (S, T)
    t3 = np.log2(cooc_row_sums).sum(axis=1)                               # (T,)
    vi_t = (t1 - 2 * vi_sums.T + t3[None, :]) / N
    return vi_t.mean(axis=1)


def minimize_posterior_expected_vi(model, chain, n_burn, cooc=None):
    """posterior_vi.py:55-82; ties go to the highest network log-likelihood.  Needs
    ``posterior_cooccurrences`` to have run on ``chain`` (its matrices stay on the device)."""
    zs_kept = model.zs_[n_burn:]
    sums = chain.post_expected_vi_sums()
    if cooc is None:
        raise ValueError('the co-occurrence matrices are needed for their row sums')
    vis = expected_vi(zs_kept, cooc.sum(axis=2), sums)
    ids = np.arange(n_burn, model.zs_.shape[0])
    mins = np.where(vis == vis.min())[0]
    if mins.shape[0] > 1:
        best, best_ll = None, -np.inf
        for m in mins:
            ll = _network_loglik(model, chain, ids[m])
            if ll > best_ll:
                best, best_ll = ids[m], ll
        return int(best), vis
    return int(ids[mins[0]]), vis


def procrustes_align_samples(model):
    """hdp_lpcm.py:1141-1146: rotate every stored sample (and its cluster means) onto X_"""
    from scipy.linalg import orthogonal_procrustes
    ref = model.X_.reshape(-1, model.X_.shape[-1])
    for idx in range(model.Xs_.shape[0]):
        flat = model.Xs_[idx].reshape(ref.shape)
        R, _ = orthogonal_procrustes(flat, ref)
        model.Xs_[idx] = flat.dot(R).reshape(model.Xs_[idx].shape)
        model.mus_[idx] = model.mus_[idx].dot(R)


def _renormalize_row(row, T, N):
    """label_utils.py:10-37 for one stored sample given as a dict (z, beta, weights, mu, sigma)"""
    active = np.unique(row['z'].ravel())
    beta = row['beta'][active]
    beta = beta / beta.sum()
    init_w = row['weights'][0, 0, active]
    init_w = init_w / init_w.sum()
    trans_w = np.zeros((T, active.shape[0], active.shape[0]))
    for t in range(1, T):
        trans_w[t] = row['weights'][t, active][:, active]
        trans_w[t] /= np.sum(trans_w[t], axis=1).reshape(-1, 1)
    _, z = np.unique(row['z'].ravel(), return_inverse=True)
    return z.reshape(T, N), beta, init_w, trans_w, row['mu'][active], row['sigma'][active]


def _trace_row(chain, sid):
    """one stored sample of the device-resident trace as a dict"""
    tr = chain.hdp_trace_read(int(sid), 1)
    return dict(X=tr['Xs'][0], z=tr['zs'][0], weights=tr['weights'][0], beta=tr['betas'][0],
                mu=tr['mus'][0], sigma=tr['sigmas'][0], intercept=tr['intercepts'][0],
                lmbda=tr['lambdas'][0])


def select_model_device(model, chain, n_burn):
    """``select_model`` for the undirected model's device-resident trace: the label counts, the
    co-occurrence matrices, the VI sums, the network log-likelihoods and the forward-algorithm
    marginal likelihood (approx_bic.py:54-76) come from kernels over the trace where it lies; the
    host sees (S, T, K) counts, (T, S) sums and the handful of samples it selects - never the
    positions or labels of all samples.  Same attributes as ``select_model``;
    ``cooccurrence_probas_`` is left on the device (the estimator pulls it when it is asked for)."""
    n_total = model.logps_.shape[0]
    S = n_total - n_burn
    T, N = chain.T, chain.N
    nk = chain.post_trace_label_counts(n_burn, S)                  # (S, T, K)
    present = nk > 0
    counts = present.any(axis=1).sum(axis=1)                       # approx_bic.py:40-51
    model._counts_t = present.sum(axis=2).T                        # approx_bic.py:26-37: (T, S)
    Y = model.Y_fit_
    off_diag_sum = np.sum(Y) - np.einsum('ikk', Y).sum()

    def network_loglik(row):
        chain.set_positions(row['X'])
        return chain.loglik_full([row['intercept']])[0]

    # approx_bic.py:79-162
    bic, models = [], []
    for k in np.unique(counts):
        lp = np.where(counts == k, model.logps_[n_burn:], -np.inf)
        map_id = int(np.argmax(lp)) + n_burn
        row = _trace_row(chain, map_id)
        z, beta, init_w, trans_w, mu, sigma = _renormalize_row(row, T, N)
        loglik_k = network_loglik(row)
        bic_k = -2 * loglik_k + np.log(0.5 * off_diag_sum)
        bic_k -= 2 * chain.post_latent_marginal_loglik(init_w, trans_w, mu, sigma, row['lmbda'],
                                                       row=map_id)
        n_params = ((model.n_features + 1) * k + (k - 1) + (k - 1) + (T - 1) * k * (k - 1))
        bic_k += n_params * np.log(N * T)
        models.append(SimpleNamespace(beta=beta, init_weights=init_w, trans_weights=trans_w,
                                      X=row['X'], mu=mu, sigma=sigma, lmbda=row['lmbda'], z=row['z'],
                                      intercept=row['intercept'], radii=None))
        bic.append([k, bic_k, loglik_k, map_id])
    model.bic_, model.models_, model.counts_ = np.array(bic), models, counts
    # hdp_lpcm.py:1180-1186 (the matrices stay on the device)
    _, row_sums = chain.post_trace_cooccurrence(n_burn, S, want_matrix=False)
    model._lazy_cooc = True
    if model.selection_type == 'vi':
        sums = chain.post_expected_vi_sums()                       # (T, S)
        nkf = nk.astype(np.float64)
        with np.errstate(divide='ignore', invalid='ignore'):
            t1 = np.where(nk > 0, nkf * np.log2(nkf), 0.0).sum(axis=2)     # posterior_vi.py:31-36
        t3 = np.log2(row_sums).sum(axis=1)
        vis = ((t1 - 2 * sums.T + t3[None, :]) / N).mean(axis=1)
        mins = np.where(vis == vis.min())[0]
        best = int(mins[0]) + n_burn
        if mins.shape[0] > 1:                                      # posterior_vi.py:62-80
            # ties (identical partitions: thousands once the chain has settled) go to the highest
            # network log-likelihood, first one on equality; the loop stored it with every sample
            lls = np.array(getattr(model, '_trace_logliks', np.full(n_total, np.nan)))[mins + n_burn]
            for q in np.where(np.isnan(lls))[0]:
                lls[q] = network_loglik(_trace_row(chain, int(mins[q]) + n_burn))
            best = int(mins[int(np.argmax(lls))]) + n_burn
        model.expected_vis_ = vis
        row = _trace_row(chain, best)
        model.logp_ = model.logps_[best]
        model.X_, model.intercept_ = row['X'].copy(), row['intercept']
        model.lambda_ = row['lmbda']
        (model.z_, model.beta_, model.init_weights_, model.trans_weights_, model.mu_,
         model.sigma_) = _renormalize_row(row, T, N)
        model.selected_id_ = best
    else:
        if model.selection_type == 'bic':
            mid = int(np.argmin(model.bic_[:, 1]))
            model.best_k_ = int(model.bic_[mid, 0])
        elif model.selection_type == 'map':
            model.best_k_ = int(np.argmax(np.bincount(model.counts_)))
            mid = int(np.argwhere(model.bic_[:, 0] == model.best_k_)[0, 0])
        else:
            raise ValueError('Selection type not recognized')
        m = model.models_[mid]
        model.selected_id_ = int(model.bic_[mid, 3])
        model.logp_ = model.logps_[model.selected_id_]
        model.X_, model.intercept_ = m.X.copy(), m.intercept
        model.mu_, model.sigma_ = m.mu, m.sigma
        _, z = np.unique(m.z.ravel(), return_inverse=True)
        model.z_ = z.reshape(T, N)
        model.beta_, model.init_weights_ = m.beta, m.init_weights
        model.trans_weights_, model.lambda_ = m.trans_weights, m.lmbda
    return model


def posterior_group_counts_from(counts_t):
    """label_utils.py:73-81 for every t, from the (T, S) numbers of labels in use"""
    ids, freqs = [], []
    for row in counts_t:
        freq = np.bincount(row)
        index = np.where(freq != 0)[0]
        ids.append(index)
        freqs.append(freq[index])
    return ids, freqs


def posterior_group_counts(model, n_burn):
    """label_utils.py:73-81 for every t"""
    ids, freqs = [], []
    for row in cluster_counts_t(model.zs_, n_burn):
        freq = np.bincount(row)
        index = np.where(freq != 0)[0]
        ids.append(index)
        freqs.append(freq[index])
    return ids, freqs


def select_model(model, chain, n_burn):
    """hdp_lpcm.py:1085-1139: fills bic_, models_, counts_, cooccurrence_probas_ and the
    selected sample's attributes according to ``model.selection_type``."""
    T, N = model.zs_.shape[1:]
    model.bic_, model.models_, model.counts_ = select_bic(model, chain, n_burn)
    model.cooccurrence_probas_ = posterior_cooccurrences(model, chain, n_burn)
    if model.selection_type == 'vi':
        best, model.expected_vis_ = minimize_posterior_expected_vi(
            model, chain, n_burn, cooc=model.cooccurrence_probas_)
        model.logp_ = model.logps_[best]
        model.X_, model.intercept_ = model.Xs_[best].copy(), model.intercepts_[best]
        model.lambda_ = model.lambdas_[best]
        if model.is_directed:
            model.radii_ = model.radiis_[best]
        (model.z_, model.beta_, model.init_weights_, model.trans_weights_, model.mu_,
         model.sigma_) = renormalize_weights(model, best)
        model.selected_id_ = best
    else:
        if model.selection_type == 'bic':
            mid = int(np.argmin(model.bic_[:, 1]))
            model.best_k_ = int(model.bic_[mid, 0])
        elif model.selection_type == 'map':
            model.best_k_ = int(np.argmax(np.bincount(model.counts_)))
            mid = int(np.argwhere(model.bic_[:, 0] == model.best_k_)[0, 0])
        else:
            raise ValueError('Selection type not recognized')
        m = model.models_[mid]
        model.selected_id_ = int(model.bic_[mid, 3])
        model.logp_ = model.logps_[model.selected_id_]
        model.X_, model.intercept_ = m.X.copy(), m.intercept
        model.mu_, model.sigma_ = m.mu, m.sigma
        if model.is_directed:
            model.radii_ = m.radii
        _, z = np.unique(m.z.ravel(), return_inverse=True)
        model.z_ = z.reshape(T, N)
        model.beta_, model.init_weights_ = m.beta, m.init_weights
        model.trans_weights_, model.lambda_ = m.trans_weights, m.lmbda
    chain.post_release()
    return model
