"""One-step-ahead forecasts of a fitted (undirected) DynamicNetworkHDPLPCM
(SURVEY.md 8f-4), mirroring hdp_lpcm.py:496-626 and forecast.pyx.

The O(n_samples N^2) accumulations run on the device (``Chain.forecast_mean_probas``,
``Chain.forecast_marginal``); label / position draws stay on the host in the reference's
MT19937 order, as do the O(n_samples N K) plug-in positions and mixture densities.
"""
import numpy as np

from .posterior import renormalize_weights

__all__ = ['forecast_probas_map', 'forecast_probas_plugin', 'forecast_probas_marginalized',
           'forecast_probas', 'forecast_probas_pp', 'mixture_density',
           'lpcm_forecast_probas_map', 'lpcm_forecast_probas_plugin',
           'lpcm_forecast_probas_marginalized', 'lpcm_forecast_probas']


def _kept(model):
    n_burn = min(-(-model.n_burn_ // (getattr(model, 'thin', None) or 1)), model.zs_.shape[0] - 1)    # ceil: hdp_lpcm.py:465
    return np.arange(n_burn, model.zs_.shape[0])


def forecast_probas_map(model, chain):
    """hdp_lpcm.py:497-508: plug-in estimate from the selected sample"""
    ws = model.trans_weights_[-1][model.z_[-1]]
    lm = np.ravel(model.lambda_)[0]
    X_ahead = np.zeros((model.X_.shape[1], model.n_features))
    for g in np.unique(model.z_[-1]):
        X_ahead += ws[:, g].reshape(-1, 1) * (lm * model.mu_[g] + (1 - lm) * model.X_[-1])
    return chain.forecast_mean_probas(X_ahead[None], np.ravel(model.intercept_)[:1])


def _plugin_positions(model, ids):
    N = model.Xs_.shape[2]
    X_hat = np.zeros((N, model.n_features))
    for idx in ids:
        z, _, _, trans_w, mu, sigma = renormalize_weights(model, idx)
        ws = trans_w[-1][z[-1]]
        lm = np.ravel(model.lambdas_[idx])[0]
        for g in np.unique(z[-1]):
            X_hat += (1. / ids.shape[0]) * ws[:, g].reshape(-1, 1) * (
                lm * mu[g] + (1 - lm) * model.Xs_[idx, -1])
    return X_hat


def forecast_probas_plugin(model, chain):
    """hdp_lpcm.py:510-527"""
    X_hat = _plugin_positions(model, _kept(model))
    return chain.forecast_mean_probas(X_hat[None], np.ravel(model.intercepts_mean_)[:1])


def mixture_density(x, x_prev, weights_rows, lmbda, mean, sigma):
    """mixture_normal_pdf (forecast.pyx:39-54) for every node: sum_k w_ik N(x_i; lmbda mu_k +
    (1 - lmbda) x_prev_i, sigma_k I)"""
    D = x.shape[1]
    m = lmbda * mean[None, :, :] + (1 - lmbda) * x_prev[:, None, :]           # (N, K, D)
    ss = 0.5 * np.sum((x[:, None, :] - m) ** 2, axis=2) / sigma[None, :]
    pdf = np.exp(-0.5 * D * np.log(2 * np.pi * sigma)[None, :] - ss)
    return np.sum(weights_rows * pdf, axis=1)


def forecast_probas_marginalized(model, chain, renormalize=True):
    """hdp_lpcm.py:529-553 + marginal_forecast (forecast.pyx:79-128)"""
    ids = _kept(model)
    X_hat = _plugin_positions(model, ids)
    W = np.empty((ids.shape[0], X_hat.shape[0]))
    for r, idx in enumerate(ids):
        z, w_t = model.zs_[idx, -1], model.weights_[idx, -1]
        mu, sigma = model.mus_[idx], model.sigmas_[idx]
        if renormalize:                                   # forecast.pyx:57-68
            active, z = np.unique(z, return_inverse=True)
            w_t = w_t[active][:, active]
            w_t = w_t / np.sum(w_t, axis=1).reshape(-1, 1)
            mu, sigma = mu[active], sigma[active]
        W[r] = mixture_density(X_hat, model.Xs_[idx, -1], w_t[z], np.ravel(model.lambdas_[idx])[0],
                               mu, sigma)
    return chain.forecast_marginal(X_hat, W, model.intercepts_[ids].ravel())


def _draw_positions(rng, z_last, wt, mu, sigma, lm, X_last, n_groups):
    """one draw of (labels, positions) in the reference's order (hdp_lpcm.py:565-581)"""
    N = z_last.shape[0]
    zt = np.zeros(N, dtype=np.int64)
    Xt = np.zeros((N, X_last.shape[1]))
    for g in range(n_groups):
        mask = z_last == g
        zt[mask] = rng.choice(np.arange(n_groups), p=wt[g, :], size=int(np.sum(mask)))
    for g in range(n_groups):
        mask = zt == g
        Xt[mask, :] = (sigma[g] * rng.randn(int(np.sum(mask)), 2) +
                       (lm * mu[g] + (1 - lm) * X_last[mask, :]))
    return Xt


def forecast_probas(model, chain, n_samples=5000, rng=None, batch=512):
    """hdp_lpcm.py:555-587: Monte Carlo over the selected sample's one-step-ahead law"""
    from .lsm import check_random_state
    rng = check_random_state(model.random_state) if rng is None else rng
    n_groups = model.mu_.shape[0]
    wt = model.trans_weights_[-1]
    lm = np.ravel(model.lambda_)[0]
    b = np.ravel(model.intercept_)[:1]
    N = model.X_.shape[1]
    probas = np.zeros((N, N))
    for s0 in range(0, n_samples, batch):
        ns = min(batch, n_samples - s0)
        Xs = np.stack([_draw_positions(rng, model.z_[-1], wt, model.mu_, model.sigma_, lm,
                                       model.X_[-1], n_groups) for _ in range(ns)])
        probas += chain.forecast_mean_probas(Xs, b, zero_diag=True) * (ns / float(n_samples))
    return probas


def forecast_probas_pp(model, chain, rng=None, batch=512):
    """hdp_lpcm.py:589-626: posterior predictive, one draw per kept sample"""
    from .lsm import check_random_state
    rng = check_random_state(model.random_state) if rng is None else rng
    ids = _kept(model)
    N = model.Xs_.shape[2]
    probas = np.zeros((N, N))
    for s0 in range(0, ids.shape[0], batch):
        chunk = ids[s0:s0 + batch]
        Xs = []
        for idx in chunk:
            z, _, _, trans_w, mu, sigma = renormalize_weights(model, idx)
            # hdp_lpcm.py:603 indexes the transition rows by NODE (wt = trans_w[-1][z[-1]])
            # and then reads row g of that for group g (:612); kept as it is
            Xs.append(_draw_positions(rng, z[-1], trans_w[-1][z[-1]], mu, sigma,
                                      np.ravel(model.lambdas_[idx])[0], model.Xs_[idx, -1],
                                      mu.shape[0]))
        probas += (chain.forecast_mean_probas(np.stack(Xs), model.intercepts_[chunk].ravel()) *
                   (chunk.shape[0] / float(ids.shape[0])))
    return probas


# ---------------------------------------------------------------------------
# DynamicNetworkLPCM (lpcm.py:228-318): time-homogeneous transition matrix
# ---------------------------------------------------------------------------
def lpcm_forecast_probas_map(model, chain):
    """lpcm.py:229-239"""
    ws = model.trans_weight_[model.z_[-1]]
    lm = np.ravel(model.lambda_)[0]
    X_ahead = np.zeros((model.X_.shape[1], model.n_features))
    for g in range(model.n_components):
        X_ahead += ws[:, g].reshape(-1, 1) * (lm * model.mu_[g] + (1 - lm) * model.X_[-1])
    return chain.forecast_mean_probas(X_ahead[None], np.ravel(model.intercept_)[:1])


def _lpcm_plugin_positions(model, ids):
    # lpcm.py:250 / :269 read the LAST stored sample's transition matrix for every
    # sample (``self.trans_weights_[-1][z]``); kept as it is
    X_hat = np.zeros((model.Xs_.shape[2], model.n_features))
    for idx in ids:
        ws = model.trans_weights_[-1][model.zs_[idx, -1]]
        lm = np.ravel(model.lambdas_[idx])[0]
        for g in range(model.n_components):
            X_hat += (1. / ids.shape[0]) * ws[:, g].reshape(-1, 1) * (
                lm * model.mus_[idx][g] + (1 - lm) * model.Xs_[idx, -1])
    return X_hat


def lpcm_forecast_probas_plugin(model, chain):
    """lpcm.py:241-257"""
    X_hat = _lpcm_plugin_positions(model, _kept(model))
    return chain.forecast_mean_probas(X_hat[None], np.ravel(model.intercepts_mean_)[:1])


def lpcm_forecast_probas_marginalized(model, chain):
    """lpcm.py:259-283: marginal_forecast with the samples' own transition matrices,
    renormalize=False"""
    ids = _kept(model)
    X_hat = _lpcm_plugin_positions(model, ids)
    W = np.stack([mixture_density(X_hat, model.Xs_[idx, -1],
                                  model.trans_weights_[idx][model.zs_[idx, -1]],
                                  np.ravel(model.lambdas_[idx])[0], model.mus_[idx],
                                  model.sigmas_[idx]) for idx in ids])
    return chain.forecast_marginal(X_hat, W, model.intercepts_[ids].ravel())


def lpcm_forecast_probas(model, chain, n_samples=5000, rng=None, batch=512):
    """lpcm.py:285-317"""
    from .lsm import check_random_state
    rng = check_random_state(model.random_state) if rng is None else rng
    n_groups = model.mu_.shape[0]
    lm = np.ravel(model.lambda_)[0]
    b = np.ravel(model.intercept_)[:1]
    N = model.X_.shape[1]
    probas = np.zeros((N, N))
    for s0 in range(0, n_samples, batch):
        ns = min(batch, n_samples - s0)
        Xs = np.stack([_draw_positions(rng, model.z_[-1], model.trans_weight_, model.mu_,
                                       model.sigma_, lm, model.X_[-1], n_groups)
                       for _ in range(ns)])
        probas += chain.forecast_mean_probas(Xs, b, zero_diag=True) * (ns / float(n_samples))
    return probas
