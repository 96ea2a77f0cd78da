"""One-step-ahead forecasts on the device against the oracle and the
reference's own outputs (tests/golden/forecast.npz: forecast_probas_map_ / _plugin_ /
_marginalized_ / forecast_probas / forecast_probas_pp_ and forecast.pyx:marginal_forecast).
Needs an MI355X: -m gpu.  float64 sums of expit terms: 1e-12 relative."""
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import load_golden
from oracle import forecast_oracle as fo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    import dynetlsm_amd
    return dynetlsm_amd


@pytest.fixture(scope='module')
def model():
    from dynetlsm_amd import posterior as post
    g, p = load_golden('forecast.npz'), load_golden('post.npz')
    m = SimpleNamespace(
        Y_fit_=p['u_Y'], zs_=p['u_zs'], Xs_=p['u_Xs'], n_burn_=int(p['u_n_burn']), thin=None,
        n_components=int(p['u_K']), n_features=2, is_directed=False,
        intercepts_=p['u_intercepts'], mus_=p['u_mus'], sigmas_=p['u_sigmas'],
        betas_=p['u_betas'], weights_=p['u_weights'], lambdas_=p['u_lambdas'],
        logps_=p['u_logps'], random_state=11)
    best = int(g['best'])
    (m.z_, m.beta_, m.init_weights_, m.trans_weights_, m.mu_, m.sigma_) = \
        post.renormalize_weights(m, best)
    m.X_, m.intercept_, m.lambda_ = m.Xs_[best], m.intercepts_[best], m.lambdas_[best]
    m.intercepts_mean_ = m.intercepts_[m.n_burn_:].mean(axis=0)
    return m, g


def test_reference_forecasts(eng, model):
    from dynetlsm_amd import forecast as fc
    m, g = model
    T, N, _ = m.Y_fit_.shape
    with eng.Chain(T, N, 2, 'undirected') as c:
        np.testing.assert_allclose(fc.forecast_probas_map(m, c), g['map'], rtol=1e-12)
        np.testing.assert_allclose(fc.forecast_probas_plugin(m, c), g['plugin'], rtol=1e-12)
        np.testing.assert_allclose(fc.forecast_probas_marginalized(m, c), g['marginalized'],
                                   rtol=1e-11, atol=1e-15)
        # same MT19937 draws as the reference (random_state=11)
        np.testing.assert_allclose(fc.forecast_probas(m, c, n_samples=25, batch=7), g['mc'],
                                   rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(fc.forecast_probas_pp(m, c, batch=5), g['pp'], rtol=1e-12)


@pytest.mark.parametrize('N,D,S', [(30, 2, 12), (257, 2, 19), (130, 3, 5), (64, 1, 33)])
def test_kernels_match_oracle(eng, N, D, S):
    rng = np.random.RandomState(N + S)
    Xs = rng.randn(S, N, D)
    b = rng.randn(S)
    with eng.Chain(2, N, D, 'undirected') as c:
        for zd in (False, True):
            np.testing.assert_allclose(c.forecast_mean_probas(Xs, b, zero_diag=zd),
                                       fo.mean_probas(Xs, b, zero_diag=zd), rtol=1e-12, atol=1e-15)
        x = rng.randn(N, D)
        W = rng.gamma(1.0, 1.0, size=(S, N))
        d = np.sqrt(((x[:, None] - x[None]) ** 2).sum(-1))
        num = sum(np.outer(W[s], W[s]) / (1 + np.exp(-(b[s] - d))) for s in range(S))
        den = sum(np.outer(W[s], W[s]) for s in range(S))
        want = num / den
        np.fill_diagonal(want, 0.0)
        np.testing.assert_allclose(c.forecast_marginal(x, W, b), want, rtol=1e-12, atol=1e-15)


def test_marginal_forecast_matches_cython(eng, model):
    from dynetlsm_amd import forecast as fc
    m, g = model
    S, N = 12, m.Xs_.shape[2]
    x = g['mf_x']
    W = np.stack([fc.mixture_density(x, m.Xs_[s, -1], m.weights_[s, -1][m.zs_[s, -1]],
                                     m.lambdas_[s, 0], m.mus_[s], m.sigmas_[s]) for s in range(S)])
    with eng.Chain(2, N, 2, 'undirected') as c:
        np.testing.assert_allclose(c.forecast_marginal(x, W, m.intercepts_[:S].ravel()),
                                   g['mf_probas'], rtol=1e-11, atol=1e-15)


def test_lpcm_reference_forecasts(eng):
    """lpcm.py:228-318 (time-homogeneous weights, its own indexing quirks)"""
    from dynetlsm_amd import forecast as fc
    g, p = load_golden('forecast.npz'), load_golden('post.npz')
    sid, n_burn = int(g['lpcm_id']), int(p['u_n_burn'])
    m = SimpleNamespace(
        Y_fit_=p['u_Y'], zs_=p['u_zs'], Xs_=p['u_Xs'], n_burn_=n_burn, thin=None,
        n_components=int(p['u_K']), n_features=2, is_directed=False,
        intercepts_=p['u_intercepts'], mus_=p['u_mus'], sigmas_=p['u_sigmas'],
        trans_weights_=np.ascontiguousarray(p['u_weights'][:, 1]), lambdas_=p['u_lambdas'],
        random_state=11)
    m.z_, m.trans_weight_ = m.zs_[sid], m.trans_weights_[sid]
    m.mu_, m.sigma_, m.lambda_ = m.mus_[sid], m.sigmas_[sid], m.lambdas_[sid]
    m.X_, m.intercept_ = m.Xs_[sid], m.intercepts_[sid]
    m.intercepts_mean_ = m.intercepts_[n_burn:].mean(axis=0)
    T, N, _ = m.Y_fit_.shape
    with eng.Chain(T, N, 2, 'undirected') as c:
        np.testing.assert_allclose(fc.lpcm_forecast_probas_map(m, c), g['lpcm_map'], rtol=1e-12)
        np.testing.assert_allclose(fc.lpcm_forecast_probas_plugin(m, c), g['lpcm_plugin'],
                                   rtol=1e-12)
        np.testing.assert_allclose(fc.lpcm_forecast_probas_marginalized(m, c),
                                   g['lpcm_marginalized'], rtol=1e-11, atol=1e-15)
        np.testing.assert_allclose(fc.lpcm_forecast_probas(m, c, n_samples=25, batch=9),
                                   g['lpcm_mc'], rtol=1e-12, atol=1e-15)


def test_fitted_model_forecasts(eng):
    rng = np.random.RandomState(0)
    T, N = 3, 40
    Y = (rng.rand(T, N, N) < 0.15).astype(np.float64)
    Y = np.triu(Y, 1); Y = Y + Y.transpose(0, 2, 1)
    m = eng.DynamicNetworkHDPLPCM(n_iter=30, burn=15, tune=15, n_components=5,
                                  random_state=2).fit(Y)
    for P in (m.forecast_probas_map_, m.forecast_probas_plugin_, m.forecast_probas_marginalized_,
              m.forecast_probas(n_samples=20), m.forecast_probas_pp_):
        assert P.shape == (N, N) and np.isfinite(P).all()
        assert (P >= 0).all() and (P <= 1).all() and np.allclose(P, P.T)
    ml = eng.DynamicNetworkLPCM(n_iter=30, burn=15, tune=15, n_components=4,
                                random_state=2).fit(Y)
    for P in (ml.forecast_probas_map_, ml.forecast_probas_plugin_,
              ml.forecast_probas_marginalized_, ml.forecast_probas(n_samples=20)):
        assert P.shape == (N, N) and np.isfinite(P).all()
        assert (P >= 0).all() and (P <= 1).all() and np.allclose(P, P.T)
