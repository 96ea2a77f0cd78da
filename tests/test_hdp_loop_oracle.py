"""Pins oracle/hdp_loop_oracle.py - the CPU restatement of the HDP-LPCM's auxiliary /
conjugate / hyper-parameter updates and log-posterior (hdp_lpcm.py:876-1023, :1188-1280) that
the device loop (dlsm_hdp_run) is compared with - against the 9-iteration trace recorded
from the reference's own DynamicNetworkHDPLPCM._fit (tests/golden/hdp_trace.npz, written by
make_golden.py).  The update code is driven here by the reference's MT19937 stream, call for
call; the GPU tests drive THE SAME code with the engine's Philox draws.  CPU only."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle as orc
from oracle import hdp_loop_oracle as hlo


@pytest.fixture(scope='module')
def g():
    return load_golden('hdp_trace.npz')


def _hyper(g):
    return hlo.Hyper(gamma=float(g['h0_gamma']), alpha_init=float(g['h0_alpha_init']),
                     alpha=float(g['h0_alpha']), kappa=float(g['h0_kappa']),
                     mean_variance_prior=float(g['h0_mean_variance_prior']),
                     b=float(g['h0_b']), a=float(g['h0_a']), a0=float(g['h0_a0']),
                     b0=float(g['h0_b0']), c0=float(g['h0_c0']), d0=float(g['h0_d0']))


def test_updates_and_logp_reproduce_the_reference_fit(g):
    Y = g['Y']
    Xs, ics = g['tr_Xs'], g['tr_intercepts']
    mus, sigmas, zs = g['tr_mus'], g['tr_sigmas'], g['tr_zs']
    betas, weights, lambdas, logps = (g['tr_betas'], g['tr_weights'], g['tr_lambdas'],
                                      g['tr_logps'])
    n_total, T, N, D = Xs.shape
    rng = np.random.RandomState(0)
    rng.set_state(('MT19937', g['rng_keys'], int(g['rng_pos']), int(g['rng_has_gauss']),
                   float(g['rng_cached'])))
    hp = _hyper(g)
    ip = g['h0_intercept_prior']
    grid = orc.SamplerGrid(T, N, float(g['h0_step_size_X']), tune=3, tune_interval=2)
    isamp = orc.ScalarMetropolis(0.1, 3, 100)            # hdp_lpcm.py:740-742
    draws = hlo.MTDraws(rng)
    for it in range(1, n_total):
        X = Xs[it - 1].copy(); ic = ics[it - 1].copy(); z = zs[it - 1].copy()
        mu = mus[it - 1].copy(); sigma = sigmas[it - 1].copy()
        w = weights[it - 1].copy(); beta = betas[it - 1].copy()
        lmbda = lambdas[it - 1].copy()
        st = orc.ChainState(X, grid, Y=Y, intercept=ic, mu=mu, sigma=sigma, lmbda=lmbda, z=z)
        X = orc.center(st.sweep_py(orc.MTDraws(rng), order='reference').copy())

        def lp(x):
            return (orc.dynamic_network_loglikelihood_undirected(Y, X, x[0]) -
                    (x[0] - ip[0]) ** 2 / (2 * 2))
        ic = isamp.step_rw(ic, lp, rng)
        z, n, nk, _ = orc.sample_labels_block_mt(X, mu, sigma, lmbda, w, rng)
        beta, lmbda, aux = hlo.gibbs_updates(X, z, n, nk, mu, sigma, beta, w, lmbda, hp, draws)
        np.testing.assert_array_equal(z, zs[it])
        np.testing.assert_allclose(beta, betas[it], rtol=1e-10)
        np.testing.assert_allclose(w, weights[it], rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(mu, mus[it], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(sigma, sigmas[it], rtol=1e-9)
        np.testing.assert_allclose(lmbda, lambdas[it], rtol=1e-10)
        ll = orc.dynamic_network_loglikelihood_undirected(Y, X, ic[0])
        got = hlo.log_posterior(ll, X, ic, mu, sigma, z, w, beta, lmbda, hp, ip, 2)
        np.testing.assert_allclose(got, logps[it], rtol=1e-9)
        # structure of the auxiliary variables (sample_auxillary.py)
        assert (aux['m'] <= n).all() and (aux['m'][n > 0] >= 1).all()
        assert (aux['w'] <= aux['m'][1:, np.arange(4), np.arange(4)]).all()
    for name in ('gamma', 'alpha_init', 'alpha', 'kappa', 'mean_variance_prior', 'b'):
        np.testing.assert_allclose(np.ravel(getattr(hp, name))[0],
                                   np.ravel(g['h1_' + name])[0], rtol=1e-9)


def test_truncnorm_quantile_is_scipys():
    from scipy.stats import truncnorm
    rng = np.random.RandomState(3)
    for mean, var in [(0.9, 0.01), (0.95, 1e-5), (1.3, 1e-4), (-0.2, 1e-4), (0.5, 4.0),
                      (3.0, 1e-3), (-2.0, 1e-3)]:
        std = np.sqrt(var)
        a, b = (0 - mean) / std, (1 - mean) / std
        for q in list(rng.rand(5)) + [1e-12, 1 - 1e-12]:
            want = truncnorm.ppf(q, a, b, loc=mean, scale=std)
            got = hlo.truncnorm_quantile(q, mean, var)
            assert 0.0 <= got <= 1.0
            np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-12)
        x = hlo.truncnorm_quantile(0.37, mean, var)
        np.testing.assert_allclose(hlo.truncnorm_logpdf(x, mean, var),
                                   truncnorm.logpdf(x, a, b, loc=mean, scale=std), rtol=1e-9)


def test_philox_draws_have_the_right_laws():
    """the counter-based samplers the device loop shares with the oracle: moments of the
    gamma (both branches), beta, binomial-by-Bernoullis and table-count draws"""
    n = 4000
    for a in (0.3, 1.0, 7.5):
        x = np.array([hlo.PhiloxDraws(5, 1, it).std_gamma(hlo.K_BETA, 3, a) for it in range(n)])
        assert abs(x.mean() - a) < 5 * np.sqrt(a / n)
        assert abs(x.var() - a) < 0.25 * a + 0.05
    x = np.array([hlo.PhiloxDraws(5, 1, it).beta(hlo.K_RHO, 0, 2.0, 5.0) for it in range(n)])
    assert abs(x.mean() - 2 / 7) < 0.02
    x = np.array([hlo.PhiloxDraws(6, 0, it).binomial(hlo.K_OVERRIDE, 4, 9, 0.3) for it in range(n)])
    assert abs(x.mean() - 2.7) < 0.1 and x.max() <= 9
    # tables: E[m] = sum_i p / (p + i)
    p, cnt = 1.7, 25
    x = np.array([hlo.PhiloxDraws(7, 2, it).tables_cell(11, p, cnt) for it in range(n)])
    assert x.min() >= 1 and abs(x.mean() - np.sum(p / (p + np.arange(cnt)))) < 0.1
    d = np.array([hlo.PhiloxDraws(8, 0, it).dirichlet(hlo.K_W, 40, np.array([0.5, 2.0, 4.0]))
                  for it in range(1500)])
    np.testing.assert_allclose(d.sum(axis=1), 1.0, rtol=1e-12)
    np.testing.assert_allclose(d.mean(axis=0), np.array([0.5, 2.0, 4.0]) / 6.5, atol=0.02)
    # different kinds / indices / iterations are different draws
    a1 = hlo.PhiloxDraws(5, 1, 3).std_gamma(hlo.K_BETA, 0, 2.0)
    assert a1 != hlo.PhiloxDraws(5, 1, 3).std_gamma(hlo.K_W0, 0, 2.0)
    assert a1 != hlo.PhiloxDraws(5, 1, 3).std_gamma(hlo.K_BETA, 1, 2.0)
    assert a1 != hlo.PhiloxDraws(5, 1, 4).std_gamma(hlo.K_BETA, 0, 2.0)
    assert a1 == hlo.PhiloxDraws(5, 1, 3).std_gamma(hlo.K_BETA, 0, 2.0)
