"""Pins the host-side part of the HDP-LPCM iteration (dynetlsm_amd.hdp_updates:
conjugate / auxiliary-variable updates and the log-posterior) against a trace
recorded from the reference's DynamicNetworkHDPLPCM._fit.  CPU only.

The loop below replays hdp_lpcm.py:823-1069 in the reference's order with its
MT19937 stream: sweep / labels / log-likelihood and the label-wise sums over the nodes
come from the (reference-pinned) oracle - the product takes those sums from the device,
tests/test_gpu_models.py - everything else is the product's host code under test."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle as orc
from oracle.hdp_sums import NumpyLabelSums
from dynetlsm_amd import hdp_updates as hu


@pytest.fixture(scope='module')
def g():
    return load_golden('hdp_trace.npz')


def test_hdp_host_updates_reproduce_reference_fit(g):
    Y = g['Y']
    Xs, ics = g['tr_Xs'], g['tr_intercepts']
    mus, sigmas, zs = g['tr_mus'], g['tr_sigmas'], g['tr_zs']
    betas, weights, lambdas, logps = (g['tr_betas'], g['tr_weights'], g['tr_lambdas'],
                                      g['tr_logps'])
    n_total, T, N, D = Xs.shape
    K = sigmas.shape[1]
    tune, tune_interval = 3, 2
    rng = np.random.RandomState(0)
    rng.set_state(('MT19937', g['rng_keys'], int(g['rng_pos']), int(g['rng_has_gauss']),
                   float(g['rng_cached'])))
    hp = hu.HDPHyper(K, gamma=float(g['h0_gamma']), alpha_init=float(g['h0_alpha_init']),
                     alpha=float(g['h0_alpha']), kappa=float(g['h0_kappa']),
                     mean_variance_prior=float(g['h0_mean_variance_prior']),
                     b=float(g['h0_b']), a=float(g['h0_a']), a0=float(g['h0_a0']),
                     b0=float(g['h0_b0']), c0=float(g['h0_c0']), d0=float(g['h0_d0']))
    intercept_prior = g['h0_intercept_prior']
    grid = orc.SamplerGrid(T, N, float(g['h0_step_size_X']), tune=tune,
                           tune_interval=tune_interval)
    isamp = orc.ScalarMetropolis(0.1, tune, 100)            # hdp_lpcm.py:740-742
    for it in range(1, n_total):
        X = Xs[it - 1].copy(); ic = ics[it - 1].copy(); z = zs[it - 1].copy()
        mu = mus[it - 1].copy(); sigma = sigmas[it - 1].copy()
        w = weights[it - 1].copy(); beta = betas[it - 1].copy()
        lmbda = lambdas[it - 1].copy()
        st = orc.ChainState(X, grid, Y=Y, intercept=ic, mu=mu, sigma=sigma, lmbda=lmbda,
                            z=z)
        X = orc.center(st.sweep_py(orc.MTDraws(rng), order='reference').copy())

        def lp(x):
            return (orc.dynamic_network_loglikelihood_undirected(Y, X, x[0]) -
                    (x[0] - intercept_prior[0]) ** 2 / (2 * 2))
        ic = isamp.step_rw(ic, lp, rng)
        z, n, nk, _ = orc.sample_labels_block_mt(X, mu, sigma, lmbda, w, rng)
        sums = NumpyLabelSums(X, z, K)
        beta, lmbda = hu.gibbs_updates(sums, n, nk, mu, sigma, beta, w, lmbda, hp, rng)
        np.testing.assert_allclose(X, Xs[it], atol=1e-9)
        np.testing.assert_allclose(ic, ics[it], atol=1e-10)
        np.testing.assert_array_equal(z, zs[it])
        np.testing.assert_allclose(beta, betas[it], rtol=1e-10)
        np.testing.assert_allclose(w, weights[it], rtol=1e-9, atol=1e-300)
        np.testing.assert_allclose(mu, mus[it], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(sigma, sigmas[it], rtol=1e-9)
        np.testing.assert_allclose(lmbda, lambdas[it], rtol=1e-10)
        ll = orc.dynamic_network_loglikelihood_undirected(Y, X, ic[0])
        lp_it = ll + hu.log_posterior_terms(sums, ic, intercept_prior, 2, mu, sigma, w,
                                            beta, lmbda, hp)
        np.testing.assert_allclose(np.ravel(lp_it)[0], logps[it], rtol=1e-9)
    for name in ('gamma', 'alpha_init', 'alpha', 'kappa', 'mean_variance_prior', 'b'):
        np.testing.assert_allclose(np.ravel(getattr(hp, name))[0],
                                   np.ravel(g['h1_' + name])[0], rtol=1e-9)


def _tables_cell_by_cell(n, beta, alpha_init, alpha, kappa, rng):
    """sample_auxillary.py:6-28 as the reference runs it: one binomial call per cell"""
    T, K, _ = n.shape
    m = np.zeros((T, K, K), dtype=np.int64)
    pr0 = alpha_init * beta
    pr = alpha * beta + kappa * np.eye(K)
    for t in range(T):
        for j in range(1 if t == 0 else K):
            for k in range(K):
                c = int(n[t, j, k])
                p = pr0[k] if t == 0 else pr[j, k]
                if c > 0:
                    m[t, j, k] = rng.binomial(1, p / (p + np.arange(c))).sum()
    return m


@pytest.mark.parametrize('seed', range(12))
def test_native_table_draws_consume_the_numpy_stream_like_the_reference(seed):
    g = np.random.RandomState(seed)
    T, K = int(g.randint(1, 6)), int(g.randint(1, 9))
    n = g.poisson(30, size=(T, K, K)).astype(np.float64) * (g.rand(T, K, K) < 0.7)
    beta = g.dirichlet(np.ones(K))
    if seed % 4 == 0:
        beta[0] = 1e-310                      # a dish whose weight has underflowed
    kappa = 4.0 * (seed % 3)
    r1, r2 = np.random.RandomState(seed), np.random.RandomState(seed)
    expect = _tables_cell_by_cell(n, beta, 0.7, 1.3, kappa, r1)
    got = hu.sample_tables(n, beta, 0.7, 1.3, kappa, r2)
    np.testing.assert_array_equal(got, expect)
    assert r1.random_sample() == r2.random_sample()       # same stream position afterwards


def test_native_table_draws_reject_invalid_probabilities():
    n = np.full((2, 2, 2), 3.0)
    with pytest.raises(ValueError):
        hu.sample_tables(n, np.array([0.0, 1.0]), 1.0, 1.0, 0.0, np.random.RandomState(0))
