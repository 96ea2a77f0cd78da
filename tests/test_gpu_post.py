"""Post-loop processing on the device against the oracle and the reference's
own outputs (tests/golden/post.npz).  Needs an MI355X: -m gpu.  Co-occurrence counts are
integers -> probabilities exact; VI values agree to 1e-12 and identical partitions tie
exactly (the tie-break by network log-likelihood then picks the reference's sample)."""
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import load_golden
from oracle import post_oracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    import dynetlsm_amd
    return dynetlsm_amd


@pytest.fixture(scope='module')
def g():
    return load_golden('post.npz')


def _model(g, tag, selection_type='vi'):
    directed = tag == 'd'
    return SimpleNamespace(
        Y_fit_=g[tag + '_Y'], zs_=g[tag + '_zs'], Xs_=g[tag + '_Xs'].copy(),
        intercepts_=g[tag + '_intercepts'], radiis_=g['d_radiis'] if directed else None,
        mus_=g[tag + '_mus'].copy(), sigmas_=g[tag + '_sigmas'], betas_=g[tag + '_betas'],
        weights_=g[tag + '_weights'], lambdas_=g[tag + '_lambdas'], logps_=g[tag + '_logps'],
        n_components=int(g[tag + '_K']), n_features=2, is_directed=directed,
        selection_type=selection_type)


def _chain(eng, m):
    T, N, _ = m.Y_fit_.shape
    c = eng.Chain(T, N, 2, 'directed' if m.is_directed else 'undirected')
    c.upload_network(m.Y_fit_)
    if m.is_directed:
        c.set_radii(m.radiis_[0])
    return c


@pytest.mark.parametrize('tag', ['u', 'd'])
def test_cooccurrence_and_vi_match_reference(eng, g, tag):
    from dynetlsm_amd import posterior as post
    m = _model(g, tag)
    n_burn = int(g[tag + '_n_burn'])
    with _chain(eng, m) as c:
        cooc = post.posterior_cooccurrences(m, c, n_burn)
        np.testing.assert_array_equal(cooc, g[tag + '_cooc'])
        best, vis = post.minimize_posterior_expected_vi(m, c, n_burn, cooc=cooc)
        np.testing.assert_allclose(vis, g[tag + '_vis'], rtol=1e-12)
        assert best == int(g[tag + '_best'])
        if tag == 'd':      # identical partitions tie exactly
            assert vis[12 - n_burn] == vis[20 - n_burn] == vis[33 - n_burn]


@pytest.mark.parametrize('tag', ['u', 'd'])
@pytest.mark.parametrize('selection_type', ['vi', 'bic', 'map'])
def test_select_model_matches_reference(eng, g, tag, selection_type):
    from dynetlsm_amd import posterior as post
    m = _model(g, tag, selection_type)
    n_burn = int(g[tag + '_n_burn'])
    with _chain(eng, m) as c:
        post.select_model(m, c, n_burn)
    np.testing.assert_array_equal(m.counts_, g[tag + '_counts'])
    np.testing.assert_allclose(m.bic_, g[tag + '_bic'], rtol=1e-9)
    for i, mod in enumerate(m.models_):
        np.testing.assert_allclose(mod.init_weights, g['%s_model%d_init_w' % (tag, i)], rtol=1e-13)
        np.testing.assert_allclose(mod.trans_weights, g['%s_model%d_trans_w' % (tag, i)], rtol=1e-13)
        np.testing.assert_allclose(mod.beta, g['%s_model%d_beta' % (tag, i)], rtol=1e-13)
    if selection_type == 'vi':
        assert m.selected_id_ == int(g[tag + '_best'])
        np.testing.assert_array_equal(m.z_, g[tag + '_z_r'])
        np.testing.assert_allclose(m.beta_, g[tag + '_beta_r'], rtol=1e-13)
        np.testing.assert_allclose(m.init_weights_, g[tag + '_init_w'], rtol=1e-13)
        np.testing.assert_allclose(m.trans_weights_, g[tag + '_trans_w'], rtol=1e-13)
        np.testing.assert_allclose(m.mu_, g[tag + '_mu_r'])
        np.testing.assert_allclose(m.sigma_, g[tag + '_sigma_r'])
    elif selection_type == 'bic':
        assert m.best_k_ == int(g[tag + '_bic'][np.argmin(g[tag + '_bic'][:, 1]), 0])
    else:
        assert m.best_k_ == int(np.argmax(np.bincount(g[tag + '_counts'])))
    ids, freqs = post.posterior_group_counts(m, n_burn)
    for t in range(m.zs_.shape[1]):
        np.testing.assert_array_equal(ids[t], g['%s_gc_index_%d' % (tag, t)])
        np.testing.assert_array_equal(freqs[t], g['%s_gc_freq_%d' % (tag, t)])


@pytest.mark.parametrize('T,N,S,K', [(2, 257, 70, 5), (1, 64, 3, 2), (3, 1000, 130, 20)])
def test_cooccurrence_and_vi_sizes(eng, T, N, S, K):
    """ragged tiles (N not a multiple of 64 / 16, S not a multiple of 64 or 4)"""
    rng = np.random.RandomState(N + S)
    zs = rng.randint(0, K, size=(S, T, N)).astype(np.int64)
    zs[S // 2] = zs[0]
    with eng.Chain(T, N, 2, 'undirected') as c:
        cooc = c.post_cooccurrence(zs, K)
        ref = po.posterior_cooccurrence(zs, 0, K)
        np.testing.assert_array_equal(cooc, ref)
        sums = c.post_expected_vi_sums()
        for s in (0, S // 2, S - 1):
            for t in range(T):
                eq = zs[s, t][:, None] == zs[s, t][None, :]
                want = np.log2((ref[t] * eq).sum(axis=1)).sum()
                np.testing.assert_allclose(sums[t, s], want, rtol=1e-12)
        assert (sums[:, 0] == sums[:, S // 2]).all()
        c.post_release()
        with pytest.raises(eng.EngineError):
            c.post_expected_vi_sums()
        with pytest.raises(eng.EngineError):
            c.post_cooccurrence(zs + K, K)            # labels out of range


@pytest.mark.parametrize('selection_type', ['vi', 'bic', 'map'])
def test_hdp_fit_selection_types(eng, selection_type):
    rng = np.random.RandomState(0)
    T, N = 3, 40
    Y = (rng.rand(T, N, N) < 0.15).astype(np.float64)
    Y = np.triu(Y, 1); Y = Y + Y.transpose(0, 2, 1)
    m = eng.DynamicNetworkHDPLPCM(n_iter=40, burn=20, tune=20, n_components=6,
                                  selection_type=selection_type, random_state=3).fit(Y)
    assert m.z_.shape == (T, N) and m.z_.min() == 0
    assert m.cooccurrence_probas_.shape == (T, N, N)
    assert np.allclose(np.diagonal(m.cooccurrence_probas_, axis1=1, axis2=2), 1.0)
    assert np.allclose(m.init_weights_.sum(), 1.0)
    assert m.mu_.shape[0] == np.unique(m.z_).shape[0]
