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


# ---- the same processing on the device-resident trace (hdp_lpcm.py:1085-1162) -------------------
def _trace_chain(eng, g, tag='u'):
    """a chain whose device-resident trace holds the golden file's synthetic stored samples"""
    from dynetlsm_amd import hdp_updates as hu
    Y, zs, Xs = g[tag + '_Y'], g[tag + '_zs'], g[tag + '_Xs']
    S, T, N = zs.shape
    K = int(g[tag + '_K'])
    c = eng.Chain(T, N, 2, 'undirected')
    c.upload_network(Y); c.set_positions(Xs[0]); c.set_intercepts([float(g[tag + '_intercepts'][0, 0])])
    c.set_samplers(eng.SamplerGrid(T, N, 0.1, tune=None))
    c.set_prior_mixture(g[tag + '_mus'][0], g[tag + '_sigmas'][0], float(g[tag + '_lambdas'][0, 0]), zs[0])
    hp = hu.HDPHyper(K)
    c.hdp_configure(hp, g[tag + '_betas'][0], g[tag + '_weights'][0], 0.5, 2.0)
    c.hdp_trace_alloc(S, logp0=float(g[tag + '_logps'][0]))
    c.hdp_trace_write(0, Xs=Xs, intercepts=g[tag + '_intercepts'][:, :1], logps=g[tag + '_logps'],
                      mus=g[tag + '_mus'], sigmas=g[tag + '_sigmas'], zs=zs, betas=g[tag + '_betas'],
                      weights=g[tag + '_weights'], lambdas=g[tag + '_lambdas'])
    return c


def test_trace_write_read_round_trip(eng, g):
    with _trace_chain(eng, g) as c:
        tr = c.hdp_trace_read(0, 40)
        np.testing.assert_array_equal(tr['Xs'], g['u_Xs'])
        np.testing.assert_array_equal(tr['zs'], g['u_zs'])
        np.testing.assert_array_equal(tr['weights'], g['u_weights'])
        np.testing.assert_array_equal(tr['mus'], g['u_mus'])
        np.testing.assert_array_equal(tr['logps'], g['u_logps'])
        np.testing.assert_array_equal(tr['intercepts'], g['u_intercepts'][:, :1])
        only = c.hdp_trace_read(3, 2, positions=False, labels=True, weights=False, small=False)
        assert sorted(only) == ['zs']
        np.testing.assert_array_equal(only['zs'], g['u_zs'][3:5])


def test_device_trace_counts_cooccurrence_and_vi_match_reference(eng, g):
    """label counts, co-occurrence matrices and their row sums, and the VI criterion of every
    kept sample, straight from the labels in HBM, against the reference's outputs"""
    n_burn, S = int(g['u_n_burn']), g['u_zs'].shape[0]
    zs = g['u_zs']
    with _trace_chain(eng, g) as c:
        nk = c.post_trace_label_counts(n_burn, S - n_burn)
        want = np.stack([[np.bincount(zs[s, t], minlength=c.K) for t in range(zs.shape[1])]
                         for s in range(n_burn, S)])
        np.testing.assert_array_equal(nk, want)
        cooc, rs = c.post_trace_cooccurrence(n_burn, S - n_burn, want_matrix=True)
        np.testing.assert_array_equal(cooc, g['u_cooc'])
        np.testing.assert_allclose(rs, g['u_cooc'].sum(axis=2), rtol=1e-13)
        np.testing.assert_array_equal(c.post_get_cooccurrence(), g['u_cooc'])
        from dynetlsm_amd import posterior as post
        vis = post.expected_vi(zs[n_burn:], rs, c.post_expected_vi_sums())
        np.testing.assert_allclose(vis, g['u_vis'], rtol=1e-12)


def test_device_forward_algorithm_matches_reference(eng, g):
    """approx_bic.py:54-76 (latent_marginal_loglikelihood) on the device: the reference's value at
    the selected sample, and the oracle's at other samples with all K components"""
    best = int(g['u_best'])
    with _trace_chain(eng, g) as c:
        got = c.post_latent_marginal_loglik(g['u_init_w'], g['u_trans_w'], g['u_mu_r'], g['u_sigma_r'],
                                            g['u_lambdas'][best], row=best)
        np.testing.assert_allclose(got, float(g['u_latent_marginal']), rtol=1e-11)
        rng = np.random.RandomState(4)
        K, T = int(g['u_K']), g['u_zs'].shape[1]
        for row in (0, 17):
            init_w = rng.dirichlet(np.ones(K))
            trans_w = rng.dirichlet(np.ones(K), size=(T, K))
            want = po.latent_marginal_loglikelihood(g['u_Xs'][row], init_w, trans_w, g['u_mus'][row],
                                                    g['u_sigmas'][row], float(g['u_lambdas'][row, 0]))
            got = c.post_latent_marginal_loglik(init_w, trans_w, g['u_mus'][row], g['u_sigmas'][row],
                                                g['u_lambdas'][row], row=row)
            np.testing.assert_allclose(got, want, rtol=1e-11)
        c.set_positions(g['u_Xs'][17])                   # row = -1: the chain's current positions
        got2 = c.post_latent_marginal_loglik(init_w, trans_w, g['u_mus'][17], g['u_sigmas'][17],
                                             g['u_lambdas'][17])
        assert got2 == got


def test_device_procrustes_of_every_sample_matches_reference(eng, g):
    """hdp_lpcm.py:1141-1149 on the device against the reference's own procrustes.py outputs
    (post.npz): every stored sample and its cluster means rotated onto the selected sample, then
    the posterior mean of the aligned positions"""
    best, n_burn, S = int(g['u_best']), int(g['u_n_burn']), g['u_zs'].shape[0]
    with _trace_chain(eng, g) as c:
        c.post_trace_align(0, S, best)
        tr = c.hdp_trace_read(0, S, labels=False, weights=False)
        np.testing.assert_allclose(tr['Xs'], g['u_Xs_aligned'], atol=1e-12)
        np.testing.assert_allclose(tr['mus'], g['u_mus_aligned'], atol=1e-12)
        np.testing.assert_allclose(tr['Xs'][best], g['u_Xs'][best], atol=1e-14)   # its own rotation: I
        np.testing.assert_allclose(c.post_trace_mean(n_burn, S - n_burn), g['u_X_mean'], atol=1e-13)


@pytest.mark.parametrize('D,N', [(1, 70), (3, 33), (4, 129)])
def test_device_procrustes_other_dimensions(eng, D, N):
    """d = 1, 3, 4 against scipy.linalg.orthogonal_procrustes (what procrustes.py:20-25 calls)"""
    from scipy.linalg import orthogonal_procrustes
    from dynetlsm_amd import hdp_updates as hu
    rng = np.random.RandomState(D * 100 + N)
    T, K, S = 2, 3, 9
    Xs = rng.randn(S, T, N, D)
    Y = (rng.rand(T, N, N) < 0.2).astype(np.float64); Y = np.triu(Y, 1); Y = Y + Y.transpose(0, 2, 1)
    mus = rng.randn(S, K, D)
    with eng.Chain(T, N, D, 'undirected') as c:
        c.upload_network(Y); c.set_positions(Xs[0]); c.set_intercepts([0.3])
        c.set_samplers(eng.SamplerGrid(T, N, 0.1, tune=None))
        c.set_prior_mixture(mus[0], np.ones(K), 0.8, rng.randint(0, K, size=(T, N)))
        c.hdp_configure(hu.HDPHyper(K), rng.dirichlet(np.ones(K)), rng.dirichlet(np.ones(K), size=(T, K)),
                        0.3, 2.0)
        c.hdp_trace_alloc(S, logp0=0.0)
        c.hdp_trace_write(0, Xs=Xs, mus=mus)
        c.post_trace_align(2, S - 2, 4)                  # rows 2 .. S-1 onto row 4
        tr = c.hdp_trace_read(0, S, labels=False, weights=False)
    ref = Xs[4].reshape(-1, D)
    for s in range(S):
        if s < 2:
            np.testing.assert_array_equal(tr['Xs'][s], Xs[s])
            continue
        R, _ = orthogonal_procrustes(Xs[s].reshape(-1, D), ref)
        np.testing.assert_allclose(tr['Xs'][s], Xs[s].reshape(-1, D).dot(R).reshape(T, N, D), atol=1e-11)
        np.testing.assert_allclose(tr['mus'][s], mus[s].dot(R), atol=1e-11)


@pytest.mark.parametrize('selection_type', ['vi', 'bic', 'map'])
def test_select_model_on_the_device_trace_matches_reference(eng, g, selection_type):
    from dynetlsm_amd import posterior as post
    n_burn = int(g['u_n_burn'])
    m = SimpleNamespace(Y_fit_=g['u_Y'], logps_=g['u_logps'], n_components=int(g['u_K']), n_features=2,
                        is_directed=False, selection_type=selection_type)
    with _trace_chain(eng, g) as c:
        post.select_model_device(m, c, n_burn)
        cooc = c.post_get_cooccurrence()
    np.testing.assert_array_equal(cooc, g['u_cooc'])
    np.testing.assert_array_equal(m.counts_, g['u_counts'])
    np.testing.assert_allclose(m.bic_, g['u_bic'], rtol=1e-9)
    for i, mod in enumerate(m.models_):
        np.testing.assert_allclose(mod.init_weights, g['u_model%d_init_w' % i], rtol=1e-13)
        np.testing.assert_allclose(mod.trans_weights, g['u_model%d_trans_w' % i], rtol=1e-13)
    if selection_type == 'vi':
        assert m.selected_id_ == int(g['u_best'])
        np.testing.assert_allclose(m.expected_vis_, g['u_vis'], rtol=1e-12)
        np.testing.assert_array_equal(m.z_, g['u_z_r'])
        np.testing.assert_allclose(m.trans_weights_, g['u_trans_w'], rtol=1e-13)
        np.testing.assert_allclose(m.mu_, g['u_mu_r'])
    ids, freqs = post.posterior_group_counts_from(m._counts_t)
    for t in range(g['u_zs'].shape[1]):
        np.testing.assert_array_equal(ids[t], g['u_gc_index_%d' % t])
        np.testing.assert_array_equal(freqs[t], g['u_gc_freq_%d' % t])


@pytest.mark.parametrize('selection_type', ['vi', 'bic'])
def test_fit_post_processing_on_device_equals_host(eng, selection_type):
    """the same chain (same seed: identical device-resident trace) finished on the device and on
    the host: same selection, same aligned samples, same posterior means; the large arrays of the
    device-finished fit reach the host only when they are read"""
    rng = np.random.RandomState(1)
    T, N = 3, 60
    Y = (rng.rand(T, N, N) < 0.12).astype(np.float64)
    Y = np.triu(Y, 1); Y = Y + Y.transpose(0, 2, 1)
    kw = dict(n_iter=30, burn=15, tune=15, n_components=5, selection_type=selection_type, random_state=7)
    a = eng.DynamicNetworkHDPLPCM(**kw).fit(Y)
    b = eng.DynamicNetworkHDPLPCM(post_processing='host', **kw)
    b._prepare(Y); b._run(1, b._n_total - 1); b.chain_.synchronize()
    b_Xs_raw = b.chain_.hdp_trace_read(0, b._n_total, labels=False, weights=False)['Xs']   # before alignment
    b._finish()
    assert a.loop_kind_ == b.loop_kind_ == 'device-resident'
    assert 'Xs_' not in a.__dict__ and 'zs_' not in a.__dict__ and 'weights_' not in a.__dict__
    assert 'Xs_' in b.__dict__
    assert a.selected_id_ == b.selected_id_
    np.testing.assert_array_equal(a.z_, b.z_)
    np.testing.assert_array_equal(a.counts_, b.counts_)
    np.testing.assert_allclose(a.bic_, b.bic_, rtol=1e-10)
    np.testing.assert_array_equal(a.logps_, b.logps_)
    np.testing.assert_allclose(a.X_, b.X_, atol=1e-14)
    np.testing.assert_allclose(a.X_mean_, b.X_mean_, atol=1e-12)
    np.testing.assert_allclose(a.mus_, b.mus_, atol=1e-11)
    np.testing.assert_allclose(a.lambda_mean_, b.lambda_mean_)
    if selection_type == 'vi':
        np.testing.assert_allclose(a.expected_vis_, b.expected_vis_, rtol=1e-12)
    for x, y in zip(a.posterior_group_counts_, b.posterior_group_counts_):
        np.testing.assert_array_equal(x, y)
    # the network log-likelihood the loop stored with every sample (the VI tie-break's input) is the
    # log-likelihood of the stored state
    for sid in (1, 17, 29):
        b.chain_.set_positions(b_Xs_raw[sid])
        np.testing.assert_allclose(a._trace_logliks[sid],
                                   b.chain_.loglik_full([b.intercepts_[sid]])[0], rtol=1e-12)
    assert np.isnan(a._trace_logliks[0])
    # on demand
    np.testing.assert_allclose(a.Xs_, b.Xs_, atol=1e-11)
    np.testing.assert_array_equal(a.zs_, b.zs_)
    np.testing.assert_array_equal(a.weights_, b.weights_)
    np.testing.assert_array_equal(a.cooccurrence_probas_, b.cooccurrence_probas_)
    assert 'Xs_' in a.__dict__ and 'cooccurrence_probas_' in a.__dict__


def test_lazy_trace_arrays_materialize_release_and_fail_as_attribute_errors(eng):
    """round-3 advice: vars() / copies see the lazily read arrays after materialize(); a read that
    cannot be served raises AttributeError (hasattr / getattr defaults keep working);
    release_device_trace() frees the handle and keeps what was read"""
    import copy
    rng = np.random.RandomState(2)
    T, N = 3, 40
    Y = (rng.rand(T, N, N) < 0.15).astype(np.float64)
    Y = np.triu(Y, 1); Y = Y + Y.transpose(0, 2, 1)
    kw = dict(n_iter=20, burn=10, tune=10, n_components=4, selection_type='map', random_state=3)
    m = eng.DynamicNetworkHDPLPCM(**kw).fit(Y)
    assert 'zs_' not in vars(m)
    m.materialize()
    for name in ('Xs_', 'zs_', 'weights_', 'cooccurrence_probas_'):
        assert name in vars(m), name
    zs = m.zs_.copy()
    m.release_device_trace()
    np.testing.assert_array_equal(m.zs_, zs)
    c = copy.copy(m)
    np.testing.assert_array_equal(c.zs_, zs)
    # a second fit that is released WITHOUT reading: the arrays are gone, and say so politely
    m2 = eng.DynamicNetworkHDPLPCM(**kw).fit(Y)
    m2.chain_.close()
    assert not hasattr(m2, 'zs_')
    assert getattr(m2, 'Xs_', None) is None
    with pytest.raises(AttributeError, match='device-resident trace'):
        m2.weights_
    # refitting resets the lazy co-occurrence flag
    m3 = eng.DynamicNetworkHDPLPCM(**kw)
    m3.fit(Y)
    assert m3._lazy_cooc
    m3._prepare(Y)
    assert not m3._lazy_cooc and not m3._lazy_trace
