"""Parity at BASELINE.json's full sizes (configs 2-4) on an MI355X: direct
comparison with the scalar C oracle where it finishes in seconds, and
size-independent properties (sum of partials = 2 x full; two independent sweep
algorithms agree; exhaustive controls = exact likelihood)."""
import time

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    import dynetlsm_amd
    return dynetlsm_amd


@pytest.fixture(scope='module')
def c2():
    from dynetlsm_amd.synthetic import synthetic_lsm_network
    return synthetic_lsm_network(T=10, N=2000, D=2, density=0.03, seed=0)


def test_c2_loglik_and_identity(eng, c2):
    Y, X, b = c2['Y'], c2['X_init'], c2['intercept']
    with eng.Chain(10, 2000, 2, 'undirected') as c:
        c.upload_network(Y); c.set_positions(X); c.set_intercepts([b])
        got = c.loglik_full([[b], [b + 0.3]])
        want = [orc.dynamic_network_loglikelihood_undirected(Y, X, v) for v in (b, b + 0.3)]
        np.testing.assert_allclose(got, want, rtol=1e-10)      # spec: 1e-6 relative
        pa = c.loglik_partial_all()
        np.testing.assert_allclose(pa.sum() / 2, want[0], rtol=1e-10)
        for t, j in [(0, 0), (9, 1999), (4, 1234)]:
            np.testing.assert_allclose(pa[t, j],
                                       orc.partial_loglikelihood(Y[t], X[t], b, j), rtol=1e-11)


def test_c2_sweep_algorithms_agree_and_match_oracle(eng, c2):
    """20 000 MH steps: slice sweep == speculative batches (algos 2, 3, 4) == C oracle"""
    Y, X, b = c2['Y'], c2['X_init'], c2['intercept']
    out = {}
    for algo in (1, 2, 3, 4):
        g = eng.SamplerGrid(10, 2000, 0.1, tune=5, tune_interval=1)
        with eng.Chain(10, 2000, 2, 'undirected', seed=99, chain_id=3) as c:
            c.upload_network(Y); c.set_positions(X); c.set_intercepts([b])
            c.set_prior_random_walk(2.0, 0.1); c.set_samplers(g)
            for it in (1, 2):
                c.sweep_positions(it, algo)
            out[algo] = (c.get_positions(), c.get_samplers(g).n_accepted.copy(),
                         g.step_size.copy())
    np.testing.assert_allclose(out[1][0], out[2][0], atol=1e-9)
    np.testing.assert_array_equal(out[1][1], out[2][1])
    np.testing.assert_allclose(out[1][0], out[3][0], atol=1e-9)
    np.testing.assert_array_equal(out[1][1], out[3][1])
    np.testing.assert_allclose(out[1][0], out[4][0], atol=1e-9)
    np.testing.assert_array_equal(out[1][1], out[4][1])
    og = orc.SamplerGrid(10, 2000, 0.1, tune=5, tune_interval=1)
    st = orc.ChainState(X, og, Y=Y, intercept=[b], tau_sq=2.0, sigma_sq=0.1, seed=99, chain=3)
    for it in (1, 2):
        st.c.iter = it
        st.sweep_c()
    np.testing.assert_allclose(out[2][0], st.X, atol=1e-9)
    np.testing.assert_array_equal(out[2][1], og.n_accepted)
    np.testing.assert_allclose(out[2][2], og.step_size, rtol=1e-13)
    moved = (st.X != X).any(axis=-1).mean()      # (counters reset while tuning)
    assert 0.05 < moved < 0.999


@pytest.mark.parametrize('algo', [0, 2])
def test_c3_hdp_pieces_at_full_size(eng, c2, algo):
    """config 3: T=10, N=2000, K_max=20: mixture-prior sweep - with the algorithm fit()
    picks at this size (algo 0 = auto = the pipelined sweep) and with the two-kernel
    speculative form - and the label update, against the C oracle"""
    Y, X, b = c2['Y'], c2['X_init'], c2['intercept']
    rng = np.random.RandomState(1)
    K = 20
    mu = rng.randn(K, 2) * 1.5; sigma = rng.uniform(0.3, 1.5, K)
    z = rng.randint(0, K, size=(10, 2000)).astype(np.int64)
    w = rng.dirichlet(np.ones(K), size=(10, K))
    og = orc.SamplerGrid(10, 2000, 0.1, tune=None)
    st = orc.ChainState(X, og, Y=Y, intercept=[b], mu=mu, sigma=sigma, lmbda=0.8, z=z,
                        seed=5, chain=0, it=1)
    st.sweep_c()
    with eng.Chain(10, 2000, 2, 'undirected', seed=5, chain_id=0) as c:
        c.upload_network(Y); c.set_positions(X); c.set_intercepts([b])
        c.set_prior_mixture(mu, sigma, 0.8, z)
        g = eng.SamplerGrid(10, 2000, 0.1, tune=None)
        c.set_samplers(g)
        if algo == 0:
            assert c.resolve_sweep_algo(0) == 4
        c.sweep_positions(1, algo)
        Xg = c.get_positions()
        np.testing.assert_allclose(Xg, st.X, atol=1e-9)
        np.testing.assert_array_equal(c.get_samplers(g).n_accepted, og.n_accepted)
        t0 = time.perf_counter()
        zg, n, nk = c.sample_labels(1, w)
        dt = time.perf_counter() - t0
    # the oracle draws its labels at the DEVICE's positions: every label must agree
    zo, no, nko = orc.sample_labels_block_philox(Xg, mu, sigma, 0.8, w, 5, 0, 1)
    np.testing.assert_array_equal(zg, zo)
    np.testing.assert_array_equal(n, no)
    np.testing.assert_array_equal(nk, nko)
    assert n.sum() == 20000 and (nk.sum(axis=1) == 2000).all()
    print('labels T=10 N=2000 K=20: %.2f ms incl. transfers' % (1e3 * dt))


def _sparse_directed(T, N, deg, seed):
    """directed edge lists built without a dense network (config 4 is sparse)"""
    from dynetlsm_amd.synthetic import synthetic_sparse_directed
    return synthetic_sparse_directed(T, N, deg, seed)


def test_c4_case_control_at_full_size(eng):
    """config 4: directed case-control, T=5, N=10 000, n_control=100"""
    T, N, C = 5, 10000, 100
    X, radii, degree, in_edges, out_edges = _sparse_directed(T, N, 20, 0)
    b = [1.0, 0.5]
    with eng.Chain(T, N, 2, 'case_control', seed=11, chain_id=1) as c:
        c.upload_edges(in_edges, out_edges, degree)
        t0 = time.perf_counter()
        c.resample_controls(0, C)
        t_res = time.perf_counter() - t0
        ci, co = c.get_controls()
        # validity on a sample of nodes (the full check is in test_gpu_parity)
        for t, i in [(0, 0), (4, 9999), (2, 5000), (1, 77)]:
            for arr, col, edges in ((co, 1, out_edges), (ci, 0, in_edges)):
                v = arr[t, i]
                assert (v >= 0).all() and len(set(v)) == C and i not in v
                assert not set(v) & set(edges[t, i, :degree[t, i, col]])
        c.set_positions(X); c.set_radii(radii); c.set_intercepts(b)
        got = c.loglik_full([b])[0]
        want = orc.approx_directed_network_loglikelihood(X, radii, in_edges, out_edges,
                                                         degree, co, b[0], b[1])
        np.testing.assert_allclose(got, want, rtol=1e-10)
        cc = dict(in_edges=in_edges, out_edges=out_edges, degree=degree,
                  control_nodes_in=ci, control_nodes_out=co)
        og = orc.SamplerGrid(T, N, 0.002, tune=None)
        st = orc.ChainState(X, og, model=2, intercept=b, radii=radii, case_control=cc,
                            tau_sq=1e-4, sigma_sq=1e-5, seed=11, chain=1, it=1)
        pa = c.loglik_partial_all()
        for t, j in [(0, 0), (4, 9999), (3, 4321)]:
            np.testing.assert_allclose(
                pa[t, j], orc.approx_directed_partial_loglikelihood(
                    X[t], radii, in_edges[t], out_edges[t], degree[t], ci[t], co[t],
                    b[0], b[1], j), rtol=1e-11)
        c.set_prior_random_walk(1e-4, 1e-5)
        c.set_samplers(eng.SamplerGrid(T, N, 0.002, tune=None))
        assert c.resolve_sweep_algo(0) == 5             # sparse correction lists at this size
        t0 = time.perf_counter()
        c.sweep_positions(1, 0)
        t_sw = time.perf_counter() - t0
        st.sweep_c()
        Xg = c.get_positions()
        np.testing.assert_allclose(Xg, st.X, atol=1e-12)
        assert 0.02 < og.n_accepted.mean() < 0.98
        # the dense-block form of the pipelined sweep (algo 4) makes the same decisions
        c.set_positions(X); c.set_samplers(eng.SamplerGrid(T, N, 0.002, tune=None))
        c.sweep_positions(1, 4)
        np.testing.assert_allclose(c.get_positions(), Xg, atol=1e-12)
    print('C4: resample %.1f ms, sweep %.1f ms' % (1e3 * t_res, 1e3 * t_sw))


def test_c2_init_pipeline_at_full_size(eng, c2):
    """SURVEY.md 8f-1 at N = 2000: hop matrices against the library the reference calls
    (scipy.sparse.csgraph, one slice) and their metric properties on all slices; SMACOF
    against the oracle over its first iterations and as a descent method over all of them;
    the Lanczos eigen step against a dense eigh of the explicit matrix."""
    from scipy.sparse import csgraph
    from oracle import init_oracle as io
    Y = c2['Y']
    T, N = Y.shape[:2]
    with eng.Chain(T, N, 2, 'undirected', seed=3) as c:
        c.upload_network(Y)
        c.init_shortest_paths()
        D0 = c.init_get_dissimilarity(0)
        ref = csgraph.shortest_path(Y[0], directed=False, unweighted=True)
        inf = np.isinf(ref)
        ref[inf] = ref[~inf].max() + 1
        np.testing.assert_array_equal(D0, ref)
        rng = np.random.RandomState(0)
        for t in (3, 9):
            Dt = c.init_get_dissimilarity(t)
            np.testing.assert_array_equal(Dt, Dt.T)
            assert (np.diag(Dt) == 0).all()
            np.testing.assert_array_equal(Dt == 1, Y[t] == 1)         # one hop = an edge
            i, j, k = rng.randint(0, N, size=(3, 20000))
            assert (Dt[i, k] <= Dt[i, j] + Dt[j, k]).all()            # triangle inequality
        X0 = rng.uniform(size=(2, N, 2))
        Xa, sa, na = c.init_smacof(0, X0, max_iter=4)
        for r in range(2):
            Xo, so, no = io.smacof_single(D0, X0[r], max_iter=4)
            assert na[r] == no
            np.testing.assert_allclose(sa[r], so, rtol=1e-10)
            assert np.abs(Xa[r] - Xo).max() < 1e-10 * np.abs(Xo).max()
        stresses = [c.init_smacof(0, X0[:1], max_iter=m, eps=0.0)[1][0] for m in (1, 5, 25, 125)]
        assert all(a > b for a, b in zip(stresses, stresses[1:]))     # majorisation descends
        Xb, sb, nb = c.init_smacof(0, X0)
        assert (sb <= stresses[-1] * 1.001).all()
        Xp = Xb[int(np.argmin(sb))]
        X1, evals, info = c.init_gmds_step(1, Xp)
        Xo, eo = io.gmds_step(c.init_get_dissimilarity(1), Xp)
        np.testing.assert_allclose(evals, eo, rtol=1e-10)
        assert np.abs(X1 - Xo).max() < 1e-9 * np.abs(Xo).max(), info


def test_directed_exact_at_full_size(eng):
    """T=10, N=2000 exact directed model: the running-product forms of the log-likelihood pass
    and of the pipelined evaluator against the C oracle and the per-slice kernel, and the
    device-resident loop's carried log-likelihood against a fresh evaluation of its last state."""
    from dynetlsm_amd.synthetic import synthetic_lsm_network
    net = synthetic_lsm_network(T=10, N=2000, D=2, density=0.03, seed=3)
    T, N, D = 10, 2000, 2
    rng = np.random.RandomState(7)
    Yd = (net['Y'] * (rng.rand(T, N, N) < 0.7)).astype(np.float64)      # asymmetric
    X = net['X_init'] * 0.01
    radii = rng.dirichlet(np.ones(N) * 5)
    b = np.array([0.4, 0.9])
    out = {}
    for algo in (1, 4):
        g = eng.SamplerGrid(T, N, 0.002, tune=None)
        with eng.Chain(T, N, D, 'directed', seed=21, chain_id=1) as c:
            c.upload_network(Yd); c.set_positions(X); c.set_radii(radii); c.set_intercepts(b)
            if algo == 1:
                got = c.loglik_full([b, b + 0.2])
                want = [orc.dynamic_network_loglikelihood_directed(Yd, X, v[0], v[1], radii)
                        for v in (b, b + 0.2)]
                np.testing.assert_allclose(got, want, rtol=1e-10)
            c.set_prior_random_walk(2.0, 0.1); c.set_samplers(g)
            for it in (1, 2):
                c.sweep_positions(it, algo)
            out[algo] = (c.get_positions(), c.get_samplers(g).n_accepted.copy())
    np.testing.assert_array_equal(out[1][1], out[4][1])
    np.testing.assert_allclose(out[1][0], out[4][0], atol=1e-12)
    assert 0.05 < out[4][1].mean() / 2 < 0.999
    # the device-resident loop: logp trace = fresh log-likelihood at the stored state + priors
    n_it = 6
    with eng.Chain(T, N, D, 'directed', seed=21, chain_id=1) as c:
        c.upload_network(Yd); c.set_positions(X); c.set_radii(radii); c.set_intercepts(b)
        c.set_prior_random_walk(2.0, 0.1)
        c.set_samplers(eng.SamplerGrid(T, N, 0.002, tune=None))
        c.lsm_configure([0.0, 0.0], 2.0, step_size_intercept=0.05, tune=None, sweep_algo=4)
        c.trace_alloc(n_it + 1)
        c.lsm_run(1, n_it)
        Xs, ics, lps = c.trace_read(0, n_it + 1)
        rad = c.trace_read_radii(0, n_it + 1)
        for it in (1, n_it):
            ll = orc.dynamic_network_loglikelihood_directed(Yd, Xs[it], ics[it, 0], ics[it, 1],
                                                            rad[it])
            x = Xs[it]
            prior = (-0.5 * np.sum(x[0] ** 2) / 2.0 -
                     0.5 * np.sum((x[1:] - x[:-1]) ** 2) / 0.1 -
                     0.5 * np.sum(ics[it] ** 2) / 2.0)
            np.testing.assert_allclose(lps[it], ll + prior, rtol=1e-10)


def test_pipelined_sweep_with_parts_longer_than_the_prefetch(eng):
    """T=12, N=2200: three parts of 768 neighbours per node, more than the 11 prefetched
    trips, so the pipelined sweep runs its long-part instantiation (software-pipelined tail);
    same decisions as the two-kernel speculative form."""
    from dynetlsm_amd.synthetic import synthetic_lsm_network
    T, N = 12, 2200
    net = synthetic_lsm_network(T=T, N=N, D=2, density=0.03, seed=5)
    out = {}
    for algo in (3, 4):
        g = eng.SamplerGrid(T, N, 0.1, tune=None)
        with eng.Chain(T, N, 2, 'undirected', seed=4, chain_id=0) as c:
            c.upload_network(net['Y']); c.set_positions(net['X_init'])
            c.set_intercepts([net['intercept']])
            c.set_prior_random_walk(2.0, 0.1); c.set_samplers(g)
            for it in (1, 2):
                c.sweep_positions(it, algo)
            out[algo] = (c.get_positions(), c.get_samplers(g).n_accepted.copy())
    np.testing.assert_array_equal(out[3][1], out[4][1])
    np.testing.assert_allclose(out[3][0], out[4][0], atol=1e-9)
    assert 0.2 < out[4][1].mean() / 2 < 0.999


@pytest.mark.parametrize('loop', ['device', 'host'])
def test_c3_facade_fit_logps_recomputed_by_the_oracle(eng, loop):
    """config 3 through the estimator facade: DynamicNetworkHDPLPCM(...).fit at T=10, N=2000,
    K_max=20 for a few iterations; the log-posterior trace is recomputed by the oracle
    (hdp_lpcm.py:1188-1280 restated) from the stored samples"""
    import torch  # noqa: F401
    from oracle import hdp_loop_oracle as hlo
    from dynetlsm_amd.synthetic import synthetic_hdp_network
    net = synthetic_hdp_network(T=10, N=2000, D=2, density=0.03, seed=0)
    K = 20
    rs = np.random.RandomState(5)
    mu0 = np.zeros((K, 2)); mu0[:6] = net['mu_true']; mu0[6:] = 3.0 * rs.randn(K - 6, 2)
    sg0 = np.full(K, float(net['sigma_true'].mean()))
    n_iter = 5
    m = eng.DynamicNetworkHDPLPCM(n_iter=n_iter, tune=None, burn=None, n_components=K,
                                  random_state=3, selection_type='map', hdp_loop=loop)
    m.fit(net['Y'], init=dict(X=net['X_init'], intercept=[net['intercept']], mu=mu0, sigma=sg0,
                              z=net['z_true']))
    assert m.loop_kind_ == ('device-resident' if loop == 'device' else 'host-driven')
    assert m.chain_.resolve_sweep_algo(0) == 4
    ip = np.atleast_1d(m.intercept_prior)
    its = range(1, n_iter) if loop == 'device' else [n_iter - 1]
    for it in its:
        hp = m.hyper_
        h = hlo.Hyper(a=hp.a, a0=hp.a0, b0=hp.b0, c0=hp.c0, d0=hp.d0, lambda_prior=hp.lambda_prior,
                      lambda_variance_prior=hp.lambda_variance_prior)
        if loop == 'device':            # the hyper-parameters of that iteration
            (h.gamma, h.alpha_init, h.alpha, h.kappa, h.mean_variance_prior, h.b) = m.hypers_[it]
        else:                           # the host loop keeps the last ones
            (h.gamma, h.alpha_init, h.alpha, h.kappa, h.mean_variance_prior, h.b) = (
                hp.gamma, hp.alpha_init, hp.alpha, hp.kappa,
                float(np.ravel(hp.mean_variance_prior)[0]), hp.b)
        ll = orc.dynamic_network_loglikelihood_undirected(net['Y'], m.Xs_[it], m.intercepts_[it, 0])
        want = hlo.log_posterior(ll, m.Xs_[it], m.intercepts_[it], m.mus_[it], m.sigmas_[it],
                                 m.zs_[it], m.weights_[it], m.betas_[it], m.lambdas_[it], h, ip,
                                 m.intercept_variance_prior)
        np.testing.assert_allclose(m.logps_[it], want, rtol=1e-9)
    assert (m.zs_[-1] != m.zs_[0]).any() and np.isfinite(m.logps_[1:]).all()
    np.testing.assert_allclose(m.weights_[-1].sum(-1)[1:], 1.0, rtol=1e-12)


def test_c3_device_loop_invariants_at_full_size(eng):
    """size-independent properties of the device-resident HDP-LPCM iteration at T=10, N=2000,
    K_max=20 (sample_auxillary.py / hdp_lpcm.py:876-1023): table counts between 1 and the
    customer counts, override counts below the diagonal tables, m_bar as defined, label counts
    that add up, distributions that are normalised, parameters in their supports"""
    from dynetlsm_amd import hdp_updates as hu
    from dynetlsm_amd.synthetic import synthetic_hdp_network
    T, N, K = 10, 2000, 20
    net = synthetic_hdp_network(T=T, N=N, D=2, density=0.03, seed=0)
    rs = np.random.RandomState(5)
    mu0 = np.zeros((K, 2)); mu0[:6] = net['mu_true']; mu0[6:] = 3.0 * rs.randn(K - 6, 2)
    sg0 = np.full(K, float(net['sigma_true'].mean()))
    beta = rs.dirichlet(np.ones(K))
    w = rs.dirichlet(np.ones(K), size=(T, K))
    hp = hu.HDPHyper(K, mean_variance_prior=40.0, b=160.0, a0=36.0, b0=2720.0, c0=16.0, d0=0.1)
    with eng.Chain(T, N, 2, 'undirected', seed=8, chain_id=1) as c:
        c.upload_network(net['Y']); c.set_positions(net['X_init'])
        c.set_intercepts([net['intercept']])
        c.set_samplers(eng.SamplerGrid(T, N, 0.1, tune=None))
        c.set_prior_mixture(mu0, sg0, 0.9, net['z_true'])
        c.hdp_configure(hp, beta, w, net['intercept'], 2.0)
        n_it = 4
        c.hdp_trace_alloc(n_it + 1)
        for it in range(1, n_it + 1):
            c.hdp_run(it, 1)
            a = c.hdp_get_aux()
            tr = c.hdp_trace_read(it, 1, positions=False)
            n, m, nk = a['n'], a['m'], a['nk']
            assert (nk.sum(axis=1) == N).all() and n[0, 0].sum() == N and (n[0, 1:] == 0).all()
            assert (n[1:].sum(axis=(1, 2)) == N).all()
            np.testing.assert_array_equal(n[1:].sum(axis=1), nk[1:])          # arrivals per label
            np.testing.assert_array_equal(n[1:].sum(axis=2), nk[:-1])         # departures per label
            assert (m <= n).all() and (m[n > 0] >= 1).all() and (m[n == 0] == 0).all()
            idx = np.arange(K)
            assert (a['w_over'] <= m[1:, idx, idx]).all() and (a['w_over'] >= 0).all()
            want = m[1:].sum(axis=(0, 1)) - a['w_over'].sum(axis=0) + m[0, 0]
            np.testing.assert_array_equal(a['m_bar'], want)
            np.testing.assert_allclose(tr['betas'][0].sum(), 1.0, rtol=1e-12)
            np.testing.assert_allclose(tr['weights'][0, 0, 0].sum(), 1.0, rtol=1e-12)
            np.testing.assert_allclose(tr['weights'][0, 1:].sum(axis=-1), 1.0, rtol=1e-12)
            assert (tr['sigmas'][0] > 0).all() and 0.0 < tr['lambdas'][0, 0] < 1.0
            assert (tr['hypers'][0] > 0).all() and np.isfinite(tr['logps'][0])
            np.testing.assert_array_equal(np.bincount(tr['zs'][0].ravel(), minlength=K),
                                          nk.sum(axis=0))


def test_c3_device_loop_matches_oracle_at_full_size(eng):
    """config 3 (T=10, N=2000, K_max=20): two iterations of the device-resident HDP-LPCM loop
    against the oracle's restatement of hdp_lpcm.py:876-1023 with the engine's Philox draws,
    value for value - labels, label and table counts (cells with ~10^3 customers), override
    variables and m-bar exactly; positions, beta, w, mu, sigma^2, lambda, the six resampled
    hyper-parameters and the log-posterior to the tolerances of tests/test_gpu_hdp_loop.py"""
    from test_gpu_hdp_loop import _run_both
    out = _run_both(eng, 10, 2000, 20, seed=21, n_it=2, algo=0)
    tr, lp = out[-1]
    assert np.isfinite(lp) and len(np.unique(tr['zs'][0])) >= 2
