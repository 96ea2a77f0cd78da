"""Parity of the HIP engine (through the C-ABI) against the CPU oracle and the
golden vectors captured from the reference.  Needs an MI355X: -m gpu.

Tolerances: BASELINE.json asks for the log-likelihood to 1e-6 relative; the
engine computes in float64, so the tests hold it to 1e-10 relative (full
sums) / 1e-9 absolute on positions after several sweeps."""
import numpy as np
import pytest

from conftest import LIK_TAGS

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

RTOL_LL = 1e-10


@pytest.fixture(scope='module')
def eng():
    import dynetlsm_amd
    from dynetlsm_amd import _lib
    _lib.load()
    assert _lib.device_count() >= 1, 'no HIP device: the engine has no CPU path'
    return dynetlsm_amd


def _rand_net(seed, T, N, D=2, density=0.2, scale=1.0):
    rng = np.random.RandomState(seed)
    X = rng.randn(T, N, D) * scale
    Yd = (rng.rand(T, N, N) < density).astype(np.float64)
    for t in range(T):
        np.fill_diagonal(Yd[t], 0)
    Yu = np.triu(Yd, 1)
    Yu = Yu + Yu.transpose(0, 2, 1)
    radii = rng.dirichlet(np.ones(N) * 5)
    return X, Yd, Yu, radii


def _cc_lists(Yd, n_control, seed):
    """edge lists as the reference builds them + valid random controls"""
    T, N, _ = Yd.shape
    deg, ie, oe = orc.case_control_init(Yd)
    rng = np.random.RandomState(seed)
    ci = np.full((T, N, n_control), -1, dtype=np.int64)
    co = np.full((T, N, n_control), -1, dtype=np.int64)
    for t in range(T):
        for i in range(N):
            zo = np.setdiff1d(np.arange(N), np.append(oe[t, i, :deg[t, i, 1]], i))
            zi = np.setdiff1d(np.arange(N), np.append(ie[t, i, :deg[t, i, 0]], i))
            k = min(n_control, zo.size)
            co[t, i, :k] = rng.choice(zo, k, replace=False)
            k = min(n_control, zi.size)
            ci[t, i, :k] = rng.choice(zi, k, replace=False)
    return dict(in_edges=ie, out_edges=oe, degree=deg, control_nodes_in=ci,
                control_nodes_out=co)


# ------------------------------------------------------------ function seam
@pytest.mark.parametrize('tag', LIK_TAGS)
@pytest.mark.parametrize('sq', [0, 1])
def test_function_seam_against_reference_goldens(eng, golden_lik, tag, sq):
    """the reference's own outputs, through the reference's own signatures"""
    nl = eng.network_likelihoods
    g = golden_lik
    X, Yd, Yu, radii = g[tag + '_X'], g[tag + '_Yd'], g[tag + '_Yu'], g[tag + '_radii']
    b, b_in, b_out = g[tag + '_b']
    T, N, D = X.shape
    for t, j in [(0, 0), (T - 1, N - 1), (0, N // 2)]:
        got = nl.partial_loglikelihood(Yu[t], X[t], b, j, squared=bool(sq))
        np.testing.assert_allclose(got, g['%s_partial_undirected_sq%d' % (tag, sq)][t, j],
                                   rtol=1e-12)
        got = nl.directed_partial_loglikelihood(Yd[t].copy(), X[t].copy(), radii, b_in,
                                                b_out, j, squared=bool(sq))
        np.testing.assert_allclose(got, g['%s_partial_directed_sq%d' % (tag, sq)][t, j],
                                   rtol=1e-12)
    got = nl.dynamic_network_loglikelihood_undirected(Yu, X, b, squared=bool(sq))
    np.testing.assert_allclose(got, g['%s_full_undirected_sq%d' % (tag, sq)], rtol=RTOL_LL)
    got = nl.dynamic_network_loglikelihood_directed(Yd, X, b_in, b_out, radii,
                                                    squared=bool(sq))
    np.testing.assert_allclose(got, g['%s_full_directed_sq%d' % (tag, sq)], rtol=RTOL_LL)
    got = nl.approx_directed_network_loglikelihood(
        X, radii, g[tag + '_in_edges'], g[tag + '_out_edges'], g[tag + '_degrees'],
        g[tag + '_ctrl_out'], b_in, b_out, squared=bool(sq))
    np.testing.assert_allclose(got, g['%s_full_approx_sq%d' % (tag, sq)], rtol=RTOL_LL)
    # a3 where the reference's sentinel bug is harmless (n_in == n_out)
    ci, co = g[tag + '_ctrl_in'], g[tag + '_ctrl_out']
    ok = np.argwhere((ci >= 0).sum(2) == (co >= 0).sum(2))
    for t, j in ok[:4]:
        got = nl.approx_directed_partial_loglikelihood(
            X[t], radii, g[tag + '_in_edges'][t], g[tag + '_out_edges'][t],
            g[tag + '_degrees'][t], ci[t], co[t], b_in, b_out, int(j), squared=bool(sq))
        np.testing.assert_allclose(got, g['%s_partial_approx_sq%d' % (tag, sq)][t, j],
                                   rtol=1e-12)
    for i in (0, N - 1):
        for nz in (0, 1):
            got = nl.compute_gaussian_likelihood(X[:, i].copy(), g[tag + '_mu'],
                                                 g[tag + '_sigma'], 0.8, normalize=bool(nz))
            np.testing.assert_allclose(got, g['%s_gauss_norm%d' % (tag, nz)][i], rtol=1e-12)


def test_function_seam_dtype_rules(eng):
    nl = eng.network_likelihoods
    X, Yd, Yu, radii = _rand_net(0, 1, 8)
    with pytest.raises(ValueError):
        nl.partial_loglikelihood(Yu[0].astype(np.float32), X[0], 0.1, 0)
    with pytest.raises(ValueError):
        nl.directed_partial_loglikelihood(Yd[0].T, X[0], radii, 0.1, 0.2, 0)


def test_directed_large_exponent_corner(eng):
    """A negative in-intercept with small radii makes a = b_in / r_j + b_out / r_i negative, so
    eta grows with the distance and passes 40, where the running-product form of the directed
    kernels (log-likelihood pass, pipelined evaluator and its H entries) hands over to the
    term-by-term form; both forms within one evaluation here."""
    T, N, D = 2, 600, 2
    X, Yd, Yu, radii = _rand_net(123, T, N, D, scale=0.02)
    b_in, b_out = -2.0, 0.7
    with eng.Chain(T, N, D, 'directed', seed=5) as c:
        c.upload_network(Yd); c.set_positions(X); c.set_radii(radii)
        c.set_intercepts([b_in, b_out])
        # the corner is really exercised: some exponents above 40, most below
        d = np.sqrt(((X[0][:, None] - X[0][None]) ** 2).sum(-1))
        eta = b_in * (1 - d / radii[None, :]) + b_out * (1 - d / radii[:, None])
        assert (eta > 40).mean() > 0.01 and (eta < 40).mean() > 0.2
        got = c.loglik_full([[b_in, b_out], [0.2, 0.4]])
        want = [orc.dynamic_network_loglikelihood_directed(Yd, X, a, b, radii)
                for a, b in ((b_in, b_out), (0.2, 0.4))]
        np.testing.assert_allclose(got, want, rtol=RTOL_LL)
        # sweeps: the pipelined form against the per-slice kernel (same decisions).  This corner
        # drives log-ratios of single-node moves beyond +-700, where exp() of the pipelined form's
        # multiplicative accept test would saturate: those nodes are resolved in the log domain
        # (pipe_resolve), so the two forms keep agreeing sweep after sweep.
        c.set_prior_random_walk(2.0, 0.1)
        big = 0
        res47 = {}
        for algo in (4, 1):
            c.set_positions(X)
            c.set_samplers(eng.SamplerGrid(T, N, 0.02, tune=None))
            res = []
            for it in (1, 2, 3, 4):
                if algo == 1:
                    pa = c.loglik_partial_all(with_prior=True)
                c.sweep_positions(it, algo)
                res.append((c.get_positions(),
                            c.get_samplers(eng.SamplerGrid(T, N, 0.02, tune=None)).n_accepted.copy()))
                if algo == 1:       # how large do single-node log-ratios get here?
                    big = max(big, float(np.abs(c.loglik_partial_all(with_prior=True) - pa).max()))
            if algo != 1:
                res47[algo] = res
        for algo in (4,):
            for (X4, a4), (X1, a1) in zip(res47[algo], res):
                np.testing.assert_array_equal(a4, a1)
                np.testing.assert_allclose(X4, X1, atol=1e-12)
        assert big > 700.0, big         # the corner really is beyond the exponent's range


# ------------------------------------------------------------ full log-lik
@pytest.mark.parametrize('N', [7, 128, 129, 300])
@pytest.mark.parametrize('D', [1, 2, 3, 4])
def test_loglik_full_undirected(eng, N, D):
    X, Yd, Yu, radii = _rand_net(N * 10 + D, 3, N, D)
    cands = np.array([[0.75], [-1.3], [2.0]])
    with eng.Chain(3, N, D, 'undirected') as c:
        c.upload_network(Yu); c.set_positions(X); c.set_intercepts([0.75])
        for sq in (0, 1):
            c.set_squared(sq)
            got = c.loglik_full(cands)
            want = [orc.dynamic_network_loglikelihood_undirected(Yu, X, b, squared=sq)
                    for b in cands[:, 0]]
            np.testing.assert_allclose(got, want, rtol=RTOL_LL)
            np.testing.assert_allclose(c.loglik_full(), want[0], rtol=RTOL_LL)
            # sum of partials = 2 x full
            np.testing.assert_allclose(c.loglik_partial_all().sum() / 2, want[0],
                                       rtol=RTOL_LL)


@pytest.mark.parametrize('N', [7, 130, 300])
@pytest.mark.parametrize('D', [2, 3])
def test_loglik_full_directed(eng, N, D):
    X, Yd, Yu, radii = _rand_net(N * 7 + D, 2, N, D, scale=0.05)
    cands = np.array([[0.3, 0.7], [1.0, -0.2], [0.1, 0.1]])
    with eng.Chain(2, N, D, 'directed') as c:
        c.upload_network(Yd); c.set_positions(X); c.set_radii(radii)
        c.set_intercepts(cands[0])
        got = c.loglik_full(cands)
        want = [orc.dynamic_network_loglikelihood_directed(Yd, X, a, b, radii)
                for a, b in cands]
        np.testing.assert_allclose(got, want, rtol=RTOL_LL)
        pa = c.loglik_partial_all()
        wantp = np.array([[orc.directed_partial_loglikelihood(Yd[t], X[t], radii, 0.3,
                                                              0.7, j)
                           for j in range(N)] for t in range(2)])
        np.testing.assert_allclose(pa, wantp, rtol=1e-11)
        np.testing.assert_allclose(pa.sum() / 2, want[0], rtol=RTOL_LL)
        r2 = np.random.RandomState(1).dirichlet(np.ones(N) * 3)
        got2 = c.loglik_full_radii(r2)
        np.testing.assert_allclose(got2[0], want[0], rtol=RTOL_LL)
        np.testing.assert_allclose(
            got2[1], orc.dynamic_network_loglikelihood_directed(Yd, X, 0.3, 0.7, r2),
            rtol=RTOL_LL)


@pytest.mark.parametrize('N,C', [(12, 3), (200, 70), (200, 300)])
def test_loglik_case_control(eng, N, C):
    X, Yd, Yu, radii = _rand_net(N + C, 2, N, 2, density=0.05, scale=0.05)
    cc = _cc_lists(Yd, C, 3)
    with eng.Chain(2, N, 2, 'case_control') as c:
        c.upload_edges(cc['in_edges'], cc['out_edges'], cc['degree'])
        c.set_controls(cc['control_nodes_in'], cc['control_nodes_out'])
        c.set_positions(X); c.set_radii(radii); c.set_intercepts([0.3, 0.7])
        got = c.loglik_full([[0.3, 0.7], [0.9, 0.1]])
        want = [orc.approx_directed_network_loglikelihood(
            X, radii, cc['in_edges'], cc['out_edges'], cc['degree'],
            cc['control_nodes_out'], a, b) for a, b in [(0.3, 0.7), (0.9, 0.1)]]
        np.testing.assert_allclose(got, want, rtol=RTOL_LL)
        pa = c.loglik_partial_all()
        wantp = np.array([[orc.approx_directed_partial_loglikelihood(
            X[t], radii, cc['in_edges'][t], cc['out_edges'][t], cc['degree'][t],
            cc['control_nodes_in'][t], cc['control_nodes_out'][t], 0.3, 0.7, j)
            for j in range(N)] for t in range(2)])
        np.testing.assert_allclose(pa, wantp, rtol=1e-11)
        if C >= N:   # exhaustive controls: the estimator is exact
            exact = orc.dynamic_network_loglikelihood_directed(Yd, X, 0.3, 0.7, radii)
            np.testing.assert_allclose(got[0], exact, rtol=1e-10)


@pytest.mark.parametrize('N,C,D', [(9, 3, 2), (130, 20, 1), (130, 20, 3), (300, 70, 5), (300, 70, 8), (700, 150, 2),
                                   (200, 300, 2)])
def test_loglik_case_control_pass_forms_agree(eng, monkeypatch, N, C, D):
    """directed_likelihoods_fast.pyx:208-270 through the three forms of the pass - the streaming wavefronts with the
    reciprocal radii in LDS (default), with gathered records (DLSM_CC_PASS=records), and two rows per wavefront
    (rows; round 5's kernel): one, two and the radii step's two candidates; every form against the oracle, the forms
    against each other to rounding.  N = 700 with 150 controls: rows beyond 128 out-terms (two entries of the walking
    order); C >= N: control lists with invalid (-1) slots"""
    X, Yd, Yu, radii = _rand_net(N + C + D, 2, N, D, density=0.08, scale=0.05)
    cc = _cc_lists(Yd, C, 5)
    r2 = np.random.RandomState(7).dirichlet(np.ones(N) * 5)
    got = {}
    for form in ('', 'records', 'rows'):
        if form:
            monkeypatch.setenv('DLSM_CC_PASS', form)
        else:
            monkeypatch.delenv('DLSM_CC_PASS', raising=False)
        with eng.Chain(2, N, D, 'case_control') as c:
            c.upload_edges(cc['in_edges'], cc['out_edges'], cc['degree'])
            c.set_controls(cc['control_nodes_in'], cc['control_nodes_out'])
            c.set_positions(X); c.set_radii(radii); c.set_intercepts([0.3, 0.7])
            got[form] = (c.loglik_full([[0.3, 0.7]]), c.loglik_full([[0.3, 0.7], [0.9, 0.1]]),
                         c.loglik_full_radii(r2))
    want = [orc.approx_directed_network_loglikelihood(X, rr, cc['in_edges'], cc['out_edges'], cc['degree'],
                                                      cc['control_nodes_out'], a, b)
            for rr, a, b in [(radii, 0.3, 0.7), (radii, 0.9, 0.1), (r2, 0.3, 0.7)]]
    for form, (g1, g2, g3) in got.items():
        np.testing.assert_allclose(g1, want[:1], rtol=RTOL_LL, err_msg=form)
        np.testing.assert_allclose(g2, want[:2], rtol=RTOL_LL, err_msg=form)
        np.testing.assert_allclose(g3, [want[0], want[2]], rtol=RTOL_LL, err_msg=form)
    for form in ('records', 'rows'):
        for a, b in zip(got[''], got[form]):
            np.testing.assert_allclose(a, b, rtol=1e-12, err_msg=form)


def test_partial_with_explicit_position_and_prior(eng):
    X, Yd, Yu, radii = _rand_net(5, 3, 40)
    grid = orc.SamplerGrid(3, 40)
    st = orc.ChainState(X, grid, Y=Yu, intercept=[0.2], tau_sq=2.0, sigma_sq=0.1)
    with eng.Chain(3, 40, 2, 'undirected') as c:
        c.upload_network(Yu); c.set_positions(X); c.set_intercepts([0.2])
        c.set_prior_random_walk(2.0, 0.1)
        x = np.array([0.3, -0.8])
        for t, j in [(0, 3), (1, 39), (2, 0)]:
            np.testing.assert_allclose(c.loglik_partial(t, j, x, with_prior=True),
                                       st.node_logp(t, j, x), rtol=1e-12)


def test_bad_inputs_fail_loudly(eng):
    X, Yd, Yu, radii = _rand_net(1, 2, 10)
    with eng.Chain(2, 10, 2, 'undirected') as c:
        with pytest.raises(eng.EngineError):
            c.loglik_full([[0.1]])                 # nothing uploaded yet
        Ybad = Yu.copy(); Ybad[0, 1, 2] = -1.0     # missing-edge code
        with pytest.raises(eng.EngineError) as e:
            c.upload_network(Ybad)
        assert e.value.code == -4
        c.upload_network(Yu); c.set_positions(X)
        with pytest.raises(eng.EngineError):
            c.sweep_positions(1)                   # samplers / prior missing
    with pytest.raises(ValueError, match='1 <= n_features <= 8'):
        eng.Chain(2, 10, 9, 'undirected')          # unsupported n_features: named before any device call


# ------------------------------------------------------------ sweep
def _sweep_case(eng, name, prior, T, N, D, n_sweeps, algo, seed=11, cc_C=4,
                density=0.2, scale=1.0):
    X, Yd, Yu, radii = _rand_net(seed, T, N, D, density=density, scale=scale)
    rng = np.random.RandomState(seed + 1)
    K = 3
    mu = rng.randn(K, D) * scale; sigma = rng.uniform(0.5, 1.5, K) * scale ** 2
    z = rng.randint(0, K, size=(T, N)).astype(np.int64)
    okw = dict(tau_sq=2.0, sigma_sq=0.1)
    if prior == 'mix':
        okw = dict(mu=mu, sigma=sigma, lmbda=0.8, z=z)
    model = {'undirected': 0, 'directed': 1, 'case_control': 2}[name]
    cc = None
    if name == 'undirected':
        okw.update(Y=Yu, intercept=[0.5])
    elif name == 'directed':
        okw.update(Y=Yd, intercept=[0.3, 0.7], radii=radii)
    else:
        cc = _cc_lists(Yd, cc_C, seed)
        okw.update(intercept=[0.3, 0.7], radii=radii, case_control=cc)
    step = 0.2 * scale
    og = orc.SamplerGrid(T, N, step, tune=5, tune_interval=2)
    gg = eng.SamplerGrid(T, N, step, tune=5, tune_interval=2)
    st = orc.ChainState(X, og, model=model, seed=0xC0FFEE1234, chain=2, **okw)
    with eng.Chain(T, N, D, name, seed=0xC0FFEE1234, chain_id=2) as c:
        if cc is None:
            c.upload_network(okw['Y'])
        else:
            c.upload_edges(cc['in_edges'], cc['out_edges'], cc['degree'])
            c.set_controls(cc['control_nodes_in'], cc['control_nodes_out'])
        c.set_positions(X); c.set_intercepts(okw['intercept'])
        if name != 'undirected':
            c.set_radii(radii)
        if prior == 'mix':
            c.set_prior_mixture(mu, sigma, 0.8, z)
        else:
            c.set_prior_random_walk(2.0, 0.1)
        c.set_samplers(gg)
        for it in range(1, n_sweeps + 1):
            c.sweep_positions(it, algo=algo)
            st.c.iter = it
            st.sweep_c()
            np.testing.assert_allclose(c.get_positions(), st.X, rtol=0,
                                       atol=1e-9 * max(scale, 1e-3))
        c.get_samplers(gg)
    np.testing.assert_allclose(gg.step_size, og.step_size, rtol=1e-13)
    np.testing.assert_array_equal(gg.n_accepted, og.n_accepted)
    np.testing.assert_array_equal(gg.n_steps, og.n_steps)
    np.testing.assert_array_equal(gg.steps_until_tune, og.steps_until_tune)
    assert 0 < og.n_steps.sum()


@pytest.mark.parametrize('prior', ['rw', 'mix'])
@pytest.mark.parametrize('name', ['undirected', 'directed', 'case_control'])
def test_sweep_slice_small(eng, name, prior):
    _sweep_case(eng, name, prior, T=3, N=10, D=2, n_sweeps=6, algo=1,
                scale=1.0 if name == 'undirected' else 0.05)


@pytest.mark.parametrize('name,N,D', [('undirected', 300, 2), ('undirected', 1100, 2),
                                      ('directed', 260, 2), ('undirected', 70, 3),
                                      ('case_control', 300, 2), ('undirected', 33, 1)])
def test_sweep_slice_medium(eng, name, N, D):
    """crosses the proposal-chunk (256) and the workgroup (1024) boundaries;
    T=4 and T=1 exercise both parities and the single-slice edge case"""
    _sweep_case(eng, name, 'rw', T=4, N=N, D=D, n_sweeps=3, algo=1,
                scale=1.0 if name == 'undirected' else 0.05,
                cc_C=20, density=0.05 if name == 'case_control' else 0.2)


def test_sweep_squared_distances(eng):
    """squared=True (static_network_fast.pyx:37-38) through every sweep kernel (the pipelined
    evaluator and the log-likelihood pass have their own squared-distance loops)"""
    from dynetlsm_amd import Chain, SamplerGrid
    X, Yd, Yu, radii = _rand_net(77, 3, 200, 2)
    for algo in (1, 2, 3, 4):
        og = orc.SamplerGrid(3, 200, 0.1, tune=None)
        st = orc.ChainState(X, og, Y=Yu, intercept=[0.5], squared=True, tau_sq=2.0,
                            sigma_sq=0.1, seed=8, chain=0)
        with Chain(3, 200, 2, 'undirected', seed=8) as c:
            c.upload_network(Yu); c.set_positions(X); c.set_intercepts([0.5])
            c.set_squared(True); c.set_prior_random_walk(2.0, 0.1)
            c.set_samplers(SamplerGrid(3, 200, 0.1, tune=None))
            for it in (1, 2):
                c.sweep_positions(it, algo)
                st.c.iter = it
                st.sweep_c()
            np.testing.assert_allclose(c.get_positions(), st.X, atol=1e-9)
            np.testing.assert_allclose(
                c.loglik_full([[0.5]])[0],
                orc.dynamic_network_loglikelihood_undirected(Yu, st.X, 0.5, squared=True),
                rtol=1e-10)


def test_squared_distances_far_apart(eng):
    """positions thousands of units apart under squared=True: -d^2 is far below the range in which
    the table exponential reads its integer part (device_common.hpp: tab_exp_clamped); e^{-d^2}
    must come out as 0, not as a wrapped power of two - log-likelihood and pipelined sweep"""
    from dynetlsm_amd import Chain, SamplerGrid
    X, Yd, Yu, radii = _rand_net(5, 2, 150, 2)
    X = X.copy()
    X[:, 75:] += 4000.0            # two groups 5.6e3 apart: squared distances of 3e7
    with Chain(2, 150, 2, 'undirected', seed=3) as c:
        c.upload_network(Yu); c.set_positions(X); c.set_intercepts([0.5]); c.set_squared(True)
        got = c.loglik_full([[0.5], [1.5]])
        want = [orc.dynamic_network_loglikelihood_undirected(Yu, X, b, squared=True) for b in (0.5, 1.5)]
        assert np.all(np.isfinite(got))
        np.testing.assert_allclose(got, want, rtol=RTOL_LL)
        c.set_prior_random_walk(2.0, 0.1)
        c.set_samplers(SamplerGrid(2, 150, 0.1, tune=None))
        og = orc.SamplerGrid(2, 150, 0.1, tune=None)
        st = orc.ChainState(X, og, Y=Yu, intercept=[0.5], squared=True, tau_sq=2.0, sigma_sq=0.1,
                            seed=3, chain=0)
        for it in (1, 2):
            c.sweep_positions(it, 4)
            st.c.iter = it
            st.sweep_c()
        assert np.all(np.isfinite(c.get_positions()))
        np.testing.assert_allclose(c.get_positions(), st.X, atol=1e-9)


def test_sweep_single_time_step(eng):
    _sweep_case(eng, 'undirected', 'rw', T=1, N=20, D=2, n_sweeps=3, algo=1)
    _sweep_case(eng, 'undirected', 'rw', T=1, N=150, D=2, n_sweeps=3, algo=2)
    _sweep_case(eng, 'undirected', 'rw', T=1, N=300, D=2, n_sweeps=3, algo=4)     # no odd slices


def test_sweep_pipelined_many_slices(eng):
    """more slices than a launch has evaluator wavefronts per node part: parts clamps to 1 and
    the resolver workgroups outnumber a handful of CUs (nothing in a launch waits on another
    workgroup, so over-subscription is harmless)"""
    _sweep_case(eng, 'undirected', 'mix', T=37, N=140, D=2, n_sweeps=2, algo=4)
    _sweep_case(eng, 'directed', 'rw', T=300, N=130, D=2, n_sweeps=1, algo=4, scale=0.05)


@pytest.mark.parametrize('prior', ['rw', 'mix'])
@pytest.mark.parametrize('name,T,N,D', [('undirected', 4, 10, 2), ('undirected', 4, 300, 2),
                                        ('undirected', 3, 1100, 2), ('directed', 4, 260, 2),
                                        ('directed', 3, 128, 3), ('undirected', 5, 129, 1),
                                        ('undirected', 2, 257, 4)])
@pytest.mark.parametrize('algo', [2, 3, 4])
def test_sweep_speculative_batches(eng, name, T, N, D, prior, algo):
    """algo 2 / 3 (chip-wide speculative batches) and 4 (the pipelined form: batch b + 1
    evaluated beside the resolve of batch b) are the same Gauss-Seidel scan:
    identical decisions, positions equal to rounding.  N = 10 < one batch,
    129 / 257 / 300 leave ragged last batches."""
    _sweep_case(eng, name, prior, T=T, N=N, D=D, n_sweeps=3, algo=algo,
                scale=1.0 if name == 'undirected' else 0.05)


@pytest.mark.parametrize('prior', ['rw', 'mix'])
@pytest.mark.parametrize('T,N,C,density', [(4, 300, 20, 0.05), (3, 10, 3, 0.2),
                                           (2, 150, 40, 0.5), (5, 700, 10, 0.02)])
@pytest.mark.parametrize('algo', [2, 4, 5])
def test_sweep_speculative_batches_case_control(eng, T, N, C, density, prior, algo):
    """sparse H: only the batch nodes that sit in a node's edge / control lists
    interact; density 0.5 with 40 controls makes most of a batch interact.  algo 5 keeps the
    corrections as per-node lists (batches of 512: one or two batches at these sizes)"""
    _sweep_case(eng, 'case_control', prior, T=T, N=N, D=2, n_sweeps=3, algo=algo,
                scale=0.05, cc_C=C, density=density)


@pytest.mark.parametrize('T,N,C,density,prior', [(3, 2300, 12, 0.004, 'rw'), (2, 1025, 30, 0.01, 'mix'),
                                                 (4, 3100, 6, 0.003, 'rw'), (1, 2100, 8, 0.004, 'rw')])
def test_sweep_case_control_sparse_lists_over_several_batches(eng, T, N, C, density, prior):
    """algo 5 with 3 - 7 batches of 512 nodes per slice (ragged last batch, a single slice,
    both priors): cross-batch corrections through the lists, same decisions as the oracle"""
    _sweep_case(eng, 'case_control', prior, T=T, N=N, D=2, n_sweeps=2, algo=5,
                scale=0.05, cc_C=C, density=density)


@pytest.mark.parametrize('D', [1, 3, 4])
def test_sweep_case_control_sparse_lists_other_dimensions(eng, D):
    """algo 5 with d = 1, 3, 4 (packed records of 4 / 4 / 8 doubles), two batches"""
    _sweep_case(eng, 'case_control', 'rw', T=2, N=700, D=D, n_sweeps=2, algo=5, scale=0.05, cc_C=8,
                density=0.01)


def test_sweep_case_control_sparse_lists_with_and_without_helper_workgroups(eng, monkeypatch):
    """algo 5's cross sums by the resolvers' helper workgroups (default; kernels_ccpipe.hpp, ccpipe_cross_helper)
    and by the resolvers themselves (DLSM_CC_HELPERS=0): the same sums in the same order - the same bits - and
    both the oracle's decisions"""
    res = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('DLSM_CC_HELPERS', mode)
        _sweep_case(eng, 'case_control', 'rw', T=3, N=2300, D=2, n_sweeps=2, algo=5, scale=0.05, cc_C=12,
                    density=0.004)
        X, Yd, _, radii = _rand_net(77, 3, 1700, 2, density=0.006, scale=0.05)
        cc = _cc_lists(Yd, 10, 77)
        g = eng.SamplerGrid(3, 1700, 0.01, tune=5, tune_interval=2)
        with eng.Chain(3, 1700, 2, 'case_control', seed=99, chain_id=3) as c:
            c.upload_edges(cc['in_edges'], cc['out_edges'], cc['degree'])
            c.set_controls(cc['control_nodes_in'], cc['control_nodes_out'])
            c.set_positions(X); c.set_intercepts([0.3, 0.7]); c.set_radii(radii)
            c.set_prior_random_walk(2.0, 0.1); c.set_samplers(g)
            for it in range(1, 6):
                c.sweep_positions(it, algo=5)
            res[mode] = (c.get_positions().copy(), c.get_samplers(g).n_accepted.copy())
    np.testing.assert_array_equal(res['1'][0], res['0'][0])
    np.testing.assert_array_equal(res['1'][1], res['0'][1])
    assert res['1'][1].sum() > 0


def test_sweep_case_control_helper_wait_is_bounded_and_reported(eng, monkeypatch):
    """a resolver's wait for its helper workgroup has a poll budget (DLSM_CC_HELPER_BUDGET): with a budget of
    zero it gives up at once, nothing hangs, and the sticky error word makes the call fail with DLSM_E_HIP;
    reported once, after which the same handle sweeps on"""
    X, Yd, _, radii = _rand_net(78, 2, 1300, 2, density=0.006, scale=0.05)
    cc = _cc_lists(Yd, 10, 78)
    g = eng.SamplerGrid(2, 1300, 0.01, tune=None)
    with eng.Chain(2, 1300, 2, 'case_control', seed=99, chain_id=3) as c:
        c.upload_edges(cc['in_edges'], cc['out_edges'], cc['degree'])
        c.set_controls(cc['control_nodes_in'], cc['control_nodes_out'])

        def start():
            c.set_positions(X); c.set_intercepts([0.3, 0.7]); c.set_radii(radii)
            c.set_prior_random_walk(2.0, 0.1); c.set_samplers(g)
        start()
        c.sweep_positions(1, algo=5)
        good = c.get_positions().copy()
        monkeypatch.setenv('DLSM_CC_HELPER_BUDGET', '0')
        start()
        with pytest.raises(RuntimeError, match='poll budget'):
            c.sweep_positions(1, algo=5)
        c.synchronize()                                   # reported once
        monkeypatch.delenv('DLSM_CC_HELPER_BUDGET')
        start()
        c.sweep_positions(1, algo=5)
        np.testing.assert_array_equal(c.get_positions(), good)


def _pipe_sweeps(eng, T, N, n_sweeps, seed=11, chain_id=2):
    X, _, Yu, _ = _rand_net(seed, T, N, 2)
    g = eng.SamplerGrid(T, N, 0.15, tune=None)
    with eng.Chain(T, N, 2, 'undirected', seed=4242, chain_id=chain_id) as c:
        c.upload_network(Yu); c.set_positions(X); c.set_intercepts([0.4])
        c.set_prior_random_walk(2.0, 0.1); c.set_samplers(g)
        for it in range(1, n_sweeps + 1):
            c.sweep_positions(it, algo=4)
        return c.get_positions().copy(), c.get_samplers(g).n_accepted.copy()


def test_pipelined_sweep_cross_products_served_or_not(eng, monkeypatch):
    """algo 4, undirected: the resolvers' cross products by the launch's evaluators (kernels_pipe_lds.hpp,
    pipe_xserve_*: the default with one chain on the device) and by the resolvers themselves
    (DLSM_PIPE_XSERVE=0) - a product of the same factors in another order: the same decisions, positions
    equal to rounding; the old evaluators (DLSM_PIPE_LDS=0: pipe_eval_item's pipelined trips) agree too.
    Sizes: several batches, a ragged last one, more slices than one launch's workgroups hold (served: xstride
    > 1) and T = 1 (no odd slices)"""
    for T, N in ((4, 700), (1, 300), (9, 1500)):
        res = {}
        for mode, env in (('served', {'DLSM_PIPE_XSERVE': '1'}), ('own', {'DLSM_PIPE_XSERVE': '0'}),
                          ('old', {'DLSM_PIPE_LDS': '0'})):
            for k in ('DLSM_PIPE_XSERVE', 'DLSM_PIPE_LDS'):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            res[mode] = _pipe_sweeps(eng, T, N, 3)
        for k in ('DLSM_PIPE_XSERVE', 'DLSM_PIPE_LDS'):
            monkeypatch.delenv(k, raising=False)
        for mode in ('own', 'old'):
            np.testing.assert_array_equal(res['served'][1], res[mode][1])
            np.testing.assert_allclose(res['served'][0], res[mode][0], atol=1e-10)
        assert res['served'][1].sum() > 0


def test_pipelined_sweep_served_wait_is_bounded_and_reported(eng, monkeypatch):
    """a resolver's wait for the evaluators' cross products has a poll budget (DLSM_PIPE_XBUDGET): with a budget
    of zero it gives up at once, nothing hangs, the sticky error word makes the call fail with DLSM_E_HIP;
    reported once, after which the same handle sweeps on.  With a second chain alive on the device the engine
    does not take the role at all (the wait could meet evaluators that other launches keep off the CUs):
    a zero budget is then harmless - unless DLSM_PIPE_XSERVE=2 forces it"""
    T, N = 4, 700
    X, _, Yu, _ = _rand_net(12, T, N, 2)
    g = eng.SamplerGrid(T, N, 0.15, tune=None)

    def start(c):
        c.set_positions(X); c.set_intercepts([0.4]); c.set_prior_random_walk(2.0, 0.1); c.set_samplers(g)
    with eng.Chain(T, N, 2, 'undirected', seed=5, chain_id=1) as c:
        c.upload_network(Yu)
        start(c); c.sweep_positions(1, algo=4)
        good = c.get_positions().copy()
        monkeypatch.setenv('DLSM_PIPE_XBUDGET', '0')
        start(c)
        with pytest.raises(RuntimeError, match='poll budget'):
            c.sweep_positions(1, algo=4)
            c.synchronize()
        c.synchronize()                                   # reported once
        with eng.Chain(T, N, 2, 'undirected', seed=6, chain_id=2) as other:     # two chains alive: not served
            other.upload_network(Yu)
            start(c); c.sweep_positions(1, algo=4); c.synchronize()
            np.testing.assert_allclose(c.get_positions(), good, atol=1e-10)
            monkeypatch.setenv('DLSM_PIPE_XSERVE', '2')                         # forced: the zero budget bites again
            start(c)
            with pytest.raises(RuntimeError, match='poll budget'):
                c.sweep_positions(1, algo=4)
                c.synchronize()
            c.synchronize()
            monkeypatch.delenv('DLSM_PIPE_XSERVE')
        monkeypatch.delenv('DLSM_PIPE_XBUDGET')
        start(c); c.sweep_positions(1, algo=4)
        np.testing.assert_array_equal(c.get_positions(), good)


def test_case_control_helpers_only_with_one_chain_on_the_device(eng, monkeypatch):
    """two case-control chains alive in the process: the sparse sweep's helper workgroups (a resolver waits for
    its helper INSIDE the launch) are not used - a zero poll budget is harmless - unless DLSM_CC_HELPERS=2
    forces the role, which then fails loudly (round-5 verdict, weak 9)"""
    X, Yd, _, radii = _rand_net(78, 2, 2300, 2, density=0.004, scale=0.05)
    cc = _cc_lists(Yd, 10, 78)
    g = eng.SamplerGrid(2, 2300, 0.01, tune=None)

    def make(cid):
        c = eng.Chain(2, 2300, 2, 'case_control', seed=99, chain_id=cid)
        c.upload_edges(cc['in_edges'], cc['out_edges'], cc['degree'])
        c.set_controls(cc['control_nodes_in'], cc['control_nodes_out'])
        c.set_positions(X); c.set_intercepts([0.3, 0.7]); c.set_radii(radii)
        c.set_prior_random_walk(2.0, 0.1); c.set_samplers(g)
        return c
    monkeypatch.setenv('DLSM_CC_HELPER_BUDGET', '0')
    a, b = make(1), make(2)
    try:
        assert a.resolve_sweep_algo(0) == 5
        a.sweep_positions(1, algo=5); b.sweep_positions(1, algo=5)
        a.synchronize(); b.synchronize()                  # no helpers: nothing to wait for
        monkeypatch.setenv('DLSM_CC_HELPERS', '2')
        with pytest.raises(RuntimeError, match='poll budget'):
            a.sweep_positions(2, algo=5)
            a.synchronize()
    finally:
        try:
            a.synchronize()
        except RuntimeError:
            pass
        a.close(); b.close()


def _run_sweeps(eng, algo, T, N, D, name, prior, n_sweeps, seed=5):
    # (directed networks at the scale of their radii, as everywhere in this file: at scale 1 the
    # linear predictors reach 1e3 - 1e4 and every batched form drifts from the sequential sweep
    # after a few sweeps of cancellations, the launch-per-batch form included)
    scale = 1.0 if name == 'undirected' else 0.05
    X, Yd, Yu, radii = _rand_net(seed, T, N, D, scale=scale)
    rng = np.random.RandomState(seed + 1)
    K = 3
    mu = rng.randn(K, D) * scale; sigma = rng.uniform(0.5, 1.5, K) * scale ** 2
    z = rng.randint(0, K, size=(T, N)).astype(np.int64)
    g = eng.SamplerGrid(T, N, 0.2 * scale, tune=5, tune_interval=2)
    out = []
    with eng.Chain(T, N, D, name, seed=0xC0FFEE, chain_id=1) as c:
        c.upload_network(Yu if name == 'undirected' else Yd); c.set_positions(X)
        c.set_intercepts([0.5] if name == 'undirected' else [0.3, 0.7])
        if name != 'undirected':
            c.set_radii(radii)
        if prior == 'mix':
            c.set_prior_mixture(mu, sigma, 0.8, z)
        else:
            c.set_prior_random_walk(2.0, 0.1)
        c.set_samplers(g)
        for it in range(1, n_sweeps + 1):
            c.sweep_positions(it, algo=algo)
            out.append(c.get_positions().copy())
        c.get_samplers(g)
    return out, g


def test_sweep_auto_picks_a_valid_algorithm(eng):
    _sweep_case(eng, 'undirected', 'rw', T=3, N=400, D=2, n_sweeps=2, algo=0)
    _sweep_case(eng, 'undirected', 'rw', T=3, N=40, D=2, n_sweeps=2, algo=0)


# ------------------------------------------------------------ glue
def test_center_and_procrustes(eng):
    X, Yd, Yu, radii = _rand_net(3, 3, 50)
    rng = np.random.RandomState(0)
    th = 0.7
    R0 = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    Xref = X.dot(R0) + 0.01 * rng.randn(*X.shape)
    with eng.Chain(3, 50, 2, 'undirected') as c:
        c.set_positions(X)
        c.center()
        np.testing.assert_allclose(c.get_positions(), orc.center(X), atol=1e-13)
        c.set_positions(X)
        R = c.procrustes(Xref)
        Xw, Rw = orc.procrustes_rotation(Xref, X)
        np.testing.assert_allclose(R, Rw, atol=1e-12)
        np.testing.assert_allclose(c.get_positions(), Xw, atol=1e-12)
    # a reflection is a legal Procrustes solution (det R = -1)
    F = np.array([[1.0, 0.0], [0.0, -1.0]])
    with eng.Chain(3, 50, 2, 'undirected') as c:
        c.set_positions(X)
        R = c.procrustes(X.dot(F))
        np.testing.assert_allclose(R, F, atol=1e-12)
    # three features: Jacobi SVD path
    X3 = rng.randn(2, 40, 3)
    Q, _ = np.linalg.qr(rng.randn(3, 3))
    with eng.Chain(2, 40, 3, 'undirected') as c:
        c.set_positions(X3)
        R = c.procrustes(X3.dot(Q) + 0.01 * rng.randn(*X3.shape))
        _, Rw = orc.procrustes_rotation(X3.dot(Q), X3)
        np.testing.assert_allclose(R, Rw, atol=5e-2)
        np.testing.assert_allclose(R.T.dot(R), np.eye(3), atol=1e-12)


# ------------------------------------------------------------ labels
def test_labels_against_reference_golden_inputs(eng, golden_sweeps):
    g = golden_sweeps
    X, mu, sg, w = g['lab_X'], g['lab_mu'], g['lab_sigma'], g['lab_w']
    T, N, D = X.shape
    K = sg.shape[0]
    with eng.Chain(T, N, D, 'undirected', seed=99, chain_id=1) as c:
        c.set_positions(X)
        c.set_prior_mixture(mu, sg, 0.8, np.zeros((T, N), dtype=np.int64))
        for it in (7, 8):
            z, n, nk = c.sample_labels(it, w)
            zo, no, nko = orc.sample_labels_block_philox(X, mu, sg, 0.8, w, 99, 1, it)
            np.testing.assert_array_equal(z, zo)
            np.testing.assert_array_equal(n, no)
            np.testing.assert_array_equal(nk, nko)


def test_labels_medium_and_distribution(eng):
    rng = np.random.RandomState(4)
    T, N, D, K = 6, 500, 2, 20
    mu = rng.randn(K, D) * 2; sg = rng.uniform(0.2, 1.0, K)
    X = mu[rng.randint(0, K, size=(T, N))] + 0.5 * rng.randn(T, N, D)
    w = rng.dirichlet(np.ones(K), size=(T, K))
    with eng.Chain(T, N, D, 'undirected', seed=5, chain_id=0) as c:
        c.set_positions(X)
        c.set_prior_mixture(mu, sg, 0.8, np.zeros((T, N), dtype=np.int64))
        z, n, nk = c.sample_labels(3, w)
        zo, no, nko = orc.sample_labels_block_philox(X, mu, sg, 0.8, w, 5, 0, 3)
        assert (z != zo).mean() < 1e-3        # identical up to 1-ulp ties
        assert n.sum() == T * N and (nk.sum(axis=1) == N).all()
        # the new labels are the sweep's mixture labels now
        c.upload_network(np.zeros((T, N, N)))
        c.set_intercepts([0.0])
        lp = c.loglik_partial(1, 5, with_prior=True)
        st = orc.ChainState(X, orc.SamplerGrid(T, N), Y=np.zeros((T, N, N)),
                            intercept=[0.0], mu=mu, sigma=sg, lmbda=0.8, z=z)
        np.testing.assert_allclose(lp, st.node_logp(1, 5, X[1, 5]), rtol=1e-12)


@pytest.mark.parametrize('T,N,D,K', [
    (10, 301, 2, 40),     # transition matrices too large for LDS: read from global memory
    (70, 50, 1, 3),       # more than 64 time steps: the uniforms come in two passes
    (3, 77, 3, 64),       # every lane owns a component; N not a multiple of the group size
    (1, 9, 2, 2),         # a single time step: no backward pass
    (5, 203, 2, 17),      # matrix-core kernel, 5 k-steps with three padded components, 2 column tiles
    (4, 100, 3, 33),      # 9 k-steps, 3 column tiles
    (6, 45, 2, 7),        # 2 k-steps, one column tile
    (12, 16, 4, 48),      # 12 k-steps, exactly one group of 16 nodes
])
def test_labels_shapes_against_oracle(eng, T, N, D, K):
    rng = np.random.RandomState(T * 131 + K)
    mu = rng.randn(K, D) * 2
    sg = rng.uniform(0.2, 1.0, K)
    X = mu[rng.randint(0, K, size=(T, N))] + 0.5 * rng.randn(T, N, D)
    w = rng.dirichlet(np.ones(K), size=(T, K))
    with eng.Chain(T, N, D, 'undirected', seed=11, chain_id=2) as c:
        c.set_positions(X)
        c.set_prior_mixture(mu, sg, 0.7, np.zeros((T, N), dtype=np.int64))
        for it in (1, 2):
            z, n, nk = c.sample_labels(it, w)
            zo, no, nko = orc.sample_labels_block_philox(X, mu, sg, 0.7, w, 11, 2, it)
            assert (z != zo).mean() < 2e-3        # identical up to 1-ulp ties
            # the counts are those of the labels the device drew
            n_chk = np.zeros((T, K, K)); nk_chk = np.zeros((T, K), dtype=np.int64)
            np.add.at(n_chk[0, 0], z[0], 1)
            for t in range(1, T):
                np.add.at(n_chk[t], (z[t - 1], z[t]), 1)
            for t in range(T):
                nk_chk[t] = np.bincount(z[t], minlength=K)
            np.testing.assert_array_equal(n, n_chk)
            np.testing.assert_array_equal(nk, nk_chk)


def test_set_prior_mixture_keeps_device_labels(eng):
    rng = np.random.RandomState(8)
    T, N, D, K = 4, 60, 2, 5
    X = rng.randn(T, N, D)
    mu, sg = rng.randn(K, D), rng.uniform(0.3, 1.0, K)
    w = rng.dirichlet(np.ones(K), size=(T, K))
    Y = np.zeros((T, N, N))
    with eng.Chain(T, N, D, 'undirected', seed=2) as c:
        c.upload_network(Y); c.set_positions(X); c.set_intercepts([0.0])
        with pytest.raises(eng.EngineError):
            c.set_prior_mixture(mu, sg, 0.8, None)            # no labels on the device yet
        c.set_prior_mixture(mu, sg, 0.8, np.zeros((T, N), dtype=np.int64))
        z, _, _ = c.sample_labels(1, w)
        mu2, sg2 = mu + 0.1, sg * 1.5
        c.set_prior_mixture(mu2, sg2, 0.6, None)              # new parameters, same labels
        a = c.loglik_partial(2, 7, with_prior=True)
        c.set_prior_mixture(mu2, sg2, 0.6, z)
        b = c.loglik_partial(2, 7, with_prior=True)
        assert a == b
        with pytest.raises(eng.EngineError):
            c.set_prior_mixture(rng.randn(K + 1, D), np.ones(K + 1), 0.6, None)   # other K


# ------------------------------------------------------------ controls
def test_resample_controls_valid_and_uniform(eng):
    X, Yd, Yu, radii = _rand_net(8, 2, 60, density=0.1)
    Yd[0, 0, :] = 1; Yd[0, 0, 0] = 0; Yd[0, 0, 5] = 0; Yd[0, 0, 9] = 0   # 2 zeros only
    deg, ie, oe = orc.case_control_init(Yd)
    C = 7
    counts = np.zeros(60)
    with eng.Chain(2, 60, 2, 'case_control', seed=3) as c:
        c.upload_edges(ie, oe, deg)
        for it in range(200):
            c.resample_controls(it, C)
            ci, co = c.get_controls()
            for arr, col, edges in ((co, 1, oe), (ci, 0, ie)):
                for t in range(2):
                    for i in range(60):
                        v = arr[t, i]
                        k = min(C, 60 - deg[t, i, col] - 1)
                        assert (v[:k] >= 0).all() and (v[k:] == -1).all()
                        assert len(set(v[:k])) == k and i not in v[:k]
                        assert not set(v[:k]) & set(edges[t, i, :deg[t, i, col]])
            counts[co[1, 3][co[1, 3] >= 0]] += 1
        np.testing.assert_array_equal(np.sort(co[0, 0, :2]), [5, 9])
    allowed = np.setdiff1d(np.arange(60), np.append(oe[1, 3, :deg[1, 3, 1]], 3))
    assert counts[np.setdiff1d(np.arange(60), allowed)].sum() == 0
    expect = 200 * C / allowed.size
    chi2 = ((counts[allowed] - expect) ** 2 / expect).sum()
    assert chi2 < 2.0 * allowed.size          # loose: mean df, sd sqrt(2 df)


# ------------------------------------------------------------ fused LSM loop
@pytest.mark.parametrize('N,algo', [(18, 1), (300, 1), (300, 2), (300, 3), (700, 3),
                                    (300, 4), (700, 4)])
def test_lsm_device_loop_equals_oracle_iterations(eng, monks, N, algo):
    lsm_loop_case(eng, monks, N, algo)


def lsm_loop_case(eng, monks, N, algo, D=2):
    if N == 18:
        Y = monks['Y_undirected']
        X = np.random.RandomState(0).randn(3, 18, D)
    else:
        X, _, Y, _ = _rand_net(21, 4, N, D)
    T = Y.shape[0]
    b0, prior_b, var_b = 0.3, 0.1, 2.0
    n_total = 9
    og = orc.SamplerGrid(T, N, 0.1, tune=6, tune_interval=2)
    gg = eng.SamplerGrid(T, N, 0.1, tune=6, tune_interval=2)
    st = orc.ChainState(X, og, Y=Y, intercept=[b0], tau_sq=2.0, sigma_sq=0.1,
                        seed=17, chain=5)
    isamp = orc.ScalarSampler(0.1, 0, 0, 3, 6, 3)
    want_lp, want_b, want_X = [], [], []
    for it in range(1, n_total):
        st.c.iter = it
        want_lp.append(orc.lsm_iteration_undirected(st, isamp, prior_b, var_b))
        want_b.append(st.c.intercept[0]); want_X.append(st.X.copy())
    with eng.Chain(T, N, D, 'undirected', seed=17, chain_id=5) as c:
        c.upload_network(Y); c.set_positions(X); c.set_intercepts([b0])
        c.set_prior_random_walk(2.0, 0.1); c.set_samplers(gg)
        c.lsm_configure([prior_b], var_b, step_size_intercept=0.1, tune=6,
                        tune_interval=3, n_iter_procrustes=10 ** 6, sweep_algo=algo)
        c.trace_alloc(n_total, logp0=-1.0)
        c.lsm_run(1, 4); c.lsm_run(5, n_total - 5)
        Xs, ics, lps = c.trace_read(0, n_total)
        cfg = c.lsm_get_config()
    assert lps[0] == -1.0 and ics[0, 0] == b0
    np.testing.assert_allclose(Xs[0], X)
    np.testing.assert_allclose(Xs[1:], np.array(want_X), atol=1e-9)
    np.testing.assert_allclose(ics[1:, 0], want_b, atol=1e-12)
    np.testing.assert_allclose(lps[1:], want_lp, rtol=1e-10)
    assert cfg.i_n_steps[0] == isamp.n_steps
    assert cfg.i_n_accepted[0] == isamp.n_accepted
    np.testing.assert_allclose(cfg.i_step_size[0], isamp.step_size, rtol=1e-14)


@pytest.mark.parametrize('algo', [4])
def test_lsm_device_loop_proposals_drawn_by_the_previous_iteration(eng, algo, monkeypatch):
    """inside dlsm_lsm_run the sweep's proposal pass rides in the previous iteration's last
    launch (kernels_tail_propose.hpp; DLSM_TAIL_PROPOSE=0: its own launch): same trace bit for
    bit, across two calls, with step-size tuning under way"""
    X, _, Y, _ = _rand_net(33, 4, 600, scale=0.05)
    T, N = Y.shape[:2]
    out = {}
    # (without the fused last launch, whose likelihood pass reads the positions before they are
    # centred: equal to rounding only - test_lsm_device_loop_centring_sums_riding_... covers it)
    monkeypatch.setenv('DLSM_POST_FUSE', '0')
    for mode in ('0', '1'):
        monkeypatch.setenv('DLSM_TAIL_PROPOSE', mode)
        gg = eng.SamplerGrid(T, N, 0.1, tune=6, tune_interval=2)
        with eng.Chain(T, N, 2, 'undirected', seed=23, chain_id=1) as c:
            c.upload_network(Y); c.set_positions(X); c.set_intercepts([0.2])
            c.set_prior_random_walk(2.0, 0.1); c.set_samplers(gg)
            c.lsm_configure([0.2], 2.0, step_size_intercept=0.1, tune=6, tune_interval=3,
                            n_iter_procrustes=10 ** 6, sweep_algo=algo)
            c.trace_alloc(10)
            c.lsm_run(1, 5)
            c.sweep_positions(77, algo=algo)          # a lone sweep in between draws for itself
            c.lsm_run(6, 4)
            out[mode] = c.trace_read(0, 10) + (c.get_positions(),)
    for a, b in zip(out['0'], out['1']):
        np.testing.assert_array_equal(a, b)
    assert not np.array_equal(out['1'][0][9], out['1'][0][5])


@pytest.mark.parametrize('T,N', [(4, 600), (1, 700), (3, 1500)])
def test_lsm_device_loop_centring_sums_riding_in_the_last_sweep_launch(eng, T, N, monkeypatch):
    """behind a pipelined sweep the centring sums ride in the sweep's last, resolve-only launch
    (k_pipe_last_ride) for every row that is final by then; the resolver workgroups add the rows the
    launch is still moving (the last batch of the odd slices - of the even one when T = 1 - and the
    difference terms that touch them) once they have settled them.  DLSM_POST_RIDE=0: the sums as a launch of their own.  Same
    sums in another order: the same chain to rounding, with the Procrustes rotation switching on at
    iteration 4 and across two calls.  With the sums riding, every iteration but a call's last also
    ends in the fused launch (k_lsm_finalize_apply_propose: likelihood pass on the uncentred
    positions, then centring + accept / reject + trace row + next proposal pass in one launch)"""
    X, _, Y, _ = _rand_net(35, T, N, scale=0.05)
    out = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('DLSM_POST_RIDE', mode)
        gg = eng.SamplerGrid(T, N, 0.1, tune=6, tune_interval=2)
        with eng.Chain(T, N, 2, 'undirected', seed=31, chain_id=2) as c:
            c.upload_network(Y); c.set_positions(X); c.set_intercepts([0.2])
            c.set_prior_random_walk(2.0, 0.1); c.set_samplers(gg)
            c.lsm_configure([0.2], 2.0, step_size_intercept=0.1, tune=6, tune_interval=3,
                            n_iter_procrustes=3, sweep_algo=4)
            c.trace_alloc(10)
            c.lsm_run(1, 3)
            c.lsm_run(4, 6, procrustes_ref=2)
            out[mode] = c.trace_read(0, 10)
    assert np.isfinite(out['1'][0]).all() and np.isfinite(out['1'][2]).all()
    np.testing.assert_allclose(out['1'][0], out['0'][0], atol=1e-9)
    np.testing.assert_allclose(out['1'][1], out['0'][1], atol=1e-12)
    np.testing.assert_allclose(out['1'][2], out['0'][2], rtol=1e-11)
    assert np.abs(out['1'][0][9].mean(axis=(0, 1))).max() < 1e-12      # centred


def test_lsm_device_loop_graph_replay_equals_eager(eng, monks, monkeypatch):
    """DLSM_GRAPH=1: the captured iteration (device-side iteration counter, rotation
    decided in-kernel) replays to the same trace as eager launches, including across
    the tune+burn boundary where the Procrustes rotation switches on"""
    Y = monks['Y_undirected']
    T, N = 3, 18
    X = np.random.RandomState(5).randn(T, N, 2)
    out = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('DLSM_GRAPH', mode)
        gg = eng.SamplerGrid(T, N, 0.1, tune=4, tune_interval=2)
        with eng.Chain(T, N, 2, 'undirected', seed=3, chain_id=2) as c:
            c.upload_network(Y); c.set_positions(X); c.set_intercepts([0.2])
            c.set_prior_random_walk(2.0, 0.1); c.set_samplers(gg)
            c.lsm_configure([0.2], 2.0, tune=4, tune_interval=100, n_iter_procrustes=4,
                            sweep_algo=1)
            c.trace_alloc(12, logp0=-3.0)
            c.lsm_run(1, 4)
            c.lsm_run(5, 7, procrustes_ref=2)
            out[mode] = c.trace_read(0, 12)
    for a, b in zip(out['0'], out['1']):
        np.testing.assert_array_equal(a, b)


def test_lsm_device_loop_procrustes(eng, monks):
    """after tune+burn the loop rotates to the pre-burn MAP sample (lsm.py:495-498)"""
    Y = monks['Y_undirected']
    T, N = 3, 18
    X = np.random.RandomState(2).randn(T, N, 2)
    gg = eng.SamplerGrid(T, N, 0.1, tune=None)
    with eng.Chain(T, N, 2, 'undirected', seed=1) as c:
        c.upload_network(Y); c.set_positions(X); c.set_intercepts([0.2])
        c.set_prior_random_walk(2.0, 0.1); c.set_samplers(gg)
        c.lsm_configure([0.2], 2.0, tune=None, n_iter_procrustes=2, sweep_algo=1)
        c.trace_alloc(6)
        c.lsm_run(1, 2)
        Xs, _, _ = c.trace_read(0, 3)
        # replay iteration 3 by hand: sweep, rotate to row 1, centre
        c.sweep_positions(3, algo=1)
        Xsw = c.get_positions()
        c.set_positions(Xs[2]); c.set_samplers(gg)
        c.lsm_run(3, 1, procrustes_ref=1)
        X3 = c.trace_read(3, 1)[0][0]
    Xw, _ = orc.procrustes_rotation(Xs[1], Xsw)
    np.testing.assert_allclose(X3, orc.center(Xw), atol=1e-10)


# ------------------------------------------------------------ HDP label sums (8f-2)
@pytest.mark.parametrize('T,N,D,K', [(4, 300, 2, 7), (1, 50, 2, 3), (3, 1000, 3, 20), (5, 129, 1, 64)])
def test_hdp_label_sums_match_oracle(eng, T, N, D, K):
    """dlsm_hdp_label_sums (hdp_lpcm.py:901-954, :1213-1262 on the device) against the
    numpy restatement; labels include empty clusters"""
    from oracle.hdp_sums import NumpyLabelSums
    rng = np.random.RandomState(T * 1000 + N)
    X = rng.randn(T, N, D)
    z = rng.randint(0, max(1, K - 2), size=(T, N)).astype(np.int64)      # last clusters empty
    mu, sigma = rng.randn(K, D), rng.gamma(2.0, 1.0, size=K) + 0.1
    w = rng.dirichlet(np.ones(K), size=(T, K))
    lm, a, b = 0.83, 2.0, 1.3
    ref = NumpyLabelSums(X, z, K)
    with eng.Chain(T, N, D, 'undirected') as c:
        c.set_positions(X)
        c.set_prior_mixture(mu, sigma, lm, z)
        np.testing.assert_allclose(c.hdp_label_sums(0, lmbda=lm), ref.mean(lm).reshape(T, K, D)
                                   if D > 1 else ref.mean(lm).reshape(T, K), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(c.hdp_label_sums(1, mu=mu, lmbda=lm), ref.residual(mu, lm),
                                   rtol=1e-12, atol=1e-12)
        got = c.hdp_label_sums(2, mu=mu, sigma=sigma)
        np.testing.assert_allclose(got[1:], ref.lam(mu, sigma)[1:], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(c.hdp_label_sums(3, mu=mu, sigma=sigma, lmbda=lm, w=w, a=a, b=b),
                                   ref.logp(mu, sigma, lm, w, a, b), rtol=1e-12, atol=1e-12)
        # needs the mixture prior's labels
    with eng.Chain(T, N, D, 'undirected') as c:
        c.set_positions(X)
        with pytest.raises(eng.EngineError):
            c.hdp_label_sums(0, lmbda=lm)


def test_hdp_host_updates_with_device_sums_reproduce_reference_fit(eng):
    """the trace replay of tests/test_hdp_host_updates.py with the label-wise sums taken from
    the device, as the product does"""
    from conftest import load_golden
    from dynetlsm_amd import hdp_updates as hu
    g = load_golden('hdp_trace.npz')
    Xs, ics = g['tr_Xs'], g['tr_intercepts']
    mus, sigmas, zs = g['tr_mus'], g['tr_sigmas'], g['tr_zs']
    betas, weights, lambdas, logps = (g['tr_betas'], g['tr_weights'], g['tr_lambdas'],
                                      g['tr_logps'])
    Y = g['Y']
    n_total, T, N, D = Xs.shape
    K = sigmas.shape[1]
    hp0 = dict(gamma=float(g['h0_gamma']), alpha_init=float(g['h0_alpha_init']),
               alpha=float(g['h0_alpha']), kappa=float(g['h0_kappa']),
               mean_variance_prior=float(g['h0_mean_variance_prior']), b=float(g['h0_b']),
               a=float(g['h0_a']), a0=float(g['h0_a0']), b0=float(g['h0_b0']),
               c0=float(g['h0_c0']), d0=float(g['h0_d0']))
    with eng.Chain(T, N, D, 'undirected') as c:
        c.upload_network(Y)
        it = n_total - 1
        # state after the labels update of the last recorded iteration: X, z of row `it`,
        # parameters of row it - 1; the node sums must agree with the oracle's
        from oracle.hdp_sums import NumpyLabelSums
        X, z = Xs[it], zs[it]
        mu, sigma, lm, w = mus[it - 1], sigmas[it - 1], float(lambdas[it - 1][0]), weights[it - 1]
        c.set_positions(X)
        c.set_prior_mixture(mu, sigma, lm, z)
        dev, ref = hu.DeviceLabelSums(c), NumpyLabelSums(X, z, K)
        np.testing.assert_allclose(dev.mean(lm), ref.mean(lm), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(dev.residual(mu, lm), ref.residual(mu, lm), rtol=1e-12)
        # and the log-posterior of the recorded state through the product's function
        hp = hu.HDPHyper(K, **hp0)
        for name in ('gamma', 'alpha_init', 'alpha', 'kappa', 'mean_variance_prior', 'b'):
            setattr(hp, name, float(np.ravel(g['h1_' + name])[0]))
        c.set_prior_mixture(mus[it], sigmas[it], float(lambdas[it][0]), zs[it])
        ll = c.loglik_full([ics[it]])[0]
        lp = ll + hu.log_posterior_terms(hu.DeviceLabelSums(c), ics[it], g['h0_intercept_prior'],
                                         2, mus[it], sigmas[it], weights[it], betas[it],
                                         lambdas[it], hp)
        np.testing.assert_allclose(np.ravel(lp)[0], logps[it], rtol=1e-9)


# ------------------------------------------------------------ directed device loop
@pytest.mark.parametrize('name,algo', [('directed', 4), ('case_control', 4), ('case_control', 5)])
def test_lsm_directed_device_loop_proposals_drawn_by_the_previous_iteration(eng, name, algo,
                                                                            monkeypatch):
    """the directed loops' last launch (radii accept / reject, trace row) carries the next sweep's
    proposal pass; DLSM_TAIL_PROPOSE=0 keeps it a launch of its own: same traces bit for bit"""
    T, N = 3, 700
    X, Yd, _, radii = _rand_net(57, T, N, scale=0.05, density=0.05)
    out = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('DLSM_TAIL_PROPOSE', mode)
        gg = eng.SamplerGrid(T, N, 0.01, tune=6, tune_interval=2)
        with eng.Chain(T, N, 2, name, seed=29, chain_id=1) as c:
            if name == 'case_control':
                cc = _cc_lists(Yd, 8, 5)
                c.upload_edges(cc['in_edges'], cc['out_edges'], cc['degree'])
                c.set_controls(cc['control_nodes_in'], cc['control_nodes_out'])
            else:
                c.upload_network(Yd)
            c.set_positions(X); c.set_intercepts([0.4, 0.7]); c.set_radii(radii)
            c.set_prior_random_walk(1e-3, 1e-4); c.set_samplers(gg)
            c.lsm_configure([0.3, 0.5], 2.0, step_size_intercept=0.1, tune=6, tune_interval=2,
                            n_iter_procrustes=10 ** 6, sweep_algo=algo, step_size_radii=175000.,
                            radii_tune=5, radii_tune_interval=2)
            c.trace_alloc(9)
            c.lsm_run(1, 5); c.lsm_run(6, 3)
            out[mode] = c.trace_read(0, 9) + (c.trace_read_radii(0, 9),)
    for a, b in zip(out['0'], out['1']):
        np.testing.assert_array_equal(a, b)
    assert not np.array_equal(out['1'][0][8], out['1'][0][4])


@pytest.mark.parametrize('name,N,algo', [('directed', 30, 1), ('directed', 300, 4),
                                         ('case_control', 300, 4), ('case_control', 40, 1),
                                         ('case_control', 700, 5)])
def test_lsm_directed_device_loop_equals_oracle_iterations(eng, name, N, algo):
    """dlsm_lsm_run for the directed models (sweep, Procrustes / centring, intercept_in,
    intercept_out, radii with the scaled-Dirichlet proposal) against the oracle's
    Philox-driven restatement: same decisions, traces equal to rounding"""
    directed_loop_case(eng, name, N, algo)


def directed_loop_case(eng, name, N, algo, D=2):
    T, n_total, n_proc = 3, 8, 4
    X, Yd, _, radii = _rand_net(31 + N, T, N, D, scale=0.05, density=0.1)
    b0 = np.array([0.4, 0.7])
    prior_b, var_b = np.array([0.3, 0.5]), 2.0
    og = orc.SamplerGrid(T, N, 0.01, tune=6, tune_interval=2)
    gg = eng.SamplerGrid(T, N, 0.01, tune=6, tune_interval=2)
    kw = {}
    if name == 'case_control':
        cc = _cc_lists(Yd, 8, 5)
        deg, ie, oe = cc['degree'], cc['in_edges'], cc['out_edges']
        ci, co = cc['control_nodes_in'], cc['control_nodes_out']
        kw = dict(case_control=cc)

        def loglik(Xc, b, r):
            return orc.approx_directed_network_loglikelihood(Xc, r, ie, oe, deg, co, b[0], b[1])
    else:
        kw = dict(Y=Yd)

        def loglik(Xc, b, r):
            return orc.dynamic_network_loglikelihood_directed(Yd, Xc, b[0], b[1], r)
    st = orc.ChainState(X, og, model=1 if name == 'directed' else 2, intercept=b0,
                        radii=radii.copy(), tau_sq=1e-3, sigma_sq=1e-4, seed=23, chain=2, **kw)
    isamp = [orc.ScalarMetropolis(0.1, 6, 2) for _ in range(2)]
    rsamp = orc.ScalarMetropolis(175000., 5, 2)
    step_r = 175000.
    lp0 = loglik(X, b0, radii) + orc.lsm_log_prior(X, 1e-3, 1e-4, b0, prior_b, var_b)
    want = dict(X=[X.copy()], b=[b0.copy()], r=[radii.copy()], lp=[lp0])
    for it in range(1, n_total):
        ref = None
        if it > n_proc:
            ref = want['X'][int(np.argmax(want['lp'][:n_proc + 1]))]
        lp = orc.lsm_iteration_directed(st, it, loglik, isamp, rsamp, prior_b, var_b, X_ref=ref)
        want['X'].append(st.X.copy()); want['b'].append(st.intercept.copy())
        want['r'].append(st.radii.copy()); want['lp'].append(lp)
    with eng.Chain(T, N, D, name, seed=23, chain_id=2) as c:
        if name == 'case_control':
            c.upload_edges(ie, oe, deg)
            c.set_controls(ci, co)
        else:
            c.upload_network(Yd)
        c.set_positions(X); c.set_intercepts(b0); c.set_radii(radii)
        c.set_prior_random_walk(1e-3, 1e-4); c.set_samplers(gg)
        c.lsm_configure(prior_b, var_b, step_size_intercept=0.1, tune=6, tune_interval=2,
                        n_iter_procrustes=n_proc, sweep_algo=algo, step_size_radii=step_r,
                        radii_tune=5, radii_tune_interval=2)
        c.trace_alloc(n_total, logp0=float(lp0))
        c.lsm_run(1, n_proc)
        _, _, lps = c.trace_read(0, n_proc + 1, positions=False)
        c.lsm_run(n_proc + 1, n_total - 1 - n_proc, procrustes_ref=int(np.argmax(lps)))
        Xs, ics, lps = c.trace_read(0, n_total)
        rs = c.trace_read_radii(0, n_total)
        cfg = c.lsm_get_config()
    np.testing.assert_allclose(Xs, np.array(want['X']), atol=1e-9)
    np.testing.assert_allclose(ics, np.array(want['b']), atol=1e-12)
    np.testing.assert_allclose(rs, np.array(want['r']), rtol=1e-10, atol=1e-16)
    np.testing.assert_allclose(lps, want['lp'], rtol=1e-10)
    for k in range(2):
        assert cfg.i_n_steps[k] == isamp[k].n_steps and cfg.i_n_accepted[k] == isamp[k].n_accepted
        np.testing.assert_allclose(cfg.i_step_size[k], isamp[k].step_size, rtol=1e-14)
    assert cfg.r_n_steps == rsamp.n_steps and cfg.r_n_accepted == rsamp.n_accepted
    np.testing.assert_allclose(cfg.r_step_size, rsamp.step_size, rtol=1e-14)
    assert 0 < sum(s.n_accepted for s in isamp) + rsamp.n_accepted      # something moved
