"""The device-resident HDP-LPCM loop (dlsm_hdp_run) against the CPU oracle:
oracle/hdp_loop_oracle.py runs the reference's own update code (pinned to the reference's
``_fit`` trace with MT19937 by tests/test_hdp_loop_oracle.py) with the engine's Philox draws;
the device must reproduce it iteration by iteration - discrete quantities exactly, float
quantities to rounding.  Through the C-ABI."""
import numpy as np
import pytest
import torch      # noqa: F401  (one HIP runtime per process: torch's first)

from oracle import oracle as orc
from oracle import hdp_loop_oracle as hlo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    import dynetlsm_amd
    return dynetlsm_amd


def _case(T, N, K, seed, n_true=3, density_b=1.0, D=2):
    rng = np.random.RandomState(seed)
    n_true = min(n_true, K)
    cen = 2.5 * rng.randn(n_true, D)
    z = np.zeros((T, N), dtype=np.int64)
    z[0] = rng.randint(0, n_true, N)
    for t in range(1, T):
        mv = rng.rand(N) < 0.15
        z[t] = np.where(mv, rng.randint(0, n_true, N), z[t - 1])
    X = np.zeros((T, N, D))
    X[0] = cen[z[0]] + 0.5 * rng.randn(N, D)
    for t in range(1, T):
        X[t] = 0.2 * X[t - 1] + 0.8 * cen[z[t]] + 0.5 * rng.randn(N, D)
    X -= X.mean(axis=(0, 1))
    Y = np.zeros((T, N, N))
    for t in range(T):
        d = np.sqrt(((X[t][:, None] - X[t][None]) ** 2).sum(-1))
        A = (rng.rand(N, N) < 1 / (1 + np.exp(-(density_b - d)))).astype(float)
        A = np.triu(A, 1)
        Y[t] = A + A.T
    mu = np.vstack([cen - X.mean(axis=(0, 1)), 2.0 * rng.randn(K - n_true, D)])[:K]
    sigma = rng.uniform(0.2, 0.8, K)
    z0 = z.copy()
    flip = rng.rand(T, N) < 0.2
    z0[flip] = rng.randint(0, K, flip.sum())
    beta = rng.dirichlet(np.ones(K))
    w = rng.dirichlet(np.ones(K) * 0.7, size=(T, K))
    return Y, X + 0.1 * rng.randn(T, N, D), mu, sigma, z0, beta, w


def _hyper(a0=True, c0=True):
    kw = dict(gamma=1.3, alpha_init=0.9, alpha=1.1, kappa=3.5, mean_variance_prior=2.4, b=1.7,
              a=2.0, lambda_prior=0.9, lambda_variance_prior=0.01, gamma_prior_shape=1.0,
              gamma_prior_rate=0.1, alpha_init_shape=1.0, alpha_init_rate=1.0,
              alpha_kappa_shape=5, alpha_kappa_rate=0.1)
    if a0:
        kw.update(a0=36.0, b0=9.3)
    if c0:
        kw.update(c0=16.0, d0=9.4)
    return hlo.Hyper(**kw)


def _run_both(eng, T, N, K, seed, n_it, tune=None, algo=0, a0=True, c0=True, lmbda=0.8,
              check_each=True, D=2):
    Y, X, mu, sigma, z, beta, w = _case(T, N, K, seed, D=D)
    b0, ip, var = 0.6, 0.5, 2.0
    hp = _hyper(a0, c0)
    og = orc.SamplerGrid(T, N, 0.15, tune=tune, tune_interval=2)
    isamp = orc.ScalarMetropolis(0.1, tune, 100)
    # (the oracle updates its arrays in place: hand it copies)
    oc = hlo.HdpChain(Y, X.copy(), [b0], mu.copy(), sigma.copy(), z.copy(), beta.copy(), w.copy(),
                      lmbda, hp.copy(), og, ip, var, isamp, seed=77 + seed, chain=2)
    out = []
    with eng.Chain(T, N, D, 'undirected', seed=77 + seed, chain_id=2) as c:
        c.upload_network(Y); c.set_positions(X); c.set_intercepts([b0])
        c.set_samplers(eng.SamplerGrid(T, N, 0.15, tune=tune, tune_interval=2))
        c.set_prior_mixture(mu, sigma, lmbda, z)
        c.hdp_configure(hp, beta, w, ip, var, step_size_intercept=0.1, tune=tune,
                        tune_interval=100, sweep_algo=algo)
        c.hdp_trace_alloc(n_it + 1, logp0=-1.0)
        for it in range(1, n_it + 1):
            c.hdp_run(it, 1)
            c.synchronize()
            lp = oc.iteration(it)
            aux = c.hdp_get_aux()
            tr = c.hdp_trace_read(it, 1)
            # discrete: labels, counts, tables, override variables
            np.testing.assert_array_equal(tr['zs'][0], oc.z)
            np.testing.assert_array_equal(aux['n'], oc.n.astype(np.int64))
            np.testing.assert_array_equal(aux['nk'], oc.nk)
            np.testing.assert_array_equal(aux['m'], oc.aux['m'])
            if T > 1:
                np.testing.assert_array_equal(aux['w_over'], oc.aux['w'].astype(np.int64))
            np.testing.assert_allclose(aux['m_bar'], oc.aux['m_bar'], rtol=0, atol=0)
            # continuous
            np.testing.assert_allclose(tr['Xs'][0], oc.X, atol=1e-9)
            np.testing.assert_allclose(tr['intercepts'][0, 0], oc.intercept[0], rtol=1e-11)
            np.testing.assert_allclose(tr['betas'][0], oc.beta, rtol=1e-9)
            np.testing.assert_allclose(tr['weights'][0], oc.weights, rtol=1e-8, atol=1e-300)
            np.testing.assert_allclose(tr['mus'][0], oc.mu, rtol=1e-8, atol=1e-10)
            np.testing.assert_allclose(tr['sigmas'][0], oc.sigma, rtol=1e-9)
            np.testing.assert_allclose(tr['lambdas'][0, 0], oc.lmbda[0], rtol=1e-9)
            h = oc.hp
            want = [h.gamma, h.alpha_init, h.alpha, h.kappa,
                    float(np.ravel(h.mean_variance_prior)[0]), h.b]
            np.testing.assert_allclose(tr['hypers'][0], want, rtol=1e-8)
            np.testing.assert_allclose(tr['logps'][0], lp, rtol=1e-9)
            out.append((tr, lp))
        cfg = c.hdp_get_config()
        assert cfg.i_n_steps == isamp.n_steps and cfg.i_n_accepted == isamp.n_accepted
        np.testing.assert_allclose(cfg.i_step_size, isamp.step_size, rtol=1e-13)
        g = c.get_samplers(eng.SamplerGrid(T, N, 0.15, tune=tune, tune_interval=2))
        np.testing.assert_array_equal(g.n_accepted, og.n_accepted)
        np.testing.assert_allclose(g.step_size, og.step_size, rtol=1e-13)
        tr0 = c.hdp_trace_read(0, 1)
        np.testing.assert_array_equal(tr0['zs'][0], z)
        np.testing.assert_allclose(tr0['mus'][0], mu)
        assert tr0['logps'][0] == -1.0
    return out


@pytest.mark.parametrize('T,N,K,seed', [(3, 60, 4, 0), (4, 150, 6, 1), (2, 33, 3, 2), (5, 70, 9, 3)])
def test_device_loop_matches_oracle_iteration_by_iteration(eng, T, N, K, seed):
    _run_both(eng, T, N, K, seed, n_it=5)


def test_device_loop_with_tuning_and_without_hyperpriors(eng):
    _run_both(eng, 3, 80, 5, 5, n_it=6, tune=4)
    _run_both(eng, 3, 50, 4, 6, n_it=3, a0=False, c0=False)
    _run_both(eng, 3, 50, 4, 7, n_it=3, a0=True, c0=False)


@pytest.mark.parametrize('T,N,K,D,seed', [(3, 45, 5, 1, 12), (2, 50, 33, 3, 13), (3, 40, 1, 2, 14),
                                          (2, 36, 64, 4, 15)])
def test_device_loop_other_dimensions_and_component_counts(eng, T, N, K, D, seed):
    """d = 1, 3, 4; a single component (every Dirichlet draw is 1); more components than half a
    wavefront; the 64-component limit"""
    _run_both(eng, T, N, K, seed, n_it=3, D=D)


def test_device_loop_single_time_step(eng):
    """T = 1: no transitions, no override variables (empty sums in hdp_lpcm.py:999-1020)"""
    _run_both(eng, 1, 60, 4, 8, n_it=4)


def test_device_loop_with_the_pipelined_sweep(eng):
    """N = 600: fit()'s default sweep at this size (algo 4), K = 20 as config 3"""
    _run_both(eng, 4, 600, 20, 9, n_it=3, algo=0)


def test_device_loop_proposals_drawn_by_the_previous_iteration(eng, monkeypatch):
    """several iterations per dlsm_hdp_run call: the sweep's proposal pass rides in the launch of
    the hyper-parameters (kernels_tail_propose.hpp); DLSM_TAIL_PROPOSE=0 keeps it a launch of its
    own - bit for bit the same trace"""
    T, N, K = 4, 600, 20
    Y, X, mu, sigma, z, beta, w = _case(T, N, K, 21)
    hp = _hyper()
    out = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('DLSM_TAIL_PROPOSE', mode)
        with eng.Chain(T, N, 2, 'undirected', seed=5, chain_id=3) as c:
            c.upload_network(Y); c.set_positions(X); c.set_intercepts([0.6])
            c.set_samplers(eng.SamplerGrid(T, N, 0.15, tune=4, tune_interval=2))
            c.set_prior_mixture(mu, sigma, 0.8, z)
            c.hdp_configure(hp, beta, w, 0.5, 2.0, step_size_intercept=0.1, tune=4,
                            tune_interval=100, sweep_algo=4)
            c.hdp_trace_alloc(9)
            c.hdp_run(1, 5); c.hdp_run(6, 3)
            out[mode] = c.hdp_trace_read(0, 9)
    for key in out['0']:
        np.testing.assert_array_equal(out['0'][key], out['1'][key], err_msg=key)
    assert not np.array_equal(out['1']['Xs'][8], out['1']['Xs'][4])


@pytest.mark.parametrize('T,N,K,ride', [(4, 600, 20, '1'), (3, 260, 7, '1'), (5, 700, 12, '0')])
def test_device_loop_on_two_queues_is_the_one_queue_trace(eng, monkeypatch, T, N, K, ride):
    """the intercept's likelihood pass on a queue of its own beside the label update and the
    conjugate draws, handed over through device flags (kernels_hdploop.hpp, HdpFork): forced
    (DLSM_HDP_QUEUES=2), chosen by the engine (one live chain: unset) and switched off (=1) - bit
    for bit the same trace, over several calls and with the proposal pass riding or not; with the
    next sweep's head (proposal pass + first launch) on the second queue as well, or not
    (DLSM_HDP_HEAD=0)"""
    Y, X, mu, sigma, z, beta, w = _case(T, N, K, 33)
    hp = _hyper()
    monkeypatch.setenv('DLSM_TAIL_PROPOSE', ride)
    out = {}
    for mode in ('1', '2', None, '2-nohead', '2-gate-kernel'):
        monkeypatch.delenv('DLSM_HDP_HEAD', raising=False)
        monkeypatch.delenv('DLSM_HDP_GATE', raising=False)
        if mode == '2-gate-kernel':         # the queues' waits as one-wavefront gate kernels instead of
            monkeypatch.setenv('DLSM_HDP_QUEUES', '2')      # hipStreamWaitValue32 (read when the second
            monkeypatch.setenv('DLSM_HDP_GATE', 'kernel')   # queue is created: per handle)
        elif mode is None:
            monkeypatch.delenv('DLSM_HDP_QUEUES', raising=False)
        elif mode == '2-nohead':            # the next sweep's head stays on the chain's queue
            monkeypatch.setenv('DLSM_HDP_QUEUES', '2')
            monkeypatch.setenv('DLSM_HDP_HEAD', '0')
        else:
            monkeypatch.setenv('DLSM_HDP_QUEUES', mode)
        with eng.Chain(T, N, 2, 'undirected', seed=9, chain_id=2) as c:
            c.upload_network(Y); c.set_positions(X); c.set_intercepts([0.6])
            c.set_samplers(eng.SamplerGrid(T, N, 0.15, tune=4, tune_interval=2))
            c.set_prior_mixture(mu, sigma, 0.8, z)
            c.hdp_configure(hp, beta, w, 0.5, 2.0, step_size_intercept=0.1, tune=4,
                            tune_interval=100, sweep_algo=4)
            c.hdp_trace_alloc(13)
            c.hdp_run(1, 1); c.hdp_run(2, 6); c.hdp_run(8, 5)
            out[mode] = c.hdp_trace_read(0, 13)
            aux = c.hdp_get_aux()
            out[mode]['aux_n'], out[mode]['aux_nk'] = aux['n'], aux['nk']
    for mode in ('2', None, '2-nohead', '2-gate-kernel'):
        for key in out['1']:
            np.testing.assert_array_equal(out['1'][key], out[mode][key], err_msg='%s (queues %s)' % (key, mode))
    assert not np.array_equal(out['2']['intercepts'][12], out['2']['intercepts'][1])


def test_device_loop_two_queue_waits_are_bounded_and_reported(eng, monkeypatch):
    """the waits INSIDE kernels of the two-queue loop have a poll budget (DLSM_HDP_FORK_BUDGET): with a
    budget of zero every such wait gives up at once, nothing hangs, and the sticky error word - host memory
    the device stores to, read behind the synchronisation without a copy - makes the next synchronising
    call fail with DLSM_E_HIP; reported once, after which the handle goes on with one queue"""
    T, N, K = 4, 600, 20
    Y, X, mu, sigma, z, beta, w = _case(T, N, K, 33)
    hp = _hyper()
    monkeypatch.setenv('DLSM_HDP_QUEUES', '2')

    def start(c):
        c.set_positions(X); c.set_intercepts([0.6])
        c.set_samplers(eng.SamplerGrid(T, N, 0.15, tune=None))
        c.set_prior_mixture(mu, sigma, 0.8, z)
        c.hdp_configure(hp, beta, w, 0.5, 2.0, sweep_algo=4)
        c.hdp_trace_alloc(6)
    with eng.Chain(T, N, 2, 'undirected', seed=9, chain_id=2) as c:
        c.upload_network(Y)
        start(c)
        c.hdp_run(1, 5)
        good = c.hdp_trace_read(0, 6)
        assert c.hdp_queues() == 2
        monkeypatch.setenv('DLSM_HDP_FORK_BUDGET', '0')
        start(c)
        c.hdp_run(1, 5)
        with pytest.raises(RuntimeError, match='poll budget'):
            c.synchronize()
        c.synchronize()                                   # reported once
        monkeypatch.delenv('DLSM_HDP_FORK_BUDGET')
        monkeypatch.setenv('DLSM_HDP_QUEUES', '1')
        start(c)
        c.hdp_run(1, 5)
        again = c.hdp_trace_read(0, 6)
    for key in good:
        np.testing.assert_array_equal(good[key], again[key], err_msg=key)


def test_truncated_normal_far_tails(eng):
    """the blending coefficient's draw when its conditional sits far outside [0, 1] or is very
    sharp: the device's log-space quantile against scipy's (through the oracle)"""
    # exercised through one iteration each with extreme lambda priors
    for lp, lv in [(5.0, 1e-4), (-3.0, 1e-4), (0.5, 1e-9), (0.999, 1e-6)]:
        Y, X, mu, sigma, z, beta, w = _case(3, 40, 4, 11)
        hp = _hyper()
        hp.lambda_prior, hp.lambda_variance_prior = lp, lv
        og = orc.SamplerGrid(3, 40, 0.15, tune=None)
        oc = hlo.HdpChain(Y, X.copy(), [0.6], mu.copy(), sigma.copy(), z.copy(), beta.copy(),
                          w.copy(), 0.8, hp.copy(), og, 0.5, 2.0,
                          orc.ScalarMetropolis(0.1, None, 100), seed=5, chain=0)
        with eng.Chain(3, 40, 2, 'undirected', seed=5, chain_id=0) as c:
            c.upload_network(Y); c.set_positions(X); c.set_intercepts([0.6])
            c.set_samplers(eng.SamplerGrid(3, 40, 0.15, tune=None))
            c.set_prior_mixture(mu, sigma, 0.8, z)
            c.hdp_configure(hp, beta, w, 0.5, 2.0)
            c.hdp_trace_alloc(3)
            for it in (1, 2):
                c.hdp_run(it, 1)
                want = oc.iteration(it)
                tr = c.hdp_trace_read(it, 1, positions=False)
                assert 0.0 <= tr['lambdas'][0, 0] <= 1.0
                np.testing.assert_allclose(tr['lambdas'][0, 0], oc.lmbda[0], rtol=1e-8, atol=1e-12)
                np.testing.assert_allclose(tr['logps'][0], want, rtol=1e-9)


def test_facade_device_and_host_loops_agree_in_distribution(eng):
    """the same small network through both loops of DynamicNetworkHDPLPCM: posterior means of
    the intercept, the blending coefficient and the occupied clusters agree within MC error"""
    Y = _case(3, 40, 4, 21)[0]
    res = {}
    for loop in ('device', 'host'):
        got = []
        for seed in range(4):
            m = eng.DynamicNetworkHDPLPCM(n_iter=300, tune=150, burn=150, n_components=4,
                                          random_state=seed, chain_id=seed, hdp_loop=loop).fit(Y)
            assert m.loop_kind_ == ('device-resident' if loop == 'device' else 'host-driven')
            keep = slice(300, None)
            nclu = np.array([[len(np.unique(z[t])) for t in range(3)] for z in m.zs_[keep]]).mean()
            got.append([m.intercepts_[keep, 0].mean(), m.lambdas_[keep, 0].mean(), nclu,
                        m.logps_[keep].mean()])
            assert np.isfinite(m.logps_).all() and (m.sigmas_[1:] > 0).all()
            np.testing.assert_allclose(m.weights_[-1].sum(-1)[1:], 1.0, rtol=1e-12)
            np.testing.assert_allclose(m.betas_[-1].sum(), 1.0, rtol=1e-12)
        res[loop] = np.array(got)
    for k, slack in ((0, 0.05), (1, 0.03), (2, 0.15), (3, 0.02)):
        a, b = res['device'][:, k], res['host'][:, k]
        se = np.sqrt(a.var(ddof=1) / 4 + b.var(ddof=1) / 4)
        assert abs(a.mean() - b.mean()) < 4 * se + slack * abs(b.mean()), (k, a, b)


# ---------------------------------------------------------------- directed models on the device
def _directed_case(T, N, K, seed):
    """a directed network with the case's positions, radii ~ Dirichlet, intercepts (0.8, 0.4)"""
    Y, X, mu, sigma, z, beta, w = _case(T, N, K, seed)
    rng = np.random.RandomState(100 + seed)
    radii = rng.dirichlet(np.ones(N) * 8.0)
    Xs = 0.02 * X                                       # the directed model's scale (radii ~ 1 / N)
    mu, sigma = 0.02 * mu, (0.02 ** 2) * sigma
    b = np.array([0.8, 0.4])
    Yd = np.zeros((T, N, N))
    for t in range(T):
        d = np.sqrt(((Xs[t][:, None] - Xs[t][None]) ** 2).sum(-1))
        eta = b[0] * (1 - d / radii[None, :]) + b[1] * (1 - d / radii[:, None])
        A = (rng.rand(N, N) < 1 / (1 + np.exp(-eta))).astype(float)
        np.fill_diagonal(A, 0.0)
        Yd[t] = A
    return Yd, Xs, radii, b, mu, sigma, z, beta, w


@pytest.mark.parametrize('T,N,K,seed,model,tune', [(3, 60, 4, 0, 'directed', None), (2, 90, 5, 1, 'directed', 3),
                                                   (3, 70, 4, 2, 'case_control', None),
                                                   (2, 600, 5, 3, 'directed', None)])
def test_directed_device_loop_matches_oracle_iteration_by_iteration(eng, T, N, K, seed, model, tune):
    """hdp_lpcm.py:823-1069 with is_directed on the device (dlsm_hdp_run): sweep with the directed
    (or case-control) partial likelihoods and the mixture prior, centring, intercept_in,
    intercept_out, radii (sample_coefficients.py:12-121), label update, conjugate draws and the
    log-posterior of hdp_lpcm.py:1188-1280 with its directed terms - against the oracle's
    restatement with the engine's draws, iteration by iteration (N = 600: the pipelined sweep)"""
    Y, X, radii, b, mu, sigma, z, beta, w = _directed_case(T, N, K, seed)
    ip, var = np.array([0.7, 0.5]), 2.0
    hp = _hyper()
    og = orc.SamplerGrid(T, N, 0.004, tune=tune, tune_interval=2)
    isamps = [orc.ScalarMetropolis(0.1, tune, 100) for _ in range(2)]
    rsamp = orc.ScalarMetropolis(20000., tune, 100)
    cc = None
    n_it = 4
    with eng.Chain(T, N, 2, model, seed=31 + seed, chain_id=1) as c:
        if model == 'case_control':
            from dynetlsm_amd.case_control import build_edge_lists
            dg, ie, oe = build_edge_lists(Y)
            c.upload_edges(ie, oe, dg)
            c.resample_controls(0, 12)
            ci, co = c.get_controls()
            cc = dict(in_edges=ie, out_edges=oe, degree=dg, control_nodes_in=ci, control_nodes_out=co)
        else:
            c.upload_network(Y)
        c.set_positions(X); c.set_intercepts(b); c.set_radii(radii)
        c.set_samplers(eng.SamplerGrid(T, N, 0.004, tune=tune, tune_interval=2))
        c.set_prior_mixture(mu, sigma, 0.8, z)
        c.hdp_configure(hp, beta, w, ip, var, step_size_intercept=0.1, tune=tune, tune_interval=100,
                        step_size_radii=20000., radii_tune=tune)
        c.hdp_trace_alloc(n_it + 1, logp0=-1.0)
        oc = hlo.HdpChainDirected(None if cc else Y, X.copy(), b.copy(), radii.copy(), mu.copy(), sigma.copy(),
                                  z.copy(), beta.copy(), w.copy(), 0.8, hp.copy(), og, ip, var, isamps, rsamp,
                                  seed=31 + seed, chain=1, case_control=cc)
        for it in range(1, n_it + 1):
            c.hdp_run(it, 1)
            c.synchronize()
            lp = oc.iteration(it)
            tr = c.hdp_trace_read(it, 1)
            rad = c.trace_read_radii(it, 1)[0]
            np.testing.assert_array_equal(tr['zs'][0], oc.z)
            np.testing.assert_allclose(tr['Xs'][0], oc.X, atol=1e-11)
            np.testing.assert_allclose(tr['intercepts'][0], oc.intercept, rtol=1e-11)
            np.testing.assert_allclose(rad, oc.radii, rtol=1e-9)
            np.testing.assert_allclose(tr['betas'][0], oc.beta, rtol=1e-9)
            np.testing.assert_allclose(tr['mus'][0], oc.mu, rtol=1e-8, atol=1e-12)
            np.testing.assert_allclose(tr['sigmas'][0], oc.sigma, rtol=1e-9)
            np.testing.assert_allclose(tr['lambdas'][0, 0], oc.lmbda[0], rtol=1e-9)
            np.testing.assert_allclose(tr['logps'][0], lp, rtol=1e-9)
            np.testing.assert_allclose(c.get_radii(), oc.radii, rtol=1e-9)
        cfg = c.hdp_get_config()
        assert (cfg.i_n_steps, cfg.i_n_steps_out, cfg.r_n_steps) == (n_it, n_it, n_it)
        assert cfg.i_n_accepted == isamps[0].n_accepted and cfg.i_n_accepted_out == isamps[1].n_accepted
        assert cfg.r_n_accepted == rsamp.n_accepted
        np.testing.assert_allclose([cfg.i_step_size, cfg.i_step_size_out, cfg.r_step_size],
                                   [isamps[0].step_size, isamps[1].step_size, rsamp.step_size], rtol=1e-13)
        tr0 = c.hdp_trace_read(0, 1)
        np.testing.assert_allclose(tr0['intercepts'][0], b)          # row 0 keeps BOTH intercepts


def test_directed_facade_device_loop_runs_and_lands_where_the_host_loop_does(eng):
    """DynamicNetworkHDPLPCM(is_directed=True, hdp_loop='device'): the estimator's surface (radii trace,
    selected model) and a posterior level comparable to the host-driven loop's on the same network"""
    Y, X, radii, b, mu, sigma, z, beta, w = _directed_case(3, 40, 4, 5)
    out = {}
    for loop in ('device', 'host'):
        lp = []
        for seed in range(3):
            m = eng.DynamicNetworkHDPLPCM(n_iter=120, tune=60, burn=60, is_directed=True, n_components=4,
                                          random_state=seed, chain_id=seed, hdp_loop=loop, selection_type='map')
            m.fit(Y)
            assert m.loop_kind_ == ('device-resident' if loop == 'device' else 'host-driven')
            assert m.radiis_.shape == (240, 40) and np.allclose(m.radiis_[1:].sum(axis=1), 1.0)
            assert m.intercepts_.shape == (240, 2) and np.isfinite(m.logps_).all()
            assert m.X_.shape == (3, 40, 2) and m.radii_.shape == (40,)
            lp.append(m.logps_[120:].mean())
            m.chain_.close()
        out[loop] = np.array(lp)
    se = np.sqrt(out['device'].var(ddof=1) / 3 + out['host'].var(ddof=1) / 3)
    assert abs(out['device'].mean() - out['host'].mean()) < 5 * se + 0.05 * abs(out['host'].mean()), out
