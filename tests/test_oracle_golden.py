"""Pin the CPU oracle against golden vectors captured from the reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import LIK_TAGS

from oracle import oracle as orc

RTOL = 1e-12


def _cc(g, tag, t):
    return (g[tag + '_in_edges'][t], g[tag + '_out_edges'][t],
            g[tag + '_degrees'][t], g[tag + '_ctrl_in'][t],
            g[tag + '_ctrl_out'][t])


@pytest.mark.parametrize('tag', LIK_TAGS)
@pytest.mark.parametrize('sq', [0, 1])
def test_partials_and_fulls(golden_lik, tag, sq):
    g = golden_lik
    X, Yd, Yu, radii = g[tag + '_X'], g[tag + '_Yd'], g[tag + '_Yu'], g[tag + '_radii']
    b, b_in, b_out = g[tag + '_b']
    T, N, D = X.shape
    pu = np.array([[orc.partial_loglikelihood(Yu[t], X[t], b, j, squared=sq)
                    for j in range(N)] for t in range(T)])
    np.testing.assert_allclose(pu, g['%s_partial_undirected_sq%d' % (tag, sq)],
                               rtol=RTOL)
    pd = np.array([[orc.directed_partial_loglikelihood(
        Yd[t], X[t], radii, b_in, b_out, j, squared=sq) for j in range(N)]
        for t in range(T)])
    np.testing.assert_allclose(pd, g['%s_partial_directed_sq%d' % (tag, sq)],
                               rtol=RTOL)
    # the reference's full log-lik goes through sklearn's expanded-form
    # euclidean_distances (latent_space.py:29): allow its cancellation error
    fu = orc.dynamic_network_loglikelihood_undirected(Yu, X, b, squared=sq)
    np.testing.assert_allclose(fu, g['%s_full_undirected_sq%d' % (tag, sq)],
                               rtol=1e-10)
    fd = orc.dynamic_network_loglikelihood_directed(Yd, X, b_in, b_out, radii,
                                                    squared=sq)
    np.testing.assert_allclose(fd, g['%s_full_directed_sq%d' % (tag, sq)],
                               rtol=1e-10)
    # identities of SURVEY 3.4-7
    np.testing.assert_allclose(pu.sum() / 2, fu, rtol=1e-12)
    np.testing.assert_allclose(pd.sum() / 2, fd, rtol=1e-12)


def test_known_answer_anchors(golden_lik):
    """SURVEY.md 8c anchors (seed 12345, T=3, N=7)."""
    g = golden_lik
    assert abs(g['a_full_undirected_sq0'] - (-43.5096731257319)) < 1e-9
    assert abs(g['a_partial_undirected_sq0'][1, 3] - (-2.3661039320691923)) < 1e-12
    assert abs(g['a_partial_undirected_sq1'][1, 3] - (-1.9357898081918465)) < 1e-12
    assert abs(g['a_full_directed_sq0'] - (-2804.043510394001)) < 1e-7


@pytest.mark.parametrize('tag', LIK_TAGS)
@pytest.mark.parametrize('sq', [0, 1])
def test_case_control(golden_lik, tag, sq):
    g = golden_lik
    X, radii = g[tag + '_X'], g[tag + '_radii']
    _, b_in, b_out = g[tag + '_b']
    T, N, D = X.shape
    ctrl_in, ctrl_out = g[tag + '_ctrl_in'], g[tag + '_ctrl_out']
    # the reference's second control loop tests the IN sentinel while indexing
    # the OUT list (directed_likelihoods_fast.pyx:160-167).  Per node:
    #   n_in == n_out : harmless, both modes must equal the reference
    #   n_in <  n_out : reference truncates the out-loop; only ref_compat matches
    #   n_in >  n_out : reference reads X[-1] out of bounds -> not comparable
    n_in = (ctrl_in >= 0).sum(axis=2)
    n_out = (ctrl_out >= 0).sum(axis=2)
    want = g['%s_partial_approx_sq%d' % (tag, sq)]
    for compat in (0, 1):
        pa = np.array([[orc.approx_directed_partial_loglikelihood(
            X[t], radii, *_cc(g, tag, t), b_in, b_out, j, squared=sq,
            ref_compat=compat) for j in range(N)] for t in range(T)])
        ok = (n_in == n_out) if compat == 0 else (n_in <= n_out)
        assert ok.sum() > 0
        np.testing.assert_allclose(pa[ok], want[ok], rtol=RTOL)
    fa = orc.approx_directed_network_loglikelihood(
        X, radii, g[tag + '_in_edges'], g[tag + '_out_edges'], g[tag + '_degrees'],
        ctrl_out, b_in, b_out, squared=sq)
    np.testing.assert_allclose(fa, g['%s_full_approx_sq%d' % (tag, sq)], rtol=RTOL)


@pytest.mark.parametrize('tag', LIK_TAGS)
def test_case_control_exhaustive_equals_exact(golden_lik, tag):
    g = golden_lik
    X, radii, Yd = g[tag + '_X'], g[tag + '_radii'], g[tag + '_Yd']
    _, b_in, b_out = g[tag + '_b']
    fa = orc.approx_directed_network_loglikelihood(
        X, radii, g[tag + '_in_edges'], g[tag + '_out_edges'], g[tag + '_degrees'],
        g[tag + '_ctrl_out_all'], b_in, b_out)
    np.testing.assert_allclose(fa, g[tag + '_full_approx_all'], rtol=RTOL)
    fd = orc.dynamic_network_loglikelihood_directed(Yd, X, b_in, b_out, radii)
    np.testing.assert_allclose(fa, fd, rtol=1e-11)


@pytest.mark.parametrize('tag', LIK_TAGS)
def test_case_control_init(golden_lik, tag):
    g = golden_lik
    deg, ie, oe = orc.case_control_init(g[tag + '_Yd'])
    np.testing.assert_array_equal(deg, g[tag + '_degrees'])
    np.testing.assert_array_equal(ie, g[tag + '_in_edges'])
    np.testing.assert_array_equal(oe, g[tag + '_out_edges'])


@pytest.mark.parametrize('tag', LIK_TAGS)
@pytest.mark.parametrize('nz', [0, 1])
def test_gaussian_likelihood(golden_lik, tag, nz):
    g = golden_lik
    X = g[tag + '_X']
    for i in range(X.shape[1]):
        tab = orc.compute_gaussian_likelihood(X[:, i], g[tag + '_mu'],
                                              g[tag + '_sigma'], 0.8,
                                              normalize=bool(nz))
        np.testing.assert_allclose(tab, g['%s_gauss_norm%d' % (tag, nz)][i],
                                   rtol=RTOL)


# ---------------------------------------------------------------- sweeps
def _grid(T, N):
    return orc.SamplerGrid(T, N, step_size=0.2, tune=5, tune_interval=2)


@pytest.mark.parametrize('prior', ['rw', 'mix'])
@pytest.mark.parametrize('name', ['undirected', 'directed', 'casecontrol'])
def test_sweep_mt_reproduces_reference(golden_sweeps, prior, name):
    g = golden_sweeps
    X0 = g['X0']
    T, N, D = X0.shape
    grid = _grid(T, N)
    kw = dict(tau_sq=2.0, sigma_sq=0.1)
    if prior == 'mix':
        kw = dict(mu=g['mu'], sigma=g['sigma'], lmbda=g['lmbda'], z=g['z'])
    if name == 'undirected':
        kw.update(Y=g['Yu'], intercept=[0.5], model=0)
    elif name == 'directed':
        kw.update(Y=g['Yd'], intercept=[0.3, 0.7], radii=g['radii'], model=1)
    else:
        kw.update(intercept=[0.3, 0.7], radii=g['radii'], model=2,
                  case_control=dict(in_edges=g['cc_in_edges'],
                                    out_edges=g['cc_out_edges'],
                                    degree=g['cc_degrees'],
                                    control_nodes_in=g['cc_ctrl_in'],
                                    control_nodes_out=g['cc_ctrl_out']))
    st = orc.ChainState(X0, grid, **kw)
    rng = np.random.RandomState(5)
    key = 'sweep_%s_%s' % (prior, name)
    want = g[key + '_X']
    for s in range(want.shape[0]):
        X = st.sweep_py(orc.MTDraws(rng), order='reference')
        np.testing.assert_allclose(X, want[s], rtol=0, atol=1e-10)
    np.testing.assert_allclose(grid.step_size, g[key + '_step'], rtol=1e-14)
    np.testing.assert_array_equal(grid.n_accepted, g[key + '_nacc'])
    np.testing.assert_array_equal(grid.n_steps, g[key + '_nsteps'])
    np.testing.assert_array_equal(grid.steps_until_tune, g[key + '_until'])


@pytest.mark.parametrize('prior', ['rw', 'mix'])
@pytest.mark.parametrize('name', ['undirected', 'directed', 'casecontrol'])
def test_sweep_c_philox_equals_python_philox(golden_sweeps, prior, name):
    """C restatement (engine order, Philox) == generic python sweep with the
    same draws: ties the C code to the reference-pinned python code."""
    g = golden_sweeps
    X0 = g['X0']
    T, N, D = X0.shape
    kw = dict(tau_sq=2.0, sigma_sq=0.1)
    if prior == 'mix':
        kw = dict(mu=g['mu'], sigma=g['sigma'], lmbda=g['lmbda'], z=g['z'])
    if name == 'undirected':
        kw.update(Y=g['Yu'], intercept=[0.5], model=0)
    elif name == 'directed':
        kw.update(Y=g['Yd'], intercept=[0.3, 0.7], radii=g['radii'], model=1)
    else:
        kw.update(intercept=[0.3, 0.7], radii=g['radii'], model=2,
                  case_control=dict(in_edges=g['cc_in_edges'],
                                    out_edges=g['cc_out_edges'],
                                    degree=g['cc_degrees'],
                                    control_nodes_in=g['cc_ctrl_in'],
                                    control_nodes_out=g['cc_ctrl_out']))
    ga, gb = _grid(T, N), _grid(T, N)
    sa = orc.ChainState(X0, ga, seed=0xABCDEF12345, chain=3, **kw)
    sb = orc.ChainState(X0, gb, seed=0xABCDEF12345, chain=3, **kw)
    for it in range(1, 7):
        sa.c.iter = it
        Xa = sa.sweep_c().copy()
        Xb = sb.sweep_py(orc.PhiloxDraws(0xABCDEF12345, 3, it), order='engine')
        np.testing.assert_allclose(Xa, Xb, rtol=0, atol=1e-12)
    np.testing.assert_allclose(ga.step_size, gb.step_size, rtol=1e-14)
    np.testing.assert_array_equal(ga.n_accepted, gb.n_accepted)
    np.testing.assert_array_equal(ga.steps_until_tune, gb.steps_until_tune)
    assert ga.n_accepted.sum() + (ga.n_steps.sum() - ga.n_accepted.sum()) > 0


def test_philox_c_equals_numpy_and_known_answer():
    import ctypes as C
    L = orc.lib()
    out = (C.c_uint32 * 4)()
    # Random123 known-answer vectors for philox4x32-10
    L.orc_philox4x32(0, 0, 0, 0, 0, C.byref(out))
    assert list(out) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    L.orc_philox4x32(0xffffffffffffffff, 0xffffffff, 0xffffffff, 0xffffffff,
                     0xffffffff, C.byref(out))
    assert list(out) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    L.orc_philox4x32(0x299f31d0a4093822, 0x243f6a88, 0x85a308d3, 0x13198a2e,
                     0x03707344, C.byref(out))
    assert list(out) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    rng = np.random.RandomState(0)
    for _ in range(20):
        seed = int(rng.randint(0, 2 ** 62))
        c = [int(v) for v in rng.randint(0, 2 ** 32, size=4, dtype=np.uint64)]
        L.orc_philox4x32(seed, *c, C.byref(out))
        got = [int(v) for v in orc.philox4x32(seed, *c)]
        assert list(out) == got
        u = (C.c_double * 2)()
        L.orc_philox_uniform2(seed, *c, C.byref(u))
        u1, u2 = orc.philox_uniform2(seed, *c)
        assert u[0] == float(u1) and u[1] == float(u2)
        assert 0.0 < u[0] <= 1.0


# ---------------------------------------------------------------- labels
def test_labels_mt_reproduces_reference(golden_sweeps):
    g = golden_sweeps
    z, n, nk, resp = orc.sample_labels_block_mt(
        g['lab_X'], g['lab_mu'], g['lab_sigma'], 0.8, g['lab_w'],
        np.random.RandomState(3))
    np.testing.assert_array_equal(z, g['lab_z'])
    np.testing.assert_array_equal(n, g['lab_n'])
    np.testing.assert_array_equal(nk, g['lab_nk'])
    np.testing.assert_array_equal(resp, g['lab_resp'])


def test_labels_c_philox_consistent(golden_sweeps):
    """C/Philox label update: counts consistent with z, and equal to the
    python restatement when it is fed the same uniforms."""
    g = golden_sweeps
    X, mu, sg, w = g['lab_X'], g['lab_mu'], g['lab_sigma'], g['lab_w']
    T, N, D = X.shape
    K = sg.shape[0]
    seed, chain, it = 99, 1, 7
    z, n, nk = orc.sample_labels_block_philox(X, mu, sg, 0.8, w, seed, chain, it)

    class _R(object):   # RandomState stand-in replaying the engine's uniforms
        def __init__(self):
            self.calls = 0

        def uniform(self, lo, hi):
            i, t = divmod(self.calls, T)
            self.calls += 1
            u, _ = orc.philox_uniform2(seed, i, t, it,
                                       orc.stream_word(chain, orc.STREAM_LABELS))
            return float(u) * hi
    z2, n2, nk2, _ = orc.sample_labels_block_mt(X, mu, sg, 0.8, w, _R())
    np.testing.assert_array_equal(z, z2)
    np.testing.assert_array_equal(n, n2)
    np.testing.assert_array_equal(nk, nk2)
    assert n.sum() == T * N and (nk.sum(axis=1) == N).all()


# ---------------------------------------------------------------- fit traces
def _rng_from(g, pre):
    rng = np.random.RandomState(0)
    rng.set_state(('MT19937', g[pre + 'rng_keys'], int(g[pre + 'rng_pos']),
                   int(g[pre + 'rng_has_gauss']), float(g[pre + 'rng_cached'])))
    return rng


@pytest.mark.parametrize('case', ['monks_u_', 'monks_d_', 'monks_cc_'])
def test_lsm_loop_reproduces_reference_fit(golden_fits, monks, case):
    """the whole per-iteration glue (a8-a10, a13-a15) in the reference's order
    with its MT19937 stream reproduces DynamicNetworkLSM.fit() on monks."""
    g = golden_fits
    directed = case != 'monks_u_'
    Y = monks['Y_directed'] if directed else monks['Y_undirected']
    Xs, ics, logps = g[case + 'Xs'], g[case + 'intercepts'], g[case + 'logps']
    n_total = Xs.shape[0]
    T, N, D = Xs.shape[1:]
    tune, burn, tune_interval = 4, 2, 2
    grid = orc.SamplerGrid(T, N, step_size=0.1, tune=tune,
                           tune_interval=tune_interval)
    cc = None
    if directed:
        isamp = [orc.ScalarMetropolis(0.1, tune, tune_interval) for _ in range(2)]
        rsamp = orc.ScalarMetropolis(175000, None, 100)
        radii0 = g[case + 'radiis'][0]
        if case == 'monks_cc_':
            cc = dict(in_edges=g[case + 'cc_in_edges'],
                      out_edges=g[case + 'cc_out_edges'],
                      degree=g[case + 'cc_degrees'],
                      control_nodes_in=g[case + 'cc_ctrl_in'],
                      control_nodes_out=g[case + 'cc_ctrl_out'])
    else:
        isamp = [orc.ScalarMetropolis(0.1, tune, 100)]   # lsm.py:465-467
        rsamp, radii0 = None, None
    got = orc.lsm_reference_loop(
        Y, Xs[0], ics[0], _rng_from(g, case), n_total, tune + burn,
        float(g[case + 'tau_sq']), 0.1, g[case + 'intercept_prior'], 2.0,
        grid, isamp, radii0=radii0, radii_sampler=rsamp, case_control=cc,
        logp0=logps[0])
    np.testing.assert_allclose(got[0], Xs, rtol=0, atol=1e-9)
    np.testing.assert_allclose(got[1], ics, rtol=0, atol=1e-9)
    if directed:
        np.testing.assert_allclose(got[2], g[case + 'radiis'], rtol=1e-9)
    np.testing.assert_allclose(got[3][1:], logps[1:], rtol=1e-9)
    np.testing.assert_allclose(grid.step_size, g[case + 'step'], rtol=1e-13)
    np.testing.assert_array_equal(grid.n_accepted, g[case + 'nacc'])
    np.testing.assert_array_equal(grid.steps_until_tune, g[case + 'until'])
