"""Config 4 (BASELINE.json configs[3]: directed case-control likelihood, T=5, N=10 000, d=2,
n_control=100) through the device-resident loop at FULL size, inside the `-m gpu` suite:

  * `dlsm_lsm_run` against the oracle's restatement of the same iteration
    (sample_latent_positions.py:92-146 with directed_likelihoods_fast.pyx:83-182, then
    sample_coefficients.py:12-121 around directed_likelihoods_fast.pyx:208-270) iteration by
    iteration, across a control resample (case_control_likelihood.py:27-33, 75-112);
  * two chains with different Philox chain ids: split R-hat of both intercepts and of the
    log-posterior trace.
"""
import time

import numpy as np
import pytest

from mcmc_diag import effective_n, split_rhat

pytestmark = pytest.mark.gpu

T, N, D, C = 5, 10000, 2, 100
SEED = 20240229


@pytest.fixture(scope='module')
def eng():
    import dynetlsm_amd
    return dynetlsm_amd


@pytest.fixture(scope='module')
def c4_tables():
    from dynetlsm_amd.synthetic import synthetic_sparse_directed
    return synthetic_sparse_directed(T, N, 20, 0)


def test_c4_device_loop_equals_oracle_iterations_at_full_size(eng, c4_tables):
    """five iterations of the directed case-control loop at T=5, N=10 000 (sparse pipelined sweep,
    algo 5; centring + Procrustes; intercept_in, intercept_out and radii steps around the
    case-control likelihood passes), the controls redrawn on the device before the third: positions,
    both intercepts, radii, log-posteriors and the samplers' counters against the C / numpy oracle
    replaying the same Philox draws with the same control tables"""
    from oracle import oracle as orc
    X, radii, degree, in_edges, out_edges = c4_tables
    b0 = np.array([1.0, 0.5])
    prior_b, var_b = np.array([1.0, 0.5]), 2.0
    n_total, it_resample = 6, 3
    gg = eng.SamplerGrid(T, N, 0.002, tune=4, tune_interval=2)
    og = orc.SamplerGrid(T, N, 0.002, tune=4, tune_interval=2)
    with eng.Chain(T, N, D, 'case_control', seed=SEED, chain_id=3) as c:
        c.upload_edges(in_edges, out_edges, degree)
        c.resample_controls(0, C)
        ci0, co0 = c.get_controls()
        c.set_positions(X); c.set_radii(radii); c.set_intercepts(b0)
        c.set_prior_random_walk(1e-4, 1e-5); c.set_samplers(gg)
        assert c.resolve_sweep_algo(0) == 5
        c.lsm_configure(prior_b, var_b, step_size_intercept=0.1, tune=4, tune_interval=2,
                        n_iter_procrustes=0, sweep_algo=0, step_size_radii=175000., radii_tune=4,
                        radii_tune_interval=2)
        lp0 = (orc.approx_directed_network_loglikelihood(X, radii, in_edges, out_edges, degree, co0,
                                                         b0[0], b0[1]) +
               orc.lsm_log_prior(X, 1e-4, 1e-5, b0, prior_b, var_b))
        c.trace_alloc(n_total, logp0=float(lp0))
        t0 = time.perf_counter()
        c.lsm_run(1, it_resample - 1, procrustes_ref=0)
        c.resample_controls(it_resample, C)
        ci1, co1 = c.get_controls()
        c.lsm_run(it_resample, n_total - it_resample, procrustes_ref=0)
        Xs, ics, lps = c.trace_read(0, n_total)
        secs = time.perf_counter() - t0
        rs = c.trace_read_radii(0, n_total)
        cfg = c.lsm_get_config()
        c.get_samplers(gg)
    # the redraw really changed the tables (Philox stream of iteration 3), and they are valid
    assert not np.array_equal(co0, co1) and not np.array_equal(ci0, ci1)
    for t, i in [(0, 0), (4, 9999), (2, 5000)]:
        for arr, col, edges in ((co1, 1, out_edges), (ci1, 0, in_edges)):
            v = arr[t, i]
            assert (v >= 0).all() and len(set(v)) == C and i not in v
            assert not set(v) & set(edges[t, i, :degree[t, i, col]])
    cc = dict(in_edges=in_edges, out_edges=out_edges, degree=degree,
              control_nodes_in=ci0.copy(), control_nodes_out=co0.copy())
    st = orc.ChainState(X, og, model=2, intercept=b0, radii=radii.copy(), case_control=cc,
                        tau_sq=1e-4, sigma_sq=1e-5, seed=SEED, chain=3)
    controls = {'out': co0}

    def loglik(Xc, b, r):
        return orc.approx_directed_network_loglikelihood(Xc, r, in_edges, out_edges, degree,
                                                         controls['out'], b[0], b[1])
    isamp = [orc.ScalarMetropolis(0.1, 4, 2) for _ in range(2)]
    rsamp = orc.ScalarMetropolis(175000., 4, 2)
    want = dict(X=[X.copy()], b=[b0.copy()], r=[radii.copy()], lp=[lp0])
    t0 = time.perf_counter()
    for it in range(1, n_total):
        if it == it_resample:           # the oracle's chain reads the tables it was built on
            st._keep[3][...] = ci1
            st._keep[4][...] = co1
            controls['out'] = co1
        lp = orc.lsm_iteration_directed(st, it, loglik, isamp, rsamp, prior_b, var_b, X_ref=X)
        want['X'].append(st.X.copy()); want['b'].append(st.intercept.copy())
        want['r'].append(st.radii.copy()); want['lp'].append(lp)
    print('C4 loop: engine %.3f s, oracle %.1f s for %d iterations' % (secs, time.perf_counter() - t0,
                                                                      n_total - 1))
    np.testing.assert_allclose(Xs, np.array(want['X']), atol=1e-11)
    np.testing.assert_allclose(ics, np.array(want['b']), atol=1e-12)
    np.testing.assert_allclose(rs, np.array(want['r']), rtol=1e-10, atol=1e-18)
    np.testing.assert_allclose(lps, want['lp'], rtol=1e-10)
    np.testing.assert_array_equal(gg.n_accepted, og.n_accepted)
    np.testing.assert_array_equal(gg.n_steps, og.n_steps)
    np.testing.assert_allclose(gg.step_size, og.step_size, rtol=1e-14)
    for k in range(2):
        assert cfg.i_n_steps[k] == isamp[k].n_steps and cfg.i_n_accepted[k] == isamp[k].n_accepted
        np.testing.assert_allclose(cfg.i_step_size[k], isamp[k].step_size, rtol=1e-14)
    assert cfg.r_n_steps == rsamp.n_steps and cfg.r_n_accepted == rsamp.n_accepted
    np.testing.assert_allclose(cfg.r_step_size, rsamp.step_size, rtol=1e-14)
    # the chain moved: positions everywhere, and the log-posterior with them
    assert 0.05 < og.n_accepted.sum() / float(og.n_steps.sum()) < 0.98
    assert not np.array_equal(Xs[n_total - 1], Xs[it_resample - 1])
