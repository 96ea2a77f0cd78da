"""n_features = 5 .. 8 (round-4 verdict, missing 4: the reference takes any n_features, lsm.py:235,254;
the engine's kernels are templates of it, built for 1..8 - DLSM_D_MAX in csrc/device_common.hpp).

Every row of the path at the wide dimensions, against the oracle - which tests/test_oracle_golden.py
and tests/test_init_oracle_golden.py pin to the reference itself at n_features 5, 6 and 8
(tests/golden/wide_*.npz: `python tests/golden/make_golden.py wide`).  The cases are those of the
1..4 tests, called with the wide dimensions.  Above four dimensions the slice-in-LDS sweep (algo 1)
the speculative-batch sweeps (2, 3) and the pipelined sweeps (4, 5) run.
"""
import numpy as np
import pytest

import test_gpu_parity as P
import test_gpu_hdp_loop as H
import test_gpu_post as POST
import test_gpu_forecast as FC
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

WIDE = [5, 6, 8]


@pytest.fixture(scope='module')
def eng():
    import dynetlsm_amd
    return dynetlsm_amd


# ------------------------------------------------------------ a4 - a6: the log-likelihood passes
@pytest.mark.parametrize('D', WIDE)
@pytest.mark.parametrize('N', [7, 129, 300])
def test_loglik_full_undirected(eng, N, D):
    P.test_loglik_full_undirected(eng, N, D)


@pytest.mark.parametrize('D', WIDE)
@pytest.mark.parametrize('N', [7, 130])
def test_loglik_full_directed(eng, N, D):
    P.test_loglik_full_directed(eng, N, D)


@pytest.mark.parametrize('D', WIDE)
@pytest.mark.parametrize('N,C', [(12, 3), (200, 70)])
def test_loglik_case_control(eng, N, C, D):
    X, Yd, Yu, radii = P._rand_net(N + C + D, 2, N, D, density=0.05, scale=0.05)
    cc = P._cc_lists(Yd, C, 3)
    with eng.Chain(2, N, D, 'case_control') as c:
        c.upload_edges(cc['in_edges'], cc['out_edges'], cc['degree'])
        c.set_controls(cc['control_nodes_in'], cc['control_nodes_out'])
        c.set_positions(X); c.set_radii(radii); c.set_intercepts([0.3, 0.7])
        got = c.loglik_full([[0.3, 0.7], [0.9, 0.1]])
        want = [orc.approx_directed_network_loglikelihood(
            X, radii, cc['in_edges'], cc['out_edges'], cc['degree'],
            cc['control_nodes_out'], a, b) for a, b in [(0.3, 0.7), (0.9, 0.1)]]
        np.testing.assert_allclose(got, want, rtol=P.RTOL_LL)
        pa = c.loglik_partial_all()
        wantp = np.array([[orc.approx_directed_partial_loglikelihood(
            X[t], radii, cc['in_edges'][t], cc['out_edges'][t], cc['degree'][t],
            cc['control_nodes_in'][t], cc['control_nodes_out'][t], 0.3, 0.7, j)
            for j in range(N)] for t in range(2)])
        np.testing.assert_allclose(pa, wantp, rtol=1e-11)


# ------------------------------------------------------------ a1 - a3: the sweeps
@pytest.mark.parametrize('D', WIDE)
@pytest.mark.parametrize('prior', ['rw', 'mix'])
@pytest.mark.parametrize('name', ['undirected', 'directed', 'case_control'])
def test_sweep_slice_small(eng, name, prior, D):
    P._sweep_case(eng, name, prior, T=3, N=10, D=D, n_sweeps=6, algo=1,
                  scale=1.0 if name == 'undirected' else 0.05)


@pytest.mark.parametrize('D', WIDE)
@pytest.mark.parametrize('name,N,algo', [('undirected', 300, 1), ('undirected', 300, 2), ('undirected', 700, 3),
                                         ('directed', 260, 1), ('directed', 300, 2),
                                         ('case_control', 300, 1), ('case_control', 300, 2),
                                         ('undirected', 300, 0), ('case_control', 2300, 0),
                                         ('undirected', 700, 4), ('directed', 600, 4), ('case_control', 700, 4)])
def test_sweep_medium_every_algorithm_the_wide_dimensions_have(eng, name, N, algo, D):
    """crosses the proposal-chunk (256) and the workgroup (1024) boundaries; algo 0 resolves to the
    speculative batches from 256 nodes on (capi.hip resolve_sweep_algo), case-control N = 2300 included"""
    big = N > 1000
    P._sweep_case(eng, name, 'rw', T=2 if big else 4, N=N, D=D, n_sweeps=2 if big else 3, algo=algo,
                  scale=1.0 if name == 'undirected' else 0.05,
                  cc_C=12 if big else 20,
                  density=(0.004 if big else 0.05) if name == 'case_control' else 0.2)


@pytest.mark.parametrize('name,D', [('undirected', 6), ('directed', 5), ('undirected', 8)])
def test_sweep_at_the_headline_size(eng, name, D):
    """T = 2, N = 2000 (BASELINE.json's N) at a wide dimension: the speculative-batch sweep the automatic choice
    resolves to, against the oracle"""
    P._sweep_case(eng, name, 'rw', T=2, N=2000, D=D, n_sweeps=1, algo=0,
                  scale=1.0 if name == 'undirected' else 0.05, density=0.03)


def test_sweep_algorithms_of_the_wide_dimensions(eng):
    """the automatic choice is the one of n_features <= 4: the pipelined sweeps (algo 4, 5) run to 8"""
    with eng.Chain(2, 600, 5, 'undirected') as c:
        assert c.resolve_sweep_algo(0) == 4
    with eng.Chain(2, 100, 8, 'undirected') as c:
        assert c.resolve_sweep_algo(0) == 1
    with eng.Chain(2, 3000, 6, 'case_control') as c:
        assert c.resolve_sweep_algo(0) == 5 and c.resolve_sweep_algo(4) == 4


@pytest.mark.parametrize('D', WIDE + [7])
@pytest.mark.parametrize('T,N,C,density,prior', [(3, 2300, 12, 0.004, 'rw'), (2, 1025, 30, 0.01, 'mix')])
def test_sweep_case_control_sparse_lists(eng, T, N, C, density, prior, D):
    """algo 5 (k_ccpipe_step) at the wide dimensions: records of 8 / 12 doubles, 64 window terms per flush from
    d = 6 on; several batches of 512 nodes, both priors"""
    P._sweep_case(eng, 'case_control', prior, T=T, N=N, D=D, n_sweeps=2, algo=5, scale=0.05, cc_C=C, density=density)


# ------------------------------------------------------------ a14: centring and Procrustes
@pytest.mark.parametrize('D', WIDE)
def test_center_and_procrustes(eng, D):
    rng = np.random.RandomState(D)
    X = rng.randn(3, 60, D)
    Q, _ = np.linalg.qr(rng.randn(D, D))
    Xref = X.dot(Q) + 0.01 * rng.randn(*X.shape)
    with eng.Chain(3, 60, D, 'undirected') as c:
        c.set_positions(X)
        c.center()
        np.testing.assert_allclose(c.get_positions(), orc.center(X), atol=1e-13)
        c.set_positions(X)
        R = c.procrustes(Xref)
        Xw, Rw = orc.procrustes_rotation(Xref, X)
        np.testing.assert_allclose(R, Rw, atol=1e-11)
        np.testing.assert_allclose(R.T.dot(R), np.eye(D), atol=1e-12)
        np.testing.assert_allclose(c.get_positions(), Xw, atol=1e-11)


@pytest.mark.parametrize('D,N', [(5, 40), (8, 33)])
def test_device_procrustes_of_a_trace(eng, D, N):
    POST.test_device_procrustes_other_dimensions(eng, D, N)


# ------------------------------------------------------------ a12: labels; 8f-2: label sums
@pytest.mark.parametrize('T,N,D,K', [(4, 100, 5, 33), (3, 77, 8, 64), (6, 45, 6, 7), (10, 301, 5, 40)])
def test_labels(eng, T, N, D, K):
    P.test_labels_shapes_against_oracle(eng, T, N, D, K)


@pytest.mark.parametrize('T,N,D,K', [(4, 300, 5, 7), (3, 1000, 8, 20)])
def test_hdp_label_sums(eng, T, N, D, K):
    P.test_hdp_label_sums_match_oracle(eng, T, N, D, K)


# ------------------------------------------------------------ the device-resident loops
@pytest.mark.parametrize('D', WIDE)
@pytest.mark.parametrize('N,algo', [(18, 1), (300, 1), (300, 2), (700, 0), (700, 4)])
def test_lsm_device_loop_equals_oracle_iterations(eng, monks, N, algo, D):
    P.lsm_loop_case(eng, monks, N, algo, D)


@pytest.mark.parametrize('D', [5, 8])
@pytest.mark.parametrize('name,N,algo', [('directed', 30, 1), ('directed', 300, 0), ('case_control', 300, 0),
                                         ('case_control', 40, 1), ('case_control', 700, 5)])
def test_lsm_directed_device_loop_equals_oracle_iterations(eng, name, N, algo, D):
    P.directed_loop_case(eng, name, N, algo, D)


@pytest.mark.parametrize('T,N,K,D,seed', [(3, 45, 5, 5, 22), (2, 50, 33, 8, 23), (3, 300, 6, 6, 24)])
def test_hdp_device_loop(eng, T, N, K, D, seed):
    H._run_both(eng, T, N, K, seed, n_it=3, D=D)


# ------------------------------------------------------------ forecasts
@pytest.mark.parametrize('N,D,S', [(40, 5, 7), (130, 8, 5)])
def test_forecast_kernels(eng, N, D, S):
    FC.test_kernels_match_oracle(eng, N, D, S)


# ------------------------------------------------------------ fit() end to end
def test_fit_from_its_own_initialisation_at_five_dimensions(eng, golden_init):
    """DynamicNetworkLSM(n_features=5).fit(Y) and the HDP-LPCM's, no init=: GMDS + MLE on the device,
    the device-resident loop, the trace's post-processing"""
    Y = golden_init['u5_Y']
    T, N = Y.shape[:2]
    m = eng.DynamicNetworkLSM(n_iter=60, tune=30, burn=30, n_features=5, random_state=4).fit(Y)
    assert np.isfinite(m.logps_).all()
    assert m.X_.shape == (T, N, 5) and m.Xs_.shape == (120, T, N, 5)
    # the chain moves off the GMDS + MLE start (the posterior's bulk sits below its mode's log-density: 450 free
    # coordinates) and mixes: a third to two thirds of the intercept's proposals are taken after tuning
    assert np.unique(m.intercepts_[60:, 0]).size > 10 and np.abs(m.Xs_[-1] - m.Xs_[0]).max() > 0.1
    m.chain_.close()
    h = eng.DynamicNetworkHDPLPCM(n_iter=40, tune=20, burn=20, n_features=5, n_components=6,
                                  random_state=5).fit(Y)
    assert h.loop_kind_ == 'device-resident' and np.isfinite(h.logps_).all()
    assert h.X_.shape == (T, N, 5) and h.mus_.shape[-1] == 5 and h.z_.shape == (T, N)
    h.chain_.close()
    lp = eng.DynamicNetworkLPCM(n_iter=40, tune=20, burn=20, n_features=6, n_components=3, random_state=6).fit(Y)
    assert np.isfinite(lp.logps_).all() and lp.X_.shape == (T, N, 6) and lp.mus_.shape[-1] == 6
    lp.chain_.close()
    with pytest.raises(ValueError, match='1 <= n_features <= 8'):
        eng.DynamicNetworkLSM(n_features=9).fit(Y)
