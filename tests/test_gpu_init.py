"""Initialisation pipeline on the device against the CPU oracle
(oracle/init_oracle.py) and the reference's own outputs (tests/golden/init.npz).
Needs an MI355X: -m gpu.

Tolerances: hop counts are integers -> exact.  SMACOF and the eigen step are
float64 fixed-point / eigenvector computations whose summation order differs from
numpy's: 1e-8 of the configuration scale against the oracle on the same inputs;
2e-6 against the reference end to end (sklearn computes distances through the
|x|^2 + |y|^2 - 2xy expansion, which alone costs ~1e-7)."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import init_oracle as io

pytestmark = pytest.mark.gpu

from conftest import INIT_CASES as CASES       # noqa: E402


@pytest.fixture(scope='module')
def eng():
    import dynetlsm_amd
    from dynetlsm_amd import _lib
    _lib.load()
    assert _lib.device_count() >= 1, 'no HIP device: the engine has no CPU path'
    return dynetlsm_amd


def _chain(eng, Y, D, directed):
    T, N, _ = Y.shape
    c = eng.Chain(T, N, D, 'directed' if directed else 'undirected', seed=7)
    c.upload_network(Y)
    return c


def _sparse_net(seed, T, N, directed, mean_degree=3.0):
    rng = np.random.RandomState(seed)
    Y = (rng.rand(T, N, N) < mean_degree / N).astype(np.float64)
    if not directed:
        Y = np.triu(Y, 1)
        Y = Y + Y.transpose(0, 2, 1)
    for t in range(T):
        np.fill_diagonal(Y[t], 0)
    return Y


@pytest.mark.parametrize('tag,directed,D', CASES)
def test_hops_match_reference(eng, golden_init, tag, directed, D):
    Y = golden_init[tag + '_Y']
    with _chain(eng, Y, D, directed) as c:
        c.init_shortest_paths()
        for t in range(Y.shape[0]):
            np.testing.assert_array_equal(c.init_get_dissimilarity(t),
                                          golden_init[tag + '_D'][t])


@pytest.mark.parametrize('directed', [False, True])
@pytest.mark.parametrize('N', [33, 257, 700])
def test_hops_sparse_disconnected(eng, N, directed):
    """sparse graphs: long paths, several components, isolated nodes, N not a
    multiple of the 32-bit words or of the wavefront"""
    Y = _sparse_net(N, 2, N, directed, mean_degree=1.6 if N > 100 else 2.5)
    Y[1] = 0                                     # an empty slice: every pair imputed
    with _chain(eng, Y, 2, directed) as c:
        c.init_shortest_paths()
        for t in range(2):
            np.testing.assert_array_equal(c.init_get_dissimilarity(t), io.hop_matrix(Y[t]))
        assert c.init_get_dissimilarity(1).max() == 1.0


@pytest.mark.parametrize('tag,directed,D', CASES)
def test_smacof_same_start_as_oracle(eng, golden_init, tag, directed, D):
    Y = golden_init[tag + '_Y']
    N = Y.shape[1]
    rng = np.random.RandomState(3)
    X0 = rng.uniform(size=(3, N, D))
    with _chain(eng, Y, D, directed) as c:
        c.init_shortest_paths()
        for t in (0, 1):
            delta = golden_init[tag + '_D'][t]
            X, stress, n_iter = c.init_smacof(t, X0)
            for r in range(3):
                Xo, so, no = io.smacof_single(delta, X0[r])
                assert n_iter[r] == no
                np.testing.assert_allclose(stress[r], so, rtol=1e-9)
                assert np.abs(X[r] - Xo).max() < 1e-8 * np.abs(Xo).max()


def test_smacof_max_iter_and_eps(eng, golden_init):
    Y = golden_init['u_Y']
    N = Y.shape[1]
    X0 = np.random.RandomState(0).uniform(size=(1, N, 2))
    with _chain(eng, Y, 2, False) as c:
        c.init_shortest_paths()
        for max_iter, eps in [(1, 1e-6), (2, 1e-6), (7, 1e-12), (300, 1e-3), (40, 0.0)]:
            X, stress, n_iter = c.init_smacof(0, X0, max_iter=max_iter, eps=eps)
            Xo, so, no = io.smacof_single(golden_init['u_D'][0], X0[0], max_iter, eps)
            assert n_iter[0] == no
            np.testing.assert_allclose(stress[0], so, rtol=1e-9)
            assert np.abs(X[0] - Xo).max() < 1e-8 * np.abs(Xo).max()


@pytest.mark.parametrize('tag,directed,D', CASES)
def test_gmds_step_matches_eigh(eng, golden_init, tag, directed, D):
    Y = golden_init[tag + '_Y']
    Xg = golden_init[tag + '_X'] * (Y.shape[1] if directed else 1.0)   # undo X /= N
    with _chain(eng, Y, D, directed) as c:
        c.init_shortest_paths()
        for t in range(1, Y.shape[0]):
            for lmbda in (10.0, 0.5):
                X, evals, info = c.init_gmds_step(t, Xg[t - 1], lmbda=lmbda)
                Xo, eo = io.gmds_step(golden_init[tag + '_D'][t], Xg[t - 1], lmbda)
                np.testing.assert_allclose(evals, eo, rtol=1e-10)
                assert np.abs(X - Xo).max() < 1e-8 * np.abs(Xo).max(), (t, lmbda, info)


@pytest.mark.parametrize('tag,directed,D', CASES)
def test_generalized_mds_matches_reference(eng, golden_init, tag, directed, D):
    from dynetlsm_amd import initialization as init_mod
    Y = golden_init[tag + '_Y']
    with _chain(eng, Y, D, directed) as c:
        X = init_mod.generalized_mds(c, is_directed=directed,
                                     random_state=np.random.RandomState(int(golden_init[tag + '_seed'])))
    Xr = golden_init[tag + '_X']
    assert np.abs(X - Xr).max() < 2e-6 * np.abs(Xr).max()


def test_mle_sums_and_mles(eng, golden_init):
    from dynetlsm_amd import initialization as init_mod
    g = golden_init
    for tag, D in (('u', 2), ('u3', 3), ('u5', 5), ('u8', 8)):
        Y, X = g[tag + '_Y'], g[tag + '_X']
        with _chain(eng, Y, D, False) as c:
            c.set_positions(X)
            for p, f, gr in zip(g[tag + '_mle_points'], g[tag + '_mle_f'], g[tag + '_mle_g']):
                s = c.init_mle_sums(p[0], p[1])
                np.testing.assert_allclose(s[0], f, rtol=1e-11)
                np.testing.assert_allclose(s[1:], gr, rtol=1e-10, atol=1e-9)
            for sq in (0, 1):
                c.set_squared(sq)
                np.testing.assert_allclose(c.init_mle_sums(0.2, 0.7),
                                           io.mle_sums_undirected(Y, X, 0.2, 0.7, sq),
                                           rtol=1e-10, atol=1e-9)
            c.set_squared(0)
            got = init_mod.scale_intercept_mle(c, X)
            np.testing.assert_allclose(got, g[tag + '_mle'], rtol=1e-5, atol=1e-6)
    for tag, D in (('d', 2), ('d6', 6)):
        Y, X, radii = g[tag + '_Y'], g[tag + '_X'], g[tag + '_radii']
        with _chain(eng, Y, D, True) as c:
            c.set_positions(X)
            c.set_radii(radii)
            for p, f in zip(g[tag + '_mle_points'], g[tag + '_mle_f']):
                s = c.init_mle_sums(p[0], p[1])
                np.testing.assert_allclose(s[0], f, rtol=1e-11)
                # the reference's in-gradient is uninitialised (see the oracle's test)
                np.testing.assert_allclose(s, io.mle_sums_directed(Y, X, radii, p[0], p[1]),
                                           rtol=1e-10, atol=1e-9)
            got = init_mod.directed_intercept_mle(c, X, radii)
            np.testing.assert_allclose(got, g[tag + '_mle'], rtol=1e-5, atol=1e-6)


def test_init_errors(eng):
    Y = _sparse_net(1, 2, 20, False)
    c = eng.Chain(2, 20, 2, 'undirected')
    with pytest.raises(eng.EngineError):
        c.init_shortest_paths()                  # network not uploaded
    c.upload_network(Y)
    with pytest.raises(eng.EngineError):
        c.init_smacof(0, np.zeros((1, 20, 2)))   # hop matrices not computed
    c.init_shortest_paths()
    c.upload_network(Y)                          # a new network invalidates them
    with pytest.raises(eng.EngineError):
        c.init_get_dissimilarity(0)
    c.close()


def test_fit_without_init_runs_device_pipeline(eng, golden_init):
    """fit(Y) end to end: the starting values come from the device pipeline"""
    Y = golden_init['u_Y']
    m = eng.DynamicNetworkLSM(n_iter=20, tune=10, burn=10, random_state=4).fit(Y)
    assert np.isfinite(m.logps_).all()
    assert m.X_.shape == (Y.shape[0], Y.shape[1], 2)


@pytest.mark.parametrize('tag', ['a', 'b', 'c', 'd'])
def test_longitudinal_kmeans_lloyd_on_device_matches_reference(tag):
    """latent_space.py:98-137 with the Lloyd iterations on the device (k_kmeans_assign /
    k_kmeans_update, scikit-learn's _kmeans_single_lloyd loop) against the reference's own outputs
    (tests/golden/kmeans.npz): same labels, centres and variances, and the caller's RandomState
    ends where scikit-learn's KMeans would leave it"""
    import dynetlsm_amd as eng
    from dynetlsm_amd import initialization as im
    g = load_golden('kmeans.npz')
    X, K = g[tag + '_X'], int(g[tag + '_K'])
    T, N, D = X.shape
    rs = np.random.RandomState(int(g[tag + '_seed']))
    with eng.Chain(1, N, D, 'undirected') as c:
        centers, variances, labels = im.longitudinal_kmeans(X, n_clusters=K, random_state=rs, chain=c)
    np.testing.assert_array_equal(labels, g[tag + '_labels'])
    np.testing.assert_allclose(centers, g[tag + '_centers'], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(variances, g[tag + '_variances'], rtol=1e-12)
    assert rs.rand() == float(g[tag + '_next_draw'])


def test_kmeans_lloyd_reports_an_emptied_cluster():
    """a seeding that starves a cluster: the device loop stops and says so (scikit-learn's
    relocation rule is then applied on the host by the caller)"""
    import dynetlsm_amd as eng
    rng = np.random.RandomState(0)
    Xc = np.vstack([rng.randn(50, 4) * 0.1 - 3.0, rng.randn(50, 4) * 0.1 + 3.0])
    Xc -= Xc.mean(axis=0)
    seeds = np.vstack([Xc[0], Xc[60], Xc[60] + 40.0])           # nobody is nearest to the third
    with eng.Chain(1, 100, 2, 'undirected') as c:
        assert c.init_kmeans_lloyd(Xc, seeds, max_iter=50, tol=0.0) is None
        cen, lab, nit = c.init_kmeans_lloyd(Xc, seeds[:2], max_iter=50, tol=0.0)
    assert (lab[:50] == 0).all() and (lab[50:] == 1).all() and nit == 2
    np.testing.assert_allclose(cen[0], Xc[:50].mean(axis=0), rtol=1e-13)
