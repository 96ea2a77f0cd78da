"""Starting values for the chains, computed on the device.

The reference builds them with scipy / scikit-learn on the host
(latent_space.py:36-95, lsm.py:32-97); at N = 2000 that costs minutes, far more
than the chain itself.  Here the O(N^2)-and-up parts are kernels of the engine
(csrc/kernels_init.hpp) behind ``Chain.init_*``:

* shortest-path dissimilarities: bitset BFS on the packed network;
* the first slice's metric MDS: SMACOF, all ``n_init`` starts concurrently, from
  the same ``RandomState.uniform`` starts sklearn would draw;
* the Sarkar-Moore eigen steps: Lanczos on the implicit double-centred matrix;
* objective + gradient of the conditional MLEs in one fused pass.

The host keeps what is O(1) or O(N): the 2-parameter BFGS driver
(scipy.optimize.minimize, as the reference), the radii (a degree count) and
seeding of longitudinal k-means (its Lloyd iterations run on the device).
"""
import warnings

import numpy as np
from scipy.optimize import minimize

__all__ = ['generalized_mds', 'initialize_radii', 'scale_intercept_mle',
           'directed_intercept_mle', 'longitudinal_kmeans']

# sklearn.manifold.MDS defaults of the scikit-learn (1.7) the fixtures were made with
MDS_N_INIT, MDS_MAX_ITER, MDS_EPS = 4, 300, 1e-6


def generalized_mds(chain, is_directed=False, lmbda=10, random_state=None,
                    max_lanczos=256, tol=1e-12):
    """Sarkar & Moore (2005) generalised MDS (latent_space.py:47-95) of the network
    uploaded to ``chain``; returns X (T, N, D).  ``random_state`` is consumed as
    sklearn's MDS consumes it (n_init uniform starting configurations)."""
    T, N, D = chain.T, chain.N, chain.D
    rng = (random_state if isinstance(random_state, np.random.RandomState)
           else np.random.RandomState(random_state))
    chain.init_shortest_paths()
    X0 = np.stack([rng.uniform(size=N * D).reshape(N, D) for _ in range(MDS_N_INIT)])
    Xs, stress, _ = chain.init_smacof(0, X0, max_iter=MDS_MAX_ITER, eps=MDS_EPS)
    X = np.empty((T, N, D))
    X[0] = Xs[int(np.argmin(stress))]          # first minimum, as sklearn's `<`
    for t in range(1, T):
        X[t], _, info = chain.init_gmds_step(t, X[t - 1], lmbda=lmbda,
                                             max_lanczos=max_lanczos, tol=tol)
        if info['residual'] > 1e3 * tol:
            warnings.warn('generalized_mds: Lanczos residual %.2e at t=%d after %d '
                          'vectors' % (info['residual'], t, info['n_lanczos']))
    chain.init_release()
    if is_directed:
        X /= N           # same scale as the radii (latent_space.py:92-93)
    return X


def initialize_radii(Y, reg=1e-5):
    radii = 0.5 * (Y.sum(axis=(0, 1)) + Y.sum(axis=(0, 2)))
    radii /= Y.sum()
    if np.any(radii == 0.):
        radii += reg
        radii /= np.sum(radii)
    return radii


def scale_intercept_mle(chain, X, tol=1e-4):
    """Conditional MLE of (log scale, intercept) of the undirected model by BFGS
    (lsm.py:47-70); objective and gradient (lsm.py:32-44) are one fused pass over
    the dyads on the device per evaluation."""
    chain.set_positions(X)

    def fun(x):
        s = chain.init_mle_sums(x[0], x[1])
        return -s[0], -s[1:]

    res = minimize(fun=fun, x0=np.array([0.0, 1.0]), method='BFGS', jac=True, tol=tol)
    return res.x[0], res.x[1]


def directed_intercept_mle(chain, X, radii, tol=1e-4):
    """Conditional MLE of (intercept_in, intercept_out) (lsm.py:73-97)."""
    chain.set_positions(X)
    chain.set_radii(radii)

    def fun(x):
        s = chain.init_mle_sums(x[0], x[1])
        return -s[0], -s[1:]

    res = minimize(fun=fun, x0=np.array([0.0, 0.0]), method='BFGS', jac=True, tol=tol)
    return res.x[0], res.x[1]


def kmeans_plusplus_seeds(Xc, n_clusters, rs):
    """scikit-learn 1.7's ``_kmeans_plusplus`` (Arthur & Vassilvitskii with 2 + log k local
    trials) restated in numpy - the same draws from ``rs`` in the same order (``choice``, then one
    ``uniform(size=n_local_trials)`` per centre), the same distance expression
    (|x|^2 - 2 x.y + |y|^2 through the same BLAS product, clipped at 0), the same cumulative sums -
    so that the seeds, and with them labels and centres, are the library's bit for bit
    (a CPU test holds it against ``sklearn.cluster.kmeans_plusplus`` itself) without
    importing ``sklearn.cluster`` (half a second, half of what ``fit`` spends before its loop)."""
    n, _ = Xc.shape
    x2 = np.einsum('ij,ij->i', Xc, Xc)          # sklearn.utils.extmath.row_norms(squared=True)
    w = np.ones(n)

    def dist_sq(C):                              # sklearn.metrics.pairwise._euclidean_distances
        d = -2 * (C @ Xc.T)
        d += np.einsum('ij,ij->i', C, C)[:, np.newaxis]
        d += x2[np.newaxis, :]
        np.maximum(d, 0, out=d)
        return d
    centers = np.empty((n_clusters, Xc.shape[1]))
    n_local_trials = 2 + int(np.log(n_clusters))
    center_id = rs.choice(n, p=w / w.sum())
    centers[0] = Xc[center_id]
    closest = dist_sq(centers[0, np.newaxis])
    current_pot = closest @ w
    for c in range(1, n_clusters):
        rand_vals = rs.uniform(size=n_local_trials) * current_pot
        cand = np.searchsorted(np.cumsum(w * closest, dtype=np.float64).ravel(), rand_vals)
        np.clip(cand, None, closest.size - 1, out=cand)
        d = dist_sq(Xc[cand])
        np.minimum(closest, d, out=d)
        pots = d @ w.reshape(-1, 1)
        best = np.argmin(pots)
        current_pot = pots[best]
        closest = d[best]
        centers[c] = Xc[cand[best]]
    return centers


def _kmeans(X_vec, n_clusters, random_state, chain):
    """``KMeans(n_clusters, random_state=random_state).fit(X_vec)`` (scikit-learn >= 1.4 defaults:
    k-means++ seeding, one start, Lloyd, tol 1e-4, 300 iterations) with the Lloyd iterations on
    the device: the data are centred and seeded exactly as ``KMeans.fit`` does - ``kmeans_plusplus``
    draws from the caller's RandomState, so the stream ends where the library's would - and
    ``Chain.init_kmeans_lloyd`` runs _kmeans_single_lloyd's loop.  Returns (labels, centers)."""
    if chain is None or not _sklearn_kmeans_restated():
        from sklearn.cluster import KMeans
        km = KMeans(n_clusters=n_clusters, random_state=random_state).fit(X_vec)
        return km.labels_, km.cluster_centers_
    from .lsm import check_random_state                    # sklearn.utils.check_random_state's rules
    rs = check_random_state(random_state)
    Xc = np.array(X_vec, dtype=np.float64, order='C')
    # KMeans._check_params_vs_input computes its absolute tolerance from the data BEFORE fit()
    # subtracts the mean (sklearn/cluster/_kmeans.py: _tolerance): same bits at the stopping test
    tol = np.mean(np.var(Xc, axis=0)) * 1e-4
    mean = Xc.mean(axis=0)
    Xc -= mean
    seeds = kmeans_plusplus_seeds(Xc, n_clusters, rs)
    res = chain.init_kmeans_lloyd(Xc, seeds, max_iter=300, tol=tol)
    if res is None:         # a cluster emptied: scikit-learn's relocation rule, on the host
        from sklearn.cluster._kmeans import _kmeans_single_lloyd
        labels, _, centers, _ = _kmeans_single_lloyd(Xc, np.ones(Xc.shape[0]), seeds.copy(),
                                                     max_iter=300, tol=tol)
    else:
        centers, labels, _ = res
    return labels, centers + mean


# scikit-learn releases whose KMeans internals (k-means++ draw order with `choice(p=w)` for the
# first centre, n_init='auto' = one start, _kmeans_single_lloyd's signature) the restatement above
# was pinned against (tests/golden/kmeans.npz and the k-means++ test beside it); any other release
# runs the library itself on the host
_KMEANS_RESTATED_FOR = ((1, 4), (1, 5), (1, 6), (1, 7))


def _sklearn_kmeans_restated():
    try:
        import sklearn
        ver = tuple(int(p) for p in sklearn.__version__.split('.')[:2])
    except Exception:       # noqa: BLE001
        return False
    return ver in _KMEANS_RESTATED_FOR


def longitudinal_kmeans(X, n_clusters=5, var_reg=1e-3, random_state=None, chain=None):
    """Genolini & Falissard (2010): k-means on the time-stacked trajectories;
    returns (centers[K, D], variances[K], labels[T, N]).  ``chain``: any chain handle of the
    device the Lloyd iterations should run on (None: scikit-learn on the host)."""
    T, N, D = X.shape
    X_vec = np.moveaxis(X, 0, -1).reshape(N, T * D)
    km_labels, km_centers = _kmeans(X_vec, n_clusters, random_state, chain)
    labels = np.tile(np.asarray(km_labels).reshape(1, -1), (T, 1))
    centers = np.empty((n_clusters, D))
    for k in range(n_clusters):
        centers[k] = km_centers[k].reshape(-1, T).T.mean(axis=0)
    variances = np.zeros(n_clusters)
    for k in range(n_clusters):
        for t in range(T):
            variances[k] += np.var(X[t][labels[t] == k], axis=0).mean()
        variances[k] /= T
    variances[variances == 0.] = var_reg
    return centers, variances, labels
