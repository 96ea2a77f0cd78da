"""Starting values for the chains (one-off, before the hot loop; host side).

SURVEY.md 8f ranks the initialisation pipeline as the first thing to move to the
device after the hot path; until then these are compact numpy / scipy / sklearn
restatements of what the reference computes with the same third-party libraries:
generalised MDS (latent_space.py:47-95), radii (:140-153), the conditional MLE
of scale / intercepts (lsm.py:32-97; each objective evaluation is one fused
log-likelihood pass on the GPU) and longitudinal k-means (latent_space.py:98-137).
"""
import numpy as np
from scipy.linalg import eigh, orthogonal_procrustes
from scipy.optimize import minimize
from scipy.sparse import csgraph

__all__ = ['generalized_mds', 'initialize_radii', 'scale_intercept_mle',
           'directed_intercept_mle', 'longitudinal_kmeans']


def _shortest_path_dissimilarity(Y):
    dist = csgraph.shortest_path(Y, directed=False, unweighted=True)
    inf = np.isinf(dist)
    dist[inf] = np.max(dist[~inf]) + 1      # unconnected: largest distance + 1
    return dist


def generalized_mds(Y, n_features=2, is_directed=False, lmbda=10, random_state=None):
    """Sarkar & Moore (2005) generalised MDS of a dynamic network."""
    from sklearn.manifold import MDS
    T, N, _ = Y.shape
    Dm = np.stack([_shortest_path_dissimilarity(Y[t]) for t in range(T)])
    X = np.empty((T, N, n_features))
    X[0] = MDS(dissimilarity='precomputed', n_components=n_features,
               random_state=random_state).fit_transform(Dm[0])
    H = np.eye(N) - np.ones((N, N)) / N
    alpha, beta = 1 / (1 + lmbda), lmbda / (1 + lmbda)
    for t in range(1, T):
        G = alpha * H.dot((-0.5 * Dm[t] ** 2).dot(H)) + beta * X[t - 1].dot(X[t - 1].T)
        evals, evecs = eigh(G)
        evals, evecs = evals[::-1], evecs[:, ::-1]
        X[t] = evecs[:, :n_features] * np.sqrt(evals[:n_features])
        R, _ = orthogonal_procrustes(X[t], X[t - 1])
        X[t] = X[t].dot(R)
    if is_directed:
        X /= N           # same scale as the radii
    return X


def initialize_radii(Y, reg=1e-5):
    radii = 0.5 * (Y.sum(axis=(0, 1)) + Y.sum(axis=(0, 2)))
    radii /= Y.sum()
    if np.any(radii == 0.):
        radii += reg
        radii /= np.sum(radii)
    return radii


def _pairwise(X):
    sq = (X * X).sum(-1)
    d2 = sq[:, :, None] + sq[:, None, :] - 2 * np.einsum('tid,tjd->tij', X, X)
    return np.sqrt(np.maximum(d2, 0.0))


def scale_intercept_mle(chain, Y, X, tol=1e-4):
    """Conditional MLE of (log scale, intercept) of the undirected model by BFGS
    (lsm.py:47-70).  The objective is evaluated on the GPU (one fused pass per
    call); the gradient is the closed form of lsm.py:32-44 in numpy."""
    T, N, _ = X.shape
    iu = np.triu_indices(N, 1)
    dist = np.stack([_pairwise(X[t:t + 1])[0][iu] for t in range(T)])
    y = np.stack([Y[t][iu] for t in range(T)])

    def fun(x):
        chain.set_positions(X * np.exp(x[0]))
        return -chain.loglik_full([[x[1]]])[0]

    def grad(x):
        sd = np.exp(x[0]) * dist
        eta = x[1] - sd
        p = 1 / (1 + np.exp(-eta))
        return -np.array([np.sum(-sd * (y - p)) * 2, np.sum(y - p)])

    res = minimize(fun=fun, x0=np.array([0.0, 1.0]), method='BFGS', jac=grad, tol=tol)
    return res.x[0], res.x[1]


def directed_intercept_mle(chain, Y, X, radii, tol=1e-4):
    """Conditional MLE of (intercept_in, intercept_out) (lsm.py:73-97)."""
    T, N, _ = X.shape
    dist = _pairwise(X)
    off = ~np.eye(N, dtype=bool)
    d_in = (1 - dist / radii[None, None, :])[:, off]
    d_out = (1 - dist / radii[None, :, None])[:, off]
    y = Y[:, off]
    chain.set_positions(X)
    chain.set_radii(radii)

    def fun(x):
        return -chain.loglik_full([[x[0], x[1]]])[0]

    def grad(x):
        eta = x[0] * d_in + x[1] * d_out
        step = y - 1 / (1 + np.exp(-eta))
        return -np.array([np.sum(d_in * step), np.sum(d_out * step)])

    res = minimize(fun=fun, x0=np.array([0.0, 0.0]), method='BFGS', jac=grad, tol=tol)
    return res.x[0], res.x[1]


def longitudinal_kmeans(X, n_clusters=5, var_reg=1e-3, random_state=None):
    """Genolini & Falissard (2010): k-means on the time-stacked trajectories;
    returns (centers[K, D], variances[K], labels[T, N])."""
    from sklearn.cluster import KMeans
    T, N, D = X.shape
    X_vec = np.moveaxis(X, 0, -1).reshape(N, T * D)
    km = KMeans(n_clusters=n_clusters, random_state=random_state).fit(X_vec)
    labels = np.tile(km.labels_.reshape(1, -1), (T, 1))
    centers = np.empty((n_clusters, D))
    for k in range(n_clusters):
        centers[k] = km.cluster_centers_[k].reshape(-1, T).T.mean(axis=0)
    variances = np.zeros(n_clusters)
    for k in range(n_clusters):
        for t in range(T):
            variances[k] += np.var(X[t][labels[t] == k], axis=0).mean()
        variances[k] /= T
    variances[variances == 0.] = var_reg
    return centers, variances, labels
