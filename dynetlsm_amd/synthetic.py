"""Synthetic dynamic networks for the benchmark configurations.

Own generator (numpy on the host): a Gaussian random walk of latent positions
and Bernoulli dyads with logit  b - ||X_ti - X_tj||  (the model of
DynamicNetworkLSM); the reference's generator is O(T N^2) Python-side.
"""
import numpy as np

__all__ = ['synthetic_lsm_network', 'synthetic_hdp_network', 'synthetic_sparse_directed',
           'synthetic_directed_from_model']


def _expit(x):
    return 1.0 / (1.0 + np.exp(-np.maximum(x, -700.0)))


def _pairwise(X):
    sq = (X * X).sum(1)
    d2 = sq[:, None] + sq[None, :] - 2.0 * X.dot(X.T)
    np.maximum(d2, 0.0, out=d2)
    return np.sqrt(d2)


def synthetic_lsm_network(T=10, N=2000, D=2, density=0.03, seed=0, directed=False,
                          x0_scale=1.5, walk_scale=0.3, init_noise=0.1):
    """Returns dict(Y, X_true, intercept, X_init).

    X[0] ~ N(0, x0_scale^2 I), X[t] = X[t-1] + N(0, walk_scale^2 I), centred;
    the intercept is solved so that the expected density at t = 0 is
    ``density``; Y_tij ~ Bernoulli(expit(b - d_tij)), symmetrised unless
    ``directed``, zero diagonal, float64 as fit(Y) expects.  ``X_init`` = truth +
    N(0, init_noise^2) is where the timing runs start the chain (the
    initialisation pipeline is outside the hot path)."""
    rng = np.random.RandomState(seed)
    X = np.zeros((T, N, D))
    X[0] = x0_scale * rng.randn(N, D)
    for t in range(1, T):
        X[t] = X[t - 1] + walk_scale * rng.randn(N, D)
    X -= X.mean(axis=(0, 1))
    d0 = _pairwise(X[0])
    iu = np.triu_indices(N, 1)
    lo, hi = -20.0, 20.0
    for _ in range(60):
        b = 0.5 * (lo + hi)
        if _expit(b - d0[iu]).mean() < density:
            lo = b
        else:
            hi = b
    b = 0.5 * (lo + hi)
    Y = np.zeros((T, N, N))
    for t in range(T):
        P = _expit(b - _pairwise(X[t]))
        U = rng.rand(N, N)
        A = (U < P).astype(np.float64)
        np.fill_diagonal(A, 0.0)
        if not directed:
            A = np.triu(A, 1)
            A = A + A.T
        Y[t] = A
    X_init = X + init_noise * rng.randn(T, N, D)
    return dict(Y=Y, X_true=X, intercept=float(b), X_init=X_init)


def synthetic_hdp_network(T=10, N=2000, D=2, n_clusters=6, density=0.03, lmbda=0.8,
                          stay=0.95, seed=0, center_scale=3.0, sigma=0.25, init_noise=0.1):
    """BASELINE.json configs[2]: a network with the HDP-LPCM's own
    dynamics.  Labels follow a sticky Markov chain (a node keeps its cluster with
    probability ``stay``), positions the blended AR(1) walk of the model,
    x_0 ~ N(mu_z, sigma^2 I), x_t ~ N((1 - lmbda) x_{t-1} + lmbda mu_z, sigma^2 I)
    (the dynamics of samples_generator.py:701-796 at N = 2000, vectorised), dyads
    Bernoulli(expit(b - d)) with b solved for the target density.

    Returns dict(Y, X_true, z_true, mu_true, sigma_true, intercept, X_init): the timing
    runs start at X_init = truth + noise with the true labels / means as the mixture's
    starting values (the initialisation pipeline is measured separately)."""
    rng = np.random.RandomState(seed)
    ang = 2.0 * np.pi * np.arange(n_clusters) / n_clusters
    mu = np.zeros((n_clusters, D))
    mu[:, 0] = center_scale * np.cos(ang)
    if D > 1:
        mu[:, 1] = center_scale * np.sin(ang)
    z = np.zeros((T, N), dtype=np.int64)
    z[0] = rng.randint(0, n_clusters, size=N)
    for t in range(1, T):
        move = rng.rand(N) >= stay
        z[t] = np.where(move, rng.randint(0, n_clusters, size=N), z[t - 1])
    X = np.zeros((T, N, D))
    X[0] = mu[z[0]] + sigma * rng.randn(N, D)
    for t in range(1, T):
        X[t] = (1 - lmbda) * X[t - 1] + lmbda * mu[z[t]] + sigma * rng.randn(N, D)
    shift = X.mean(axis=(0, 1))
    X -= shift
    mu = mu - shift
    d0 = _pairwise(X[0])[np.triu_indices(N, 1)]
    if d0.size > 400000:                      # the bisection needs the mean to 3 digits only
        d0 = d0[:: d0.size // 400000]
    lo, hi = -20.0, 20.0
    for _ in range(40):
        b = 0.5 * (lo + hi)
        if _expit(b - d0).mean() < density:
            lo = b
        else:
            hi = b
    b = 0.5 * (lo + hi)
    Y = np.zeros((T, N, N))
    for t in range(T):
        A = (rng.rand(N, N) < _expit(b - _pairwise(X[t]))).astype(np.float64)
        np.fill_diagonal(A, 0.0)
        A = np.triu(A, 1)
        Y[t] = A + A.T
    X_init = X + init_noise * rng.randn(T, N, D)
    return dict(Y=Y, X_true=X, z_true=z, mu_true=mu, sigma_true=np.full(n_clusters, sigma ** 2),
                intercept=float(b), X_init=X_init)


def synthetic_sparse_directed(T=5, N=10000, deg=20, seed=0):
    """BASELINE.json configs[3]: a sparse directed network given directly
    as the case-control sampler's tables - the dense T x N x N tensor (4 GB at T = 5,
    N = 10 000) is never formed.  Every node draws ``deg`` out-neighbours uniformly (duplicates
    and self-loops dropped: mean out-degree just under ``deg``); positions ~ N(0, 0.01^2 I) and
    radii ~ Dirichlet(10) as in the reference's directed generators
    (samples_generator.py:123-131, 249-253).

    Returns (X, radii, degree[T, N, 2], in_edges, out_edges) in the layouts of
    case_control_likelihood.py:45-68 (zero padded, sources in increasing order)."""
    rng = np.random.RandomState(seed)
    X = 0.01 * rng.randn(T, N, 2)
    radii = rng.dirichlet(np.ones(N) * 10)
    out = rng.randint(0, N, size=(T, N, deg))
    out_lists = [[np.setdiff1d(np.unique(out[t, i]), [i]) for i in range(N)] for t in range(T)]
    degree = np.zeros((T, N, 2), dtype=np.int64)
    for t in range(T):
        for i in range(N):
            degree[t, i, 1] = out_lists[t][i].size
            np.add.at(degree[t, :, 0], out_lists[t][i], 1)
    out_edges = np.zeros((T, N, degree[:, :, 1].max()), dtype=np.int64)
    in_edges = np.zeros((T, N, degree[:, :, 0].max()), dtype=np.int64)
    fill = np.zeros((T, N), dtype=np.int64)
    for t in range(T):
        for i in range(N):
            e = out_lists[t][i]
            out_edges[t, i, :e.size] = e
            in_edges[t, e, fill[t, e]] = i       # sources arrive in increasing i
            fill[t, e] += 1
    return X, radii, degree, in_edges, out_edges


def synthetic_directed_from_model(T=5, N=10000, mean_degree=20.0, seed=0, intercepts=(0.3, 0.7),
                                  walk=0.1, chunk=1000, use_torch=None):
    """A sparse DIRECTED network drawn from the model itself at config 4's size (the posterior
    tests' network: `synthetic_sparse_directed` above draws its edges uniformly, which no
    parameter of the model generates).  Radii ~ Dirichlet(10) (mean 1 / N), positions a Gaussian
    cloud whose width is solved for the target mean out-degree, x_t = x_{t-1} + walk * width *
    N(0, I) (the directed generators of samples_generator.py:123-131, 249-253 at N = 10 000:
    radii from a Dirichlet, intercepts (0.3, 0.7), positions at the scale of the radii), edges
    i -> j ~ Bernoulli(expit(b_in (1 - d_ij / r_j) + b_out (1 - d_ij / r_i)))
    (directed_likelihoods_fast.pyx:185-205).  The T x N x N tensor is never formed: rows are drawn
    `chunk` at a time - with torch on the GPU when one is there (``use_torch=None``: if available;
    5e8 Bernoulli draws take numpy a minute), so the network depends on where it was drawn; the
    parameters do not.

    Returns dict(X, radii, intercepts, degree[T, N, 2], in_edges, out_edges, width) with the
    tables in the layouts of case_control_likelihood.py:45-68 (zero padded, sources in increasing
    order)."""
    rng = np.random.RandomState(seed)
    b_in, b_out = float(intercepts[0]), float(intercepts[1])
    radii = rng.dirichlet(np.ones(N) * 10)
    Z0 = rng.randn(N, 2)
    steps = rng.randn(T - 1, N, 2) if T > 1 else np.zeros((0, N, 2))

    def rows_proba(Xt, lo, hi):
        d = np.sqrt(((Xt[lo:hi, None, :] - Xt[None, :, :]) ** 2).sum(axis=2))
        eta = b_in * (1.0 - d / radii[None, :]) + b_out * (1.0 - d / radii[lo:hi, None])
        p = _expit(eta)
        p[np.arange(hi - lo), np.arange(lo, hi)] = 0.0
        return p

    # width of the cloud: bisection on the expected out-degree of a sample of rows at t = 0
    sample = np.sort(rng.choice(N, size=min(N, 400), replace=False))
    lo_w, hi_w = 1e-6, 1.0
    for _ in range(40):
        w = np.sqrt(lo_w * hi_w)
        Xt = w * Z0
        d = np.sqrt(((Xt[sample, None, :] - Xt[None, :, :]) ** 2).sum(axis=2))
        eta = b_in * (1.0 - d / radii[None, :]) + b_out * (1.0 - d / radii[sample, None])
        deg = _expit(eta).sum(axis=1).mean() - _expit(b_in + b_out)
        if deg > mean_degree:
            lo_w = w
        else:
            hi_w = w
    width = float(np.sqrt(lo_w * hi_w))
    X = np.zeros((T, N, 2))
    X[0] = width * Z0
    for t in range(1, T):
        X[t] = X[t - 1] + walk * width * steps[t - 1]
    X -= X.mean(axis=(0, 1))
    src, dst = [], []
    tch = None
    if use_torch is None or use_torch:
        try:
            import torch as tch
            if not tch.cuda.is_available():
                if use_torch:
                    raise RuntimeError('use_torch=True without a GPU')
                tch = None
        except ImportError:
            if use_torch:
                raise
            tch = None
    if tch is not None:
        gen = tch.Generator(device='cuda'); gen.manual_seed(int(seed))
        r_d = tch.as_tensor(radii, device='cuda')
    for t in range(T):
        s_t, d_t = [], []
        if tch is not None:
            Xd = tch.as_tensor(X[t], device='cuda')
        for lo in range(0, N, chunk):
            hi = min(N, lo + chunk)
            if tch is not None:
                d = tch.cdist(Xd[lo:hi], Xd, compute_mode='donot_use_mm_for_euclid_dist')
                eta = b_in * (1.0 - d / r_d[None, :]) + b_out * (1.0 - d / r_d[lo:hi, None])
                hit = tch.rand(eta.shape, generator=gen, device='cuda', dtype=tch.float64) < tch.sigmoid(eta)
                hit[tch.arange(hi - lo), tch.arange(lo, hi)] = False
                ij = tch.nonzero(hit).cpu().numpy()
                i, j = ij[:, 0], ij[:, 1]
            else:
                hit = rng.rand(hi - lo, N) < rows_proba(X[t], lo, hi)
                i, j = np.nonzero(hit)
            s_t.append(i + lo); d_t.append(j)
        src.append(np.concatenate(s_t)); dst.append(np.concatenate(d_t))
    degree = np.zeros((T, N, 2), dtype=np.int64)
    for t in range(T):
        degree[t, :, 1] = np.bincount(src[t], minlength=N)
        degree[t, :, 0] = np.bincount(dst[t], minlength=N)
    out_edges = np.zeros((T, N, max(1, int(degree[:, :, 1].max()))), dtype=np.int64)
    in_edges = np.zeros((T, N, max(1, int(degree[:, :, 0].max()))), dtype=np.int64)
    for t in range(T):
        # rows arrive sorted by source, targets increasing inside a row
        start = np.concatenate([[0], np.cumsum(degree[t, :, 1])[:-1]])
        out_edges[t, src[t], np.arange(src[t].size) - start[src[t]]] = dst[t]
        order = np.lexsort((src[t], dst[t]))           # by target, sources increasing
        ds, ss = dst[t][order], src[t][order]
        start = np.concatenate([[0], np.cumsum(degree[t, :, 0])[:-1]])
        in_edges[t, ds, np.arange(ds.size) - start[ds]] = ss
    return dict(X=X, radii=radii, intercepts=np.array([b_in, b_out]), degree=degree,
                in_edges=in_edges, out_edges=out_edges, width=width)
