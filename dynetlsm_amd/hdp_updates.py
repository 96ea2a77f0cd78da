"""Host side of one DynamicNetworkHDPLPCM Gibbs iteration: the O(TN + TK^2)
conjugate / auxiliary-variable updates that follow the latent-position sweep and
the label block update (hdp_lpcm.py:880-1023) and the log-posterior trace
(hdp_lpcm.py:1188-1280).

These are not kernels ("host-side, restated in the build's own
Python"): small numpy updates driven by a numpy ``RandomState``.  They follow
the reference's draw order call for call, which tests/test_hdp_host_updates.py
pins against a trace recorded from the reference.  The one draw whose count grows
with N T (the table counts) is made by the engine's native host helper from the same
RandomState.
"""
import math

import numpy as np
from scipy.special import log_ndtr, ndtr, ndtri_exp
from scipy.stats import dirichlet

from . import _lib

SMALL_EPS = np.finfo('float64').tiny
_HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)

__all__ = ['HDPHyper', 'sample_tables', 'sample_mbar', 'sample_concentration_param',
           'sample_dirichlet', 'gibbs_updates', 'log_posterior_terms']


class HDPHyper(object):
    """Hyper-parameters that the loop itself resamples (hdp_lpcm.py:957-1023)
    plus the fixed hyper-priors set up before it (:760-793)."""

    def __init__(self, n_components, gamma=1.0, alpha_init=1.0, alpha=1.0, kappa=4.0,
                 mean_variance_prior=2.0, b=1.0, a=2.0, a0=None, b0=None, c0=None,
                 d0=None, lambda_prior=0.9, lambda_variance_prior=0.01,
                 gamma_prior_shape=1.0, gamma_prior_rate=0.1, alpha_init_shape=1.0,
                 alpha_init_rate=1.0, alpha_kappa_shape=5, alpha_kappa_rate=0.1):
        self.n_components = n_components
        self.gamma, self.alpha_init, self.alpha, self.kappa = gamma, alpha_init, alpha, kappa
        self.mean_variance_prior, self.b, self.a = mean_variance_prior, b, a
        self.a0, self.b0, self.c0, self.d0 = a0, b0, c0, d0
        self.lambda_prior, self.lambda_variance_prior = lambda_prior, lambda_variance_prior
        self.gamma_prior_shape, self.gamma_prior_rate = gamma_prior_shape, gamma_prior_rate
        self.alpha_init_shape, self.alpha_init_rate = alpha_init_shape, alpha_init_rate
        self.alpha_kappa_shape, self.alpha_kappa_rate = alpha_kappa_shape, alpha_kappa_rate


def sample_dirichlet(alphas, rng):
    """distributions.py:85-92"""
    if np.any(alphas <= 0.):
        alphas = np.clip(alphas, a_min=SMALL_EPS, a_max=None)
    return rng.dirichlet(alphas)


def dirichlet_logpdf(x, alphas):
    """distributions.py:95-100"""
    if np.any(alphas <= 0.):
        alphas = np.clip(alphas, a_min=SMALL_EPS, a_max=None)
    if np.any(x <= 0):
        x = np.clip(x, a_min=SMALL_EPS, a_max=None)
    return dirichlet.logpdf(x, alphas)


def _log_gauss_mass(a, b):
    """log of the standard normal mass of [a, b], computed in the left tail"""
    if b <= 0:
        la, lb = log_ndtr(a), log_ndtr(b)
        return lb + math.log1p(-math.exp(la - lb))
    if a > 0:
        return _log_gauss_mass(-b, -a)
    return math.log1p(-ndtr(a) - ndtr(-b))


def _truncnorm_bounds(mean, var, lower, upper):
    std = math.sqrt(float(np.ravel(var)[0]))
    mean = float(np.ravel(mean)[0])
    return mean, std, (lower - mean) / std, (upper - mean) / std


def truncated_normal(mean, var, rng, lower=0, upper=1):
    """distributions.py:68-73: ``truncnorm.rvs(a, b, size=1, loc, scale, random_state)`` is one
    ``uniform`` draw pushed through the quantile function; the same log-space formulas as
    scipy's, on scalars (its generic machinery costs more than the rest of the update)."""
    mean, std, a, b = _truncnorm_bounds(mean, var, lower, upper)
    q = float(rng.uniform(size=1)[0])
    lm = _log_gauss_mass(a, b)
    with np.errstate(divide='ignore'):
        if a < 0:
            x = ndtri_exp(np.logaddexp(log_ndtr(a), np.log(q) + lm))
        else:
            x = -ndtri_exp(np.logaddexp(log_ndtr(-b), np.log1p(-q) + lm))
    return np.array([x * std + mean])


def truncated_normal_logpdf(x, mean, var, lower=0, upper=1):
    """distributions.py:76-80"""
    mean, std, a, b = _truncnorm_bounds(mean, var, lower, upper)
    shape = np.shape(x)
    y = (float(np.ravel(x)[0]) - mean) / std
    if y < a or y > b:
        lp = -np.inf
    else:
        lp = -0.5 * y * y - _HALF_LOG_2PI - _log_gauss_mass(a, b) - math.log(std)
    return np.full(shape, lp) if shape else lp


def sample_tables(n, beta, alpha_init, alpha, kappa, rng):
    """sample_auxillary.py:6-28 : number of tables serving each dish.

    The reference draws ``rng.binomial(1, p / (p + arange(n_tjk)))`` cell by cell
    (about N T Bernoulli draws per iteration); ``dlsm_host_sample_tables`` makes the
    same draws, in the same order, from ``rng``'s own MT19937 state through numpy's
    bit-generator interface, without the T*K*K Python-level calls."""
    T, K, _ = n.shape
    n = np.ascontiguousarray(n, dtype=np.float64)
    beta = np.ascontiguousarray(beta, dtype=np.float64)
    m = np.zeros((T, K, K), dtype=np.int64)
    bitgen = rng._bit_generator
    with bitgen.lock:
        rc = _lib.load().dlsm_host_sample_tables(
            bitgen.ctypes.bit_generator, T, K, n.ctypes.data_as(_lib.c_double_p),
            beta.ctypes.data_as(_lib.c_double_p), float(np.ravel(alpha_init)[0]),
            float(np.ravel(alpha)[0]), float(np.ravel(kappa)[0]), m.ctypes.data_as(_lib.c_i64_p))
    if rc != 0:
        raise ValueError('p < 0, p > 1 or p contains NaNs')
    return m


def sample_mbar(m, beta, kappa, alpha, rng):
    """sample_auxillary.py:31-50 : override variables of the sticky HDP.  One
    ``rng.binomial`` call on the (t, j)-ordered arrays: the legacy sampler walks its
    arguments element by element, so the draws are those of the reference's loop."""
    T, K, _ = m.shape
    rho = kappa / (alpha + kappa)
    idx = np.arange(K)
    if T > 1:
        p = rho / (rho + beta * (1 - rho))
        w = rng.binomial(m[1:, idx, idx], np.broadcast_to(p, (T - 1, K))).astype(np.float64)
    else:
        w = np.zeros((0, K))
    m_bar = m[1:].sum(axis=0).astype(np.float64)
    m_bar[idx, idx] -= w.sum(axis=0)
    return np.sum(m_bar, axis=0) + m[0, 0], w


def sample_concentration_param(alpha, n_clusters, n_samples, prior_shape, prior_rate, rng):
    """sample_concentration.py:6-21 (Escobar and West, 1995)"""
    eta = rng.beta(alpha + 1, n_samples)
    m_shape = prior_shape + n_clusters - 1
    m_scale = prior_rate - np.log(eta)
    log_odds = (m_shape / m_scale) * (1 / n_samples)
    mix_indicator = rng.binomial(1, log_odds / (1 + log_odds))
    m_shape = m_shape + 1 if mix_indicator else m_shape
    return rng.gamma(shape=m_shape, scale=1. / m_scale)


class DeviceLabelSums(object):
    """The O(T N) sums of the conjugate updates, on the device:
    ``Chain.hdp_label_sums`` at the chain's current positions and labels."""

    def __init__(self, chain):
        self.chain = chain
        self.T, self.N, self.D = chain.T, chain.N, chain.D

    def mean(self, lmbda):
        return self.chain.hdp_label_sums(0, lmbda=lmbda)

    def residual(self, mu, lmbda):
        return self.chain.hdp_label_sums(1, mu=mu, lmbda=lmbda)

    def lam(self, mu, sigma):
        return self.chain.hdp_label_sums(2, mu=mu, sigma=sigma)

    def logp(self, mu, sigma, lmbda, weights, a, b):
        return self.chain.hdp_label_sums(3, mu=mu, sigma=sigma, lmbda=lmbda, w=weights,
                                         a=a, b=b)


def _cluster_updates(sums, nk, mu, sigma, lmbda, hp, rng):
    """Cluster means, variances, blending coefficient and the two variance hyper-parameters
    (hdp_lpcm.py:901-972 = lpcm.py:585-668), in the reference's draw order; mu and sigma are
    updated in place, hp.mean_variance_prior / hp.b are mutated; returns the new lmbda."""
    T, N, D = sums.T, sums.N, sums.D
    K = hp.n_components
    # cluster means (:901-921)
    lm = float(np.ravel(lmbda)[0])
    S = sums.mean(lm)                                              # (T, K, D)
    has = nk > 0                                                   # (T, K)
    wt = np.full(T, lm ** 2); wt[0] = 1.0                           # precision weights
    ws = np.full(T, lm); ws[0] = 1.0
    pk = 1 / hp.mean_variance_prior + ((wt[:, None] * nk * has).sum(axis=0) / sigma)
    mk = ((ws[:, None, None] * S * has[:, :, None]).sum(axis=0) / sigma[:, None])
    pk = 1 / pk
    mk = mk * pk[:, None]
    # multivariate_normal(mean, pk I) of the legacy sampler = mean + sqrt(pk) * standard
    # normals (its SVD of a scaled identity is the identity), drawn in (k, d) order
    mu[:] = mk + np.sqrt(pk)[:, None] * rng.standard_normal((K, D))
    # cluster variances (:924-938): squared residuals about the new means, by label
    Q = sums.residual(mu, lm)                                      # (T, K)
    ak = 0.5 * (nk.sum(axis=0) * D + hp.a)
    bk = 0.5 * hp.b + 0.5 * (Q * has).sum(axis=0)
    sigma[:] = 1. / rng.gamma(shape=ak, scale=1. / bk)
    # blending coefficient (:941-954)
    if T > 1:
        L = sums.lam(mu, sigma)                                    # (T, K, 2), row 0 unused
        ml = np.sum(L[1:, :, 0])
        sl = 1.0 / hp.lambda_variance_prior + np.sum(L[1:, :, 1])
    else:
        ml, sl = 0.0, 1.0 / hp.lambda_variance_prior
    sl = 1. / sl
    ml += hp.lambda_prior / hp.lambda_variance_prior
    ml *= sl
    lmbda = truncated_normal(mean=ml, var=sl, rng=rng)
    # hyper-parameters (:957-972)
    if hp.a0 is not None:
        b = 0.5 * hp.b0 + 0.5 * np.sum(mu * mu)
        a = 0.5 * (hp.a0 + K)
        hp.mean_variance_prior = 1 / rng.gamma(shape=a, scale=1. / b, size=1)
    if hp.c0 is not None:
        scale = 0.5 * hp.d0 + 0.5 * np.sum(1. / sigma)
        shape = 0.5 * (hp.c0 + K * hp.a)
        hp.b = rng.gamma(shape=shape, scale=1. / scale)
    return lmbda


def gibbs_updates(sums, n, nk, mu, sigma, beta, weights, lmbda, hp, rng):
    """hdp_lpcm.py:880-1023.  ``n`` (T,K,K), ``nk`` (T,K) are the label
    update's counts; mu, sigma, weights are updated in place; returns
    (beta, lmbda) and mutates ``hp`` (gamma, alpha_init, alpha, kappa,
    mean_variance_prior, b).  ``sums`` supplies the label-wise sums over the nodes
    (:901-954): ``DeviceLabelSums`` in the product; the tests inject the numpy
    restatement of the oracle to pin the draws on a CPU."""
    T, N, D = sums.T, sums.N, sums.D
    K = hp.n_components
    m = sample_tables(n, beta, hp.alpha_init, hp.alpha, hp.kappa, rng)
    m_bar, w = sample_mbar(m, beta, hp.kappa, hp.alpha, rng)
    # global transition distribution (:887)
    beta = rng.dirichlet((hp.gamma / K) + m_bar)
    # initial distribution (:890) and transition distributions (:894-898).  A legacy
    # ``dirichlet`` draw is ``standard_gamma`` per component (index order) times the
    # inverse of their running sum, so all (t, k) rows come from ONE standard_gamma
    # call on the row-major stack of their parameters.
    weights[0, 0] = sample_dirichlet(hp.alpha_init * beta + nk[0], rng)
    if T > 1:
        al = (hp.alpha * beta + hp.kappa * np.eye(K))[None, :, :] + n[1:]
        al = np.where(al <= 0., SMALL_EPS, al)
        g = rng.standard_gamma(al)
        weights[1:] = g * (1.0 / np.add.accumulate(g, axis=-1)[..., -1:])
    lmbda = _cluster_updates(sums, nk, mu, sigma, lmbda, hp, rng)
    # concentration parameters (:977-1023)
    hp.gamma = sample_concentration_param(hp.gamma, np.sum(m_bar > 0), np.sum(m_bar),
                                          hp.gamma_prior_shape, hp.gamma_prior_rate, rng)
    hp.alpha_init = sample_concentration_param(hp.alpha_init, np.sum(m[0, 0]), N,
                                               hp.alpha_init_shape, hp.alpha_init_rate, rng)
    alpha_kappa = hp.alpha + hp.kappa
    n_dot = np.sum(n[1:], axis=2)
    valid = n_dot > 0
    valid_n_dot = n_dot[valid]
    s = rng.binomial(1, p=(valid_n_dot / (valid_n_dot + alpha_kappa)))
    r = rng.beta(alpha_kappa + 1, valid_n_dot)
    shape = hp.alpha_kappa_shape + np.sum(m[1:], axis=2)[valid].sum() - np.sum(s)
    rate = hp.alpha_kappa_rate - np.sum(np.log(r))
    alpha_kappa = rng.gamma(shape=shape, scale=1. / rate)
    rho_a, rho_b = 8, 2
    n_success = np.sum(w)
    rho = rng.beta(a=rho_a + n_success, b=np.sum(m[1:]) - n_success + rho_b)
    hp.kappa = alpha_kappa * rho
    hp.alpha = alpha_kappa - hp.kappa
    return beta, lmbda


def _dirichlet_logpdf_rows(x, alphas):
    """row-wise Dirichlet log-density with the clipping of distributions.py:95-100"""
    from scipy.special import gammaln, xlogy
    alphas = np.where(alphas <= 0., SMALL_EPS, alphas)
    x = np.where(x <= 0, SMALL_EPS, x)
    return (gammaln(alphas.sum(axis=-1)) - gammaln(alphas).sum(axis=-1) +
            xlogy(alphas - 1, x).sum(axis=-1))


def log_posterior_terms(sums, intercept, intercept_prior, intercept_variance_prior, mu,
                        sigma, weights, beta, lmbda, hp, radii=None):
    """Everything in DynamicNetworkHDPLPCM.logp (hdp_lpcm.py:1188-1280) except
    the network log-likelihood, which the device supplies.  The terms that sum over the
    nodes (label transitions, Gaussian and inverse-gamma terms) come from ``sums``."""
    T, N, D = sums.T, sums.N, sums.D
    K = hp.n_components
    lp = _dirichlet_logpdf_rows(beta, np.repeat(hp.gamma / K, K))
    lp += _dirichlet_logpdf_rows(weights[0, 0], hp.alpha_init * beta)
    if T > 1:
        al = hp.alpha * beta[None, :] + hp.kappa * np.eye(K)          # (K, K)
        lp += _dirichlet_logpdf_rows(weights[1:], al[None, :, :]).sum()
    diff = intercept - intercept_prior
    if radii is not None:
        lp -= np.sum(0.5 * (diff * diff) / intercept_variance_prior)
    else:
        lp = lp - 0.5 * (diff * diff) / intercept_variance_prior
    lp = lp + np.sum(sums.logp(mu, sigma, np.ravel(lmbda)[0], weights, hp.a, hp.b))
    lp = lp - 0.5 * np.sum(mu * mu) / hp.mean_variance_prior
    lp = lp + truncated_normal_logpdf(lmbda, mean=hp.lambda_prior,
                                      var=hp.lambda_variance_prior)
    if radii is not None:
        lp = lp + dirichlet.logpdf(radii, np.ones(N))
    if hp.a0 is not None:
        lp = lp + (-(0.5 * hp.a0 + 1) * np.log(hp.mean_variance_prior) -
                   (0.5 * hp.b0 / hp.mean_variance_prior))
    if hp.c0 is not None:
        lp = lp + (hp.c0 - 1) * np.log(hp.b) - hp.d0 * hp.b
    return lp



# ---------------------------------------------------------------------------
# DynamicNetworkLPCM (finite mixture, time-homogeneous label chain; lpcm.py)
# ---------------------------------------------------------------------------
def lpcm_gibbs_updates(sums, n, nk, mu, sigma, init_weights, trans_weights, lmbda, hp, rng,
                       dirichlet_prior):
    """lpcm.py:572-668: initial / transition distributions, then the cluster updates shared
    with the HDP-LPCM.  init_weights (K,), trans_weights (K, K), mu, sigma are updated in
    place; returns lmbda."""
    K = hp.n_components
    init_weights[:] = sample_dirichlet(dirichlet_prior + nk[0], rng)
    # K legacy dirichlet draws = one standard_gamma call on the row-major stack
    al = dirichlet_prior + n[1:].sum(axis=0)
    al = np.where(al <= 0., SMALL_EPS, al)
    g = rng.standard_gamma(al)
    trans_weights[:] = g * (1.0 / np.add.accumulate(g, axis=-1)[..., -1:])
    return _cluster_updates(sums, nk, mu, sigma, lmbda, hp, rng)


def lpcm_log_posterior_terms(sums, intercept, intercept_prior, intercept_variance_prior, mu,
                             sigma, init_weights, trans_weights, lmbda, hp, dirichlet_prior,
                             radii=None):
    """DynamicNetworkLPCM.logp (lpcm.py:770-856) without the network log-likelihood."""
    T, N, D = sums.T, sums.N, sums.D
    K = hp.n_components
    prior = dirichlet_prior * np.ones(K)
    lp = _dirichlet_logpdf_rows(init_weights, prior)
    lp += _dirichlet_logpdf_rows(trans_weights, prior[None, :]).sum()
    diff = intercept - intercept_prior
    if radii is not None:
        lp -= np.sum(0.5 * (diff * diff) / intercept_variance_prior)
    else:
        lp = lp - 0.5 * (diff * diff) / intercept_variance_prior
    w = np.empty((T, K, K))
    w[:] = trans_weights[None]
    w[0, 0] = init_weights
    lp = lp + np.sum(sums.logp(mu, sigma, np.ravel(lmbda)[0], w, hp.a, hp.b))
    lp = lp - 0.5 * np.sum(mu * mu) / hp.mean_variance_prior
    lp = lp + truncated_normal_logpdf(lmbda, mean=hp.lambda_prior,
                                      var=hp.lambda_variance_prior)
    if radii is not None:
        lp = lp + dirichlet.logpdf(radii, np.ones(N))
    if hp.a0 is not None:
        lp = lp + (-(0.5 * hp.a0 + 1) * np.log(hp.mean_variance_prior) -
                   (0.5 * hp.b0 / hp.mean_variance_prior))
    if hp.c0 is not None:
        lp = lp + (hp.c0 - 1) * np.log(hp.b) - hp.d0 * hp.b
    return lp
