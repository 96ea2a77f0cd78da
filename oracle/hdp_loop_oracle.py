"""CPU restatement of one DynamicNetworkHDPLPCM Gibbs iteration after the latent-position
sweep (hdp_lpcm.py:876-1023: auxiliary tables, override variables, global / initial /
transition distributions, cluster means and variances, blending coefficient,
hyper-parameters, concentration parameters) and of its log-posterior
(hdp_lpcm.py:1188-1280), plus the whole iteration in the engine's order.

TEST INFRASTRUCTURE ONLY (tests/, bench.py's cpu_baseline leg, __graft_entry__.smoke()):
the product makes these draws on the device (dlsm_hdp_run) or, for the bit-level pin, in
dynetlsm_amd/hdp_updates.py on the caller's MT19937 stream.

The update code below is written once, against a small `draws` interface, in the
reference's statement order:

  * ``MTDraws(rng)`` maps every call onto the numpy / scipy call the reference makes at that
    point, so the code consumes a RandomState exactly as hdp_lpcm.py does -
    tests/test_hdp_loop_oracle.py replays the 9-iteration trace recorded from the reference's
    own ``_fit`` (tests/golden/hdp_trace.npz) through it: PARITY PINNED;
  * ``PhiloxDraws(seed, chain, it)`` makes the engine's counter-based draws (Philox4x32-10,
    stream HDP = 6, counter (index, kind | attempt << 8, iteration)): the same code is then
    what the device loop must reproduce value for value.
"""
import math

import numpy as np
from scipy.special import gammaln, log_ndtr, ndtr, ndtri_exp, xlogy
from scipy.stats import truncnorm

from . import oracle as orc

SMALL_EPS = np.finfo('float64').tiny
STREAM_HDP = 6
# draw kinds (low byte of counter word 1), shared with csrc/kernels_hdploop.hpp
(K_TABLES, K_OVERRIDE, K_BETA, K_W0, K_W, K_MU, K_SIGMA, K_LAMBDA, K_MVP, K_B,
 K_CONC_GAMMA, K_CONC_ALPHA0, K_AK_S, K_AK_R, K_AK, K_RHO) = range(16)


# ------------------------------------------------------------------------------------
# draws
# ------------------------------------------------------------------------------------
class MTDraws(object):
    """every call = the reference's own numpy / scipy call at that statement"""

    def __init__(self, rng):
        self.rng = rng

    def tables_cell(self, cell, p, n):                    # sample_auxillary.py:17-18,26-28
        return int(np.sum(self.rng.binomial(1, p / (p + np.arange(n)))))

    def binomial(self, kind, idx, n, p):                  # sample_auxillary.py:41-42
        return self.rng.binomial(n, p)

    def dirichlet(self, kind, idx0, alphas):              # hdp_lpcm.py:887,890,897
        return self.rng.dirichlet(alphas)

    def mvn_iso(self, kind, idx, mean, var):              # hdp_lpcm.py:920-921
        return self.rng.multivariate_normal(mean=mean, cov=var * np.eye(mean.shape[0]))

    def gamma(self, kind, idx, shape, scale):             # hdp_lpcm.py:938,966,972 ...
        return self.rng.gamma(shape=shape, scale=scale)

    def gamma_size1(self, kind, idx, shape, scale):       # hdp_lpcm.py:964-966 (size=1)
        return self.rng.gamma(shape=shape, scale=scale, size=1)

    def truncnorm(self, kind, mean, var):                 # distributions.py:68-73
        std = np.sqrt(var)
        a, b = (0 - mean) / std, (1 - mean) / std
        return truncnorm.rvs(a, b, size=1, loc=mean, scale=std, random_state=self.rng)

    def beta(self, kind, idx, a, b):                      # sample_concentration.py:11
        return self.rng.beta(a, b)

    def bernoulli(self, kind, idx, p):                    # sample_concentration.py:17
        return self.rng.binomial(1, p)

    def bernoulli_grid(self, kind, valid, p):             # hdp_lpcm.py:1005-1006
        return self.rng.binomial(1, p=p)

    def beta_grid(self, kind, valid, a, b):               # hdp_lpcm.py:1007
        return self.rng.beta(a, b)


def _log_gauss_mass(a, b):
    """log of the standard normal mass of [a, b] (scipy's formulas, scalars)"""
    if b <= 0:
        la, lb = log_ndtr(a), log_ndtr(b)
        return lb + math.log1p(-math.exp(la - lb))
    if a > 0:
        return _log_gauss_mass(-b, -a)
    return math.log1p(-ndtr(a) - ndtr(-b))


def truncnorm_quantile(q, mean, var, lower=0.0, upper=1.0):
    """the q-quantile of N(mean, var) truncated to [lower, upper] in log space (what
    scipy.stats.truncnorm.ppf computes)"""
    std = math.sqrt(var)
    a, b = (lower - mean) / std, (upper - mean) / std
    lm = _log_gauss_mass(a, b)
    with np.errstate(divide='ignore'):
        if a < 0:
            x = ndtri_exp(np.logaddexp(log_ndtr(a), np.log(q) + lm))
        else:
            x = -ndtri_exp(np.logaddexp(log_ndtr(-b), np.log1p(-q) + lm))
    return float(x) * std + mean


def truncnorm_logpdf(x, mean, var, lower=0.0, upper=1.0):
    std = math.sqrt(var)
    a, b = (lower - mean) / std, (upper - mean) / std
    y = (x - mean) / std
    if y < a or y > b:
        return -np.inf
    return -0.5 * y * y - 0.5 * math.log(2.0 * math.pi) - _log_gauss_mass(a, b) - math.log(std)


class PhiloxDraws(object):
    """the engine's draws: Philox stream HDP, counter (index, kind | attempt << 8, iteration)"""

    def __init__(self, seed, chain, it):
        self.seed, self.chain, self.it = int(seed), int(chain), int(it)
        self.sw = orc.stream_word(self.chain, STREAM_HDP)

    def u2(self, kind, idx, att):
        u0, u1 = orc.philox_uniform2(self.seed, idx, kind | (att << 8), self.it, self.sw)
        return float(u0), float(u1)

    def u2_vec(self, kind, idx, att):
        return orc.philox_uniform2(self.seed, idx, np.asarray(kind) | (np.asarray(att) << 8),
                                   self.it, self.sw)

    def std_gamma(self, kind, idx, a):
        """Gamma(a, 1), Marsaglia & Tsang (2000); a < 1 through Gamma(a + 1) U^(1/a).  Attempt
        k draws its normal from (idx, kind | 2k << 8) and its uniforms from
        (idx, kind | (2k + 1) << 8)."""
        aa = a + 1.0 if a < 1.0 else a
        d = aa - 1.0 / 3.0
        cc = 1.0 / math.sqrt(9.0 * d)
        for att in range(4096):
            u0, u1 = self.u2(kind, idx, 2 * att)
            z0 = math.sqrt(-2.0 * math.log(u0)) * math.cos(6.283185307179586476925286766559 * u1)
            w0, w1 = self.u2(kind, idx, 2 * att + 1)
            t = 1.0 + cc * z0
            if t <= 0.0:
                continue
            v = t * t * t
            x2 = z0 * z0
            if w0 < 1.0 - 0.0331 * x2 * x2 or math.log(w0) < 0.5 * x2 + d * (1.0 - v + math.log(v)):
                out = d * v
                if a < 1.0:
                    out *= w1 ** (1.0 / a)
                return out
        return 0.0

    def _count(self, kind, idx, probs):
        """number of successes among trials i = 0 .. len(probs) - 1; trial i is the uniform
        i & 1 of attempt i >> 1, success iff u <= p_i"""
        n = len(probs)
        if n == 0:
            return 0
        att = np.arange((n + 1) // 2)
        u0, u1 = self.u2_vec(kind, idx, att)
        u = np.empty(2 * att.size)
        u[0::2], u[1::2] = u0, u1
        return int(np.sum(u[:n] <= np.asarray(probs, dtype=np.float64)))

    def tables_cell(self, cell, p, n):
        return self._count(K_TABLES, cell, p / (p + np.arange(n)))

    def binomial(self, kind, idx, n, p):
        return self._count(kind, idx, np.full(int(n), p))

    def dirichlet(self, kind, idx0, alphas):
        g = np.array([self.std_gamma(kind, idx0 + k, a) for k, a in enumerate(alphas)])
        return g * (1.0 / g.sum())

    def mvn_iso(self, kind, idx, mean, var):
        D = mean.shape[0]
        z = np.zeros(D)
        for d in range(D):
            u0, u1 = self.u2(kind, idx, d >> 1)
            r = math.sqrt(-2.0 * math.log(u0))
            ang = 6.283185307179586476925286766559 * u1
            z[d] = r * (math.sin(ang) if d & 1 else math.cos(ang))
        return mean + math.sqrt(var) * z

    def gamma(self, kind, idx, shape, scale):
        return self.std_gamma(kind, idx, float(shape)) * float(scale)

    def gamma_size1(self, kind, idx, shape, scale):
        return np.array([self.gamma(kind, idx, shape, np.ravel(scale)[0])])

    def truncnorm(self, kind, mean, var):
        q, _ = self.u2(kind, 0, 0)
        return np.array([truncnorm_quantile(q, float(np.ravel(mean)[0]), float(np.ravel(var)[0]))])

    def beta(self, kind, idx, a, b):
        ga = self.std_gamma(kind, 2 * idx, float(a))
        gb = self.std_gamma(kind, 2 * idx + 1, float(b))
        return ga / (ga + gb)

    def bernoulli(self, kind, idx, p):
        u0, _ = self.u2(kind, idx, 0)
        return int(u0 <= p)

    def bernoulli_grid(self, kind, valid, p):
        """one Bernoulli per valid cell of the (T - 1) x K grid, indexed by the cell's
        row-major position in the FULL grid"""
        pos = np.flatnonzero(np.ravel(valid))
        return np.array([self.bernoulli(kind, int(i), float(pp)) for i, pp in zip(pos, p)],
                        dtype=np.int64)

    def beta_grid(self, kind, valid, a, b):
        pos = np.flatnonzero(np.ravel(valid))
        return np.array([self.beta(kind, int(i), a, float(bb)) for i, bb in zip(pos, b)])


# ------------------------------------------------------------------------------------
# the updates, in the reference's statement order
# ------------------------------------------------------------------------------------
class Hyper(object):
    """hyper-parameters the loop resamples + the fixed hyper-priors (hdp_lpcm.py:760-793)"""

    def __init__(self, **kw):
        self.gamma = kw.get('gamma', 1.0)
        self.alpha_init = kw.get('alpha_init', 1.0)
        self.alpha = kw.get('alpha', 1.0)
        self.kappa = kw.get('kappa', 4.0)
        self.mean_variance_prior = kw.get('mean_variance_prior', 2.0)
        self.b = kw.get('b', 1.0)
        self.a = kw.get('a', 2.0)
        self.a0, self.b0 = kw.get('a0'), kw.get('b0')
        self.c0, self.d0 = kw.get('c0'), kw.get('d0')
        self.lambda_prior = kw.get('lambda_prior', 0.9)
        self.lambda_variance_prior = kw.get('lambda_variance_prior', 0.01)
        self.gamma_prior_shape = kw.get('gamma_prior_shape', 1.0)
        self.gamma_prior_rate = kw.get('gamma_prior_rate', 0.1)
        self.alpha_init_shape = kw.get('alpha_init_shape', 1.0)
        self.alpha_init_rate = kw.get('alpha_init_rate', 1.0)
        self.alpha_kappa_shape = kw.get('alpha_kappa_shape', 5)
        self.alpha_kappa_rate = kw.get('alpha_kappa_rate', 0.1)

    def copy(self):
        h = Hyper()
        h.__dict__.update(self.__dict__)
        return h


def _clip_alphas(alphas):
    """distributions.py:85-92"""
    if np.any(alphas <= 0.):
        alphas = np.clip(alphas, a_min=SMALL_EPS, a_max=None)
    return alphas


def sample_tables(n, beta, alpha_init, alpha, kappa, draws):
    """sample_auxillary.py:6-28"""
    T, K, _ = n.shape
    m = np.zeros((T, K, K), dtype=np.int64)
    probas = alpha_init * beta
    for k in range(K):
        m[0, 0, k] = draws.tables_cell(k, probas[k], int(n[0, 0, k]))
    probas = alpha * beta + kappa * np.eye(K)
    for t in range(1, T):
        for j in range(K):
            for k in range(K):
                m[t, j, k] = draws.tables_cell((t * K + j) * K + k, probas[j, k], int(n[t, j, k]))
    return m


def sample_mbar(m, beta, kappa, alpha, draws):
    """sample_auxillary.py:31-50"""
    T, K, _ = m.shape
    w = np.zeros((T - 1, K), dtype=np.float64)
    rho = kappa / (alpha + kappa)
    for t in range(T - 1):
        for j in range(K):
            w[t, j] = draws.binomial(K_OVERRIDE, t * K + j, m[t + 1, j, j],
                                     rho / (rho + beta[j] * (1 - rho)))
    m_bar = np.zeros((T - 1, K, K), dtype=np.float64)
    for t in range(T - 1):
        m_bar[t] = m[t + 1] - np.diag(w[t])
    return np.sum(m_bar, axis=(0, 1)) + m[0, 0], w


def sample_concentration_param(alpha, n_clusters, n_samples, prior_shape, prior_rate, draws, kind):
    """sample_concentration.py:6-21 (Escobar and West, 1995)"""
    eta = draws.beta(kind, 0, alpha + 1, n_samples)
    m_shape = prior_shape + n_clusters - 1
    m_scale = prior_rate - np.log(eta)
    log_odds = (m_shape / m_scale) * (1 / n_samples)
    mix_indicator = draws.bernoulli(kind, 2, log_odds / (1 + log_odds))
    m_shape = m_shape + 1 if mix_indicator else m_shape
    return draws.gamma(kind, 3, m_shape, 1. / m_scale)


def gibbs_updates(X, z, n, nk, mu, sigma, beta, weights, lmbda, hp, draws):
    """hdp_lpcm.py:880-1023 after the label update: mu, sigma, weights are updated in place,
    hp is mutated; returns (beta, lmbda, aux) with aux = dict(m=, m_bar=, w=)."""
    T, N, D = X.shape
    K = sigma.shape[0]
    m = sample_tables(n, beta, hp.alpha_init, hp.alpha, hp.kappa, draws)
    m_bar, w = sample_mbar(m, beta, hp.kappa, hp.alpha, draws)
    # global transition distribution (:887)
    beta = draws.dirichlet(K_BETA, 0, (hp.gamma / K) + m_bar)
    # initial distribution (:890)
    weights[0, 0] = draws.dirichlet(K_W0, 0, _clip_alphas(hp.alpha_init * beta + nk[0]))
    # transition distributions (:894-898)
    probas = hp.alpha * beta + hp.kappa * np.eye(K)
    for t in range(1, T):
        for k in range(K):
            weights[t, k] = draws.dirichlet(K_W, (t * K + k) * K, _clip_alphas(probas[k] + n[t, k]))
    lm = float(np.ravel(lmbda)[0])
    # cluster means (:901-921)
    for k in range(K):
        pk = 1 / float(np.ravel(hp.mean_variance_prior)[0])
        mk = np.zeros(D)
        for t in range(T):
            if nk[t, k] > 0:
                mask = z[t] == k
                if t == 0:
                    pk += nk[0, k] / sigma[k]
                    mk += (1 / sigma[k]) * np.sum(X[t, mask], axis=0)
                else:
                    pk += (lm ** 2 / sigma[k]) * nk[t, k]
                    mk += (lm / sigma[k]) * np.sum(X[t, mask] - (1 - lm) * X[t - 1, mask], axis=0)
        pk = 1 / pk
        mk *= pk
        mu[k] = draws.mvn_iso(K_MU, k, mk, pk)
    # cluster variances (:924-938)
    for k in range(K):
        ak = 0.5 * (np.sum(nk[:, k]) * D + hp.a)
        bk = 0.5 * hp.b
        for t in range(T):
            if nk[t, k] > 0:
                mask = z[t] == k
                if t == 0:
                    bk += 0.5 * np.sum((X[t, mask] - mu[k]) ** 2)
                else:
                    bk += 0.5 * np.sum((X[t, mask] - (1 - lm) * X[t - 1, mask] - lm * mu[k]) ** 2)
        sigma[k] = 1. / draws.gamma(K_SIGMA, k, ak, 1. / bk)
    # blending coefficient (:941-954)
    ml = 0.0
    sl = 1.0 / hp.lambda_variance_prior
    for t in range(1, T):
        ml_diff = (mu[z[t]] - X[t - 1]) / sigma[z[t]].reshape(-1, 1)
        X_diff = X[t] - X[t - 1]
        ml += np.sum(ml_diff * X_diff)
        ml_diff = (mu[z[t]] - X[t - 1]) / np.sqrt(sigma[z[t]].reshape(-1, 1))
        sl += np.sum(ml_diff ** 2)
    sl = 1. / sl
    ml += hp.lambda_prior / hp.lambda_variance_prior
    ml *= sl
    lmbda = draws.truncnorm(K_LAMBDA, ml, sl)
    # hyper-parameters (:957-972)
    if hp.a0 is not None:
        b = 0.5 * hp.b0
        for k in range(K):
            b += 0.5 * np.sum(mu[k] ** 2)
        a = 0.5 * (hp.a0 + K)
        hp.mean_variance_prior = 1 / draws.gamma_size1(K_MVP, 0, a, 1. / b)
    if hp.c0 is not None:
        scale = 0.5 * hp.d0
        for k in range(K):
            scale += 0.5 * (1. / sigma[k])
        shape = 0.5 * (hp.c0 + K * hp.a)
        hp.b = draws.gamma(K_B, 0, shape, 1. / scale)
    # concentration parameters (:977-1023)
    hp.gamma = sample_concentration_param(hp.gamma, np.sum(m_bar > 0), np.sum(m_bar),
                                          hp.gamma_prior_shape, hp.gamma_prior_rate, draws,
                                          K_CONC_GAMMA)
    hp.alpha_init = sample_concentration_param(hp.alpha_init, np.sum(m[0, 0]), N,
                                               hp.alpha_init_shape, hp.alpha_init_rate, draws,
                                               K_CONC_ALPHA0)
    alpha_kappa = hp.alpha + hp.kappa
    n_dot = np.sum(n[1:], axis=2)
    valid = n_dot > 0
    valid_n_dot = n_dot[valid]
    s = draws.bernoulli_grid(K_AK_S, valid, valid_n_dot / (valid_n_dot + alpha_kappa))
    r = draws.beta_grid(K_AK_R, valid, alpha_kappa + 1, valid_n_dot)
    shape = hp.alpha_kappa_shape + np.sum(m[1:], axis=2)[valid].sum() - np.sum(s)
    rate = hp.alpha_kappa_rate - np.sum(np.log(r))
    alpha_kappa = draws.gamma(K_AK, 0, shape, 1. / rate)
    rho_a, rho_b = 8, 2
    n_success = np.sum(w)
    rho = draws.beta(K_RHO, 0, rho_a + n_success, np.sum(m[1:]) - n_success + rho_b)
    hp.kappa = alpha_kappa * rho
    hp.alpha = alpha_kappa - hp.kappa
    return beta, lmbda, dict(m=m, m_bar=m_bar, w=w)


def _dirichlet_logpdf(x, alphas):
    """distributions.py:95-100 around scipy.stats.dirichlet.logpdf, restated"""
    alphas = np.where(alphas <= 0., SMALL_EPS, alphas)
    x = np.where(x <= 0, SMALL_EPS, x)
    return gammaln(alphas.sum()) - gammaln(alphas).sum() + xlogy(alphas - 1, x).sum()


def log_posterior(loglik, X, intercept, mu, sigma, z, weights, beta, lmbda, hp, intercept_prior,
                  intercept_variance_prior, n_radii=None):
    """DynamicNetworkHDPLPCM.logp (hdp_lpcm.py:1188-1280); ``loglik`` is the network
    log-likelihood at (X, intercept[, radii]); ``n_radii`` = N for the directed models"""
    T, N, D = X.shape
    K = sigma.shape[0]
    lm = float(np.ravel(lmbda)[0])
    lp = _dirichlet_logpdf(beta, np.repeat(hp.gamma / K, K))
    lp += _dirichlet_logpdf(weights[0, 0], hp.alpha_init * beta)
    deltas = hp.kappa * np.eye(K)
    for t in range(1, T):
        for k in range(K):
            lp += _dirichlet_logpdf(weights[t, k], hp.alpha * beta + deltas[k])
    with np.errstate(divide='ignore'):
        lp += np.sum(np.log(weights[0, 0, z[0]]))
        for t in range(1, T):
            lp += np.sum(np.log(weights[t, z[t - 1], z[t]]))
    lp += loglik
    if n_radii is not None:             # directed models: both intercepts (hdp_lpcm.py:1234-1237)
        diff = np.ravel(intercept)[:2] - np.ravel(intercept_prior)[:2]
        lp -= np.sum(0.5 * (diff * diff) / intercept_variance_prior)
        lp += gammaln(float(n_radii))   # stats.dirichlet.logpdf(radii, ones(N)), hdp_lpcm.py:1268-1269
    else:
        diff = np.ravel(intercept)[0] - np.ravel(intercept_prior)[0]
        lp -= 0.5 * (diff * diff) / intercept_variance_prior
    for t in range(T):
        if t == 0:
            df = X[t] - mu[z[t]]
        else:
            df = X[t] - (1 - lm) * X[t - 1] - lm * mu[z[t]]
        lp += np.sum(-0.5 * np.log(sigma[z[t]]) - 0.5 * np.sum(df * df, axis=1) / sigma[z[t]])
    mvp = float(np.ravel(hp.mean_variance_prior)[0])
    for k in range(K):
        lp -= 0.5 * np.sum(mu[k] ** 2) / mvp
    lp += np.sum(-(0.5 * hp.a + 1) * np.log(sigma[z]) - (0.5 * hp.b / sigma[z]))
    lp += truncnorm_logpdf(lm, hp.lambda_prior, hp.lambda_variance_prior)
    if hp.a0 is not None:
        lp += -(0.5 * hp.a0 + 1) * np.log(mvp) - (0.5 * hp.b0 / mvp)
    if hp.c0 is not None:
        lp += (hp.c0 - 1) * np.log(hp.b) - hp.d0 * hp.b
    return float(lp)


# ------------------------------------------------------------------------------------
# the whole iteration in the engine's order and with the engine's draws
# ------------------------------------------------------------------------------------
class HdpChain(object):
    """State of one undirected HDP-LPCM chain for the Philox-order iteration: the C oracle's
    ChainState (sweep with the AR-mixture prior, Philox sweep draws) + the mixture's
    parameters + the intercept sampler."""

    def __init__(self, Y, X, intercept, mu, sigma, z, beta, weights, lmbda, hp, grid,
                 intercept_prior, intercept_variance_prior, isamp, seed, chain):
        self.Y = np.ascontiguousarray(Y, dtype=np.float64)
        self.st = orc.ChainState(X, grid, Y=self.Y, intercept=np.ravel(intercept), mu=mu,
                                 sigma=sigma, lmbda=lmbda, z=z, seed=seed, chain=chain)
        self.beta = np.array(beta, dtype=np.float64)
        self.weights = np.array(weights, dtype=np.float64)
        self.lmbda = np.array(np.ravel(lmbda)[:1], dtype=np.float64)
        self.hp = hp
        self.ip, self.var = float(np.ravel(intercept_prior)[0]), float(intercept_variance_prior)
        self.isamp = isamp                       # orc.ScalarMetropolis
        self.seed, self.chain = seed, chain
        self.aux = None

    @property
    def X(self):
        return self.st.X

    @property
    def mu(self):
        return self.st.mu

    @property
    def sigma(self):
        return self.st.sigma

    @property
    def z(self):
        return self.st.z

    @property
    def intercept(self):
        return self.st.intercept[:1]

    def loglik(self, b=None):
        b = self.st.c.intercept[0] if b is None else b
        return orc.dynamic_network_loglikelihood_undirected(self.Y, self.st.X, b)

    def iteration(self, it):
        """one Gibbs iteration in the engine's order; returns the log-posterior trace value"""
        st = self.st
        st.c.iter = it
        st.c.lmbda = float(self.lmbda[0])
        st.sweep_c()
        st.X[:] = orc.center(st.X)
        # intercept RW-MH (sample_coefficients.py:76-86) with the engine's draws
        sw = orc.stream_word(self.chain, orc.STREAM_INTERCEPT)
        u0, u1 = orc.philox_uniform2(self.seed, 0, 0, it, sw)
        z0 = math.sqrt(-2.0 * math.log(float(u0))) * math.cos(6.283185307179586476925286766559 * float(u1))
        b0 = st.c.intercept[0]
        b1 = b0 + self.isamp.step_size * z0
        ll0, ll1 = self.loglik(b0), self.loglik(b1)
        lp0 = ll0 - (b0 - self.ip) ** 2 / (2 * self.var)
        lp1 = ll1 - (b1 - self.ip) ** 2 / (2 * self.var)
        lu, _ = orc.philox_uniform2(self.seed, 0, 1, it, sw)
        accepted = int(not (math.log(float(lu)) >= lp1 - lp0))
        ll = ll1 if accepted else ll0
        st.c.intercept[0] = b1 if accepted else b0
        self.isamp.book(accepted)
        # label block update (sample_labels.py:134-190), engine draws, in C
        z, n, nk = orc.sample_labels_block_philox(st.X, st.mu, st.sigma, self.lmbda, self.weights,
                                                  self.seed, self.chain, it)
        st.z[:] = z
        self.beta, self.lmbda, self.aux = gibbs_updates(
            st.X, st.z, n, nk, st.mu, st.sigma, self.beta, self.weights, self.lmbda, self.hp,
            PhiloxDraws(self.seed, self.chain, it))
        self.lmbda = np.array(np.ravel(self.lmbda)[:1], dtype=np.float64)
        self.n, self.nk = n, nk
        return log_posterior(ll, st.X, st.c.intercept[0], st.mu, st.sigma, st.z, self.weights,
                             self.beta, self.lmbda, self.hp, self.ip, self.var)


class HdpChainDirected(HdpChain):
    """The directed (exact or case-control) HDP-LPCM chain in the engine's order
    (hdp_lpcm.py:823-1069 with is_directed): sweep with the directed partial likelihoods, centring,
    the two intercept steps and the radii step (oracle.directed_coefficient_steps = the steps of
    the directed LSM loop), then the label update and the conjugate draws of the undirected chain."""

    def __init__(self, Y, X, intercept, radii, mu, sigma, z, beta, weights, lmbda, hp, grid,
                 intercept_prior, intercept_variance_prior, isamps, rsamp, seed, chain, case_control=None):
        self.Y = None if Y is None else np.ascontiguousarray(Y, dtype=np.float64)
        self.cc = case_control
        self.st = orc.ChainState(X, grid, Y=self.Y, intercept=np.ravel(intercept), radii=np.array(radii),
                                 model=2 if case_control is not None else 1, case_control=case_control,
                                 mu=mu, sigma=sigma, lmbda=lmbda, z=z, seed=seed, chain=chain)
        self.beta = np.array(beta, dtype=np.float64)
        self.weights = np.array(weights, dtype=np.float64)
        self.lmbda = np.array(np.ravel(lmbda)[:1], dtype=np.float64)
        self.hp = hp
        self.ip = np.array(np.ravel(intercept_prior)[:2], dtype=np.float64)
        self.var = float(intercept_variance_prior)
        self.isamps, self.rsamp = isamps, rsamp
        self.seed, self.chain = seed, chain
        self.aux = None

    @property
    def intercept(self):
        return self.st.intercept[:2]

    @property
    def radii(self):
        return self.st.radii

    def loglik_at(self, X, b, r):
        if self.cc is not None:
            return orc.approx_directed_network_loglikelihood(
                X, r, self.cc['in_edges'], self.cc['out_edges'], self.cc['degree'],
                self.cc['control_nodes_out'], b[0], b[1])
        return orc.dynamic_network_loglikelihood_directed(self.Y, X, b[0], b[1], r)

    def iteration(self, it):
        st = self.st
        st.c.iter = it
        st.c.lmbda = float(self.lmbda[0])
        st.sweep_c()
        st.X[:] = orc.center(st.X)
        ll = orc.directed_coefficient_steps(st, it, self.loglik_at, self.isamps, self.rsamp, self.ip, self.var)
        z, n, nk = orc.sample_labels_block_philox(st.X, st.mu, st.sigma, self.lmbda, self.weights,
                                                  self.seed, self.chain, it)
        st.z[:] = z
        self.beta, self.lmbda, self.aux = gibbs_updates(
            st.X, st.z, n, nk, st.mu, st.sigma, self.beta, self.weights, self.lmbda, self.hp,
            PhiloxDraws(self.seed, self.chain, it))
        self.lmbda = np.array(np.ravel(self.lmbda)[:1], dtype=np.float64)
        self.n, self.nk = n, nk
        return log_posterior(ll, st.X, self.intercept, st.mu, st.sigma, st.z, self.weights, self.beta,
                             self.lmbda, self.hp, self.ip, self.var, n_radii=st.radii.shape[0])
