"""DynamicNetworkLPCM (the finite-mixture latent position cluster model with a
time-homogeneous label chain) with the reference's constructor, ``fit(Y)`` and per-sample
traces (lpcm.py:135-760), Gibbs loop on one MI355X.

Every device piece is shared with DynamicNetworkHDPLPCM: latent-position sweep with the
AR-mixture prior, centring, intercept (and radii) MH over fused log-likelihood passes, the
label block update (``sample_labels_block_lpcm`` is ``sample_labels_block`` with w[t] = the
transition matrix for every t >= 1 and w[0, 0] = the initial distribution), the label-wise
sums of the conjugate updates and the post-loop co-occurrence / expected-VI kernels.  The
host keeps the O(K^2) Dirichlet draws and the cluster draws in the reference's MT19937
order (``hdp_updates.lpcm_gibbs_updates``).
"""
import time

import numpy as np

from .engine import Chain, SamplerGrid, check_n_features
from . import forecast as fc
from . import hdp_updates as hu
from . import initialization as init_mod
from . import posterior as post
from .imputer import SimpleNetworkImputer
from .metrics import FittedQuantities
from .lsm import (DynamicNetworkLSM, _ScalarMetropolis, _dirichlet_logpdf,
                  check_random_state)

__all__ = ['DynamicNetworkLPCM']


class DynamicNetworkLPCM(FittedQuantities):
    """Constructor parameters are the reference's (lpcm.py:135-187) plus ``device``,
    ``chain_id`` and ``sweep_algo``."""

    def __init__(self, n_features=2, n_components=5, is_directed=False, selection_type='map',
                 n_iter=5000, tune=2500, tune_interval=100, burn=2500, thin=None,
                 intercept_prior='auto', intercept_variance_prior=2, mean_variance_prior='auto',
                 a=2.0, b='auto', lambda_prior=0.9, lambda_variance_prior=0.01,
                 dirichlet_prior='uniform', sigma_prior_std=4.0, mean_variance_prior_std=4.0,
                 step_size_X='auto', step_size_intercept=0.1, step_size_radii=175000,
                 n_control=None, n_resample_control=100, copy=True, random_state=None,
                 device=0, chain_id=0, sweep_algo=0):
        self.n_iter = n_iter
        self.is_directed = is_directed
        self.selection_type = selection_type
        self.n_features = n_features
        self.n_components = n_components
        self.dirichlet_prior = dirichlet_prior
        self.step_size_X = step_size_X
        self.intercept_prior = intercept_prior
        self.intercept_variance_prior = intercept_variance_prior
        self.step_size_intercept = step_size_intercept
        self.mean_variance_prior = mean_variance_prior
        self.a = a
        self.b = b
        self.lambda_prior = lambda_prior
        self.lambda_variance_prior = lambda_variance_prior
        self.mean_variance_prior_std = mean_variance_prior_std
        self.sigma_prior_std = sigma_prior_std
        self.step_size_radii = step_size_radii
        self.tune = tune
        self.tune_interval = tune_interval
        self.burn = burn
        self.thin = thin
        self.n_control = n_control
        self.n_resample_control = n_resample_control
        self.copy = copy
        self.random_state = random_state
        self.device = device
        self.chain_id = chain_id
        self.sweep_algo = sweep_algo

    @property
    def n_burn_(self):
        n_burn = (self.burn or 0) + (self.tune or 0)
        return int(np.ceil(n_burn / self.thin)) if self.thin else n_burn      # lpcm.py:189-197

    # -- one-step-ahead forecasts (lpcm.py:228-318; undirected models) ----------------
    def _forecast_ready(self):
        if not hasattr(self, 'X_'):
            raise ValueError('Model not fit.')
        if self.is_directed:
            raise ValueError('forecasts are implemented for undirected models '
                             '(as the reference formulas are)')
        return self.chain_

    @property
    def forecast_probas_map_(self):
        return fc.lpcm_forecast_probas_map(self, self._forecast_ready())

    @property
    def forecast_probas_plugin_(self):
        return fc.lpcm_forecast_probas_plugin(self, self._forecast_ready())

    @property
    def forecast_probas_marginalized_(self):
        return fc.lpcm_forecast_probas_marginalized(self, self._forecast_ready())

    def forecast_probas(self, n_samples=5000):
        return fc.lpcm_forecast_probas(self, self._forecast_ready(), n_samples=n_samples)

    # ------------------------------------------------------------------ init
    def _init_sampler(self, Y, Y_raw, rng, init):
        """lpcm.py:45-132: LSM warm start, longitudinal k-means, empirical initial
        distribution, uniform transition matrix; missing dyads are re-imputed by
        thresholding the warm start's edge probabilities (:77-81)"""
        T, N, _ = Y.shape
        K, D = self.n_components, check_n_features(self.n_features)
        if init is not None and 'X' in init:
            X = np.array(init['X'], dtype=np.float64)
            intercept = np.atleast_1d(np.asarray(init['intercept'], dtype=np.float64)).copy()
            radii = np.array(init['radii'], dtype=np.float64) if self.is_directed else None
        else:
            kw = (dict(sigma_sq=0.001, tau_sq='auto', step_size_X=0.0075,
                       n_control=self.n_control, n_resample_control=self.n_resample_control)
                  if self.is_directed else dict(sigma_sq=0.1, tau_sq=2.0, step_size_X=0.1))
            emb = DynamicNetworkLSM(n_iter=500, n_features=D, tune=250, burn=250,
                                    is_directed=self.is_directed, random_state=rng,
                                    device=self.device, chain_id=self.chain_id,
                                    sweep_algo=self.sweep_algo, **kw).fit(Y_raw)
            X, intercept = emb.X_.copy(), np.array(emb.intercept_, dtype=np.float64)
            radii = emb.radii_.copy() if self.is_directed else None
            if np.any(Y_raw == -1):
                Y = Y_raw.copy()
                Y[Y_raw == -1] = emb.probas_[Y_raw == -1] > 0.5
            emb.chain_.close()
        if init is not None and 'mu' in init:
            mu = np.array(init['mu'], dtype=np.float64)
            sigma = np.array(init['sigma'], dtype=np.float64)
            z = np.array(init['z'], dtype=np.int64)
        else:
            mu, sigma, z = init_mod.longitudinal_kmeans(X, n_clusters=K, random_state=rng)
            z = z.astype(np.int64)
        init_w = np.bincount(z[0], minlength=K) / float(N)
        trans_w = np.full((K, K), 1. / K)
        check_random_state(rng)                       # lpcm.py:126 (no draw)
        lmbda = np.array([self.lambda_prior], dtype=np.float64)
        return Y, X, intercept, mu, sigma, z, init_w, trans_w, lmbda, radii

    @staticmethod
    def _stack(init_w, trans_w, T):
        w = np.empty((T,) + trans_w.shape)
        w[:] = trans_w[None]
        w[0, 0] = init_w
        return w

    # ------------------------------------------------------------------- fit
    def fit(self, Y, init=None):
        Y_raw = np.array(Y, dtype=np.float64, copy=self.copy, order='C')
        if Y_raw.ndim != 3 or Y_raw.shape[1] != Y_raw.shape[2]:
            raise ValueError('Y must have shape (n_time_steps, n_nodes, n_nodes)')
        if np.any(np.isnan(Y_raw)):
            raise ValueError('NaN entries are not supported: code missing dyads as -1')
        if self.n_control is not None and not self.is_directed:
            raise ValueError('The case-control likelihood currently only '
                             'supported for directed networks.')
        T, N, _ = Y_raw.shape
        K, D = self.n_components, check_n_features(self.n_features)
        rng = check_random_state(self.random_state)
        self.nan_mask_, miss = None, None
        Y = Y_raw
        if np.any(Y_raw == -1):                            # lpcm.py:352-366
            if not self.is_directed:
                miss = np.nonzero(np.triu(Y_raw == -1, 1))
                iu = np.nonzero(np.triu(np.ones(Y_raw.shape, dtype=bool), 1))
                self.nan_mask_ = Y_raw[iu] == -1
                self.missings_ = np.zeros(miss[0].shape[0])
            else:
                off = np.nonzero(~np.eye(N, dtype=bool)[None].repeat(T, 0))
                self.nan_mask_ = Y_raw[off] == -1
            Y = SimpleNetworkImputer(strategy='random', missing_value=-1).fit_transform(Y_raw)
        if self.burn is not None:
            self.n_iter += self.burn
        if self.tune is not None:
            self.n_iter += self.tune
        n_total = self.n_iter
        (Y, X, intercept, mu, sigma, z, init_w, trans_w, lmbda, radii) = \
            self._init_sampler(Y, Y_raw, rng, init)
        self.Y_fit_ = Y
        self.dirichlet_prior_ = 1. if self.dirichlet_prior == 'uniform' else 1. / K
        if isinstance(self.step_size_X, str) and self.step_size_X == 'auto':
            self.step_size_X = 0.01 if self.is_directed else 0.1
        if isinstance(self.intercept_prior, str) and self.intercept_prior == 'auto':
            self.intercept_prior = intercept.copy()
        ip = np.atleast_1d(np.asarray(self.intercept_prior, dtype=np.float64))
        if isinstance(self.mean_variance_prior, str) and self.mean_variance_prior == 'auto':
            mvp = (2 * (1. / N) ** (2. / D) if self.is_directed else (N ** (2. / D)) / 50.)
        else:
            mvp = self.mean_variance_prior
        hp = hu.HDPHyper(K, mean_variance_prior=mvp, a=self.a, lambda_prior=self.lambda_prior,
                         lambda_variance_prior=self.lambda_variance_prior)
        if self.mean_variance_prior_std is not None:
            hp.a0 = (self.mean_variance_prior_std ** 2 + 2) * 2
            hp.b0 = (hp.a0 - 2) * mvp * 2
        hp.b = (self.a + 2) * mvp if (isinstance(self.b, str) and self.b == 'auto') else self.b
        if self.sigma_prior_std is not None:
            hp.d0 = (self.sigma_prior_std ** 2 / hp.b) * 2
            hp.c0 = hp.b * hp.d0
        self.hyper_ = hp

        model = ('undirected' if not self.is_directed else
                 'case_control' if self.n_control is not None else 'directed')
        seed = int(rng.randint(0, 2 ** 31 - 1)) | (int(rng.randint(0, 2 ** 31 - 1)) << 31)
        chain = Chain(T, N, D, model, seed=seed, chain_id=self.chain_id, device=self.device)
        self.chain_ = chain
        self.case_control_sampler_ = None
        if model == 'case_control':
            from .case_control import DirectedCaseControlSampler
            self.case_control_sampler_ = DirectedCaseControlSampler(
                n_control=self.n_control, n_resample=self.n_resample_control,
                chain=chain).init(Y)
        else:
            chain.upload_network(Y)
        chain.set_positions(X)
        chain.set_intercepts(intercept)
        if self.is_directed:
            chain.set_radii(radii)
        self.latent_samplers = SamplerGrid(T, N, self.step_size_X, tune=self.tune,
                                           tune_interval=self.tune_interval)
        chain.set_samplers(self.latent_samplers)
        n_ic = 2 if self.is_directed else 1
        isamp = [_ScalarMetropolis(self.step_size_intercept, self.tune) for _ in range(n_ic)]
        rsamp = _ScalarMetropolis(self.step_size_radii, self.tune, dirichlet=True)
        self.intercept_samplers, self.radii_sampler = isamp, rsamp

        self.Xs_ = np.zeros((n_total, T, N, D))
        self.intercepts_ = np.zeros((n_total, n_ic))
        self.mus_ = np.zeros((n_total, K, D))
        self.sigmas_ = np.zeros((n_total, K))
        self.zs_ = np.zeros((n_total, T, N), dtype=np.int64)
        self.init_weights_ = np.zeros((n_total, K))
        self.trans_weights_ = np.zeros((n_total, K, K))
        self.lambdas_ = np.zeros((n_total, 1))
        self.radiis_ = np.zeros((n_total, N)) if self.is_directed else None
        self.logps_ = np.zeros(n_total)

        def store(it, ll):
            self.Xs_[it], self.intercepts_[it] = X, intercept
            self.mus_[it], self.sigmas_[it], self.zs_[it] = mu, sigma, z
            self.init_weights_[it], self.trans_weights_[it] = init_w, trans_w
            self.lambdas_[it] = lmbda
            if self.is_directed:
                self.radiis_[it] = radii
            self.logps_[it] = np.ravel(ll + hu.lpcm_log_posterior_terms(
                sums, intercept, ip, self.intercept_variance_prior, mu, sigma, init_w, trans_w,
                lmbda, hp, self.dirichlet_prior_, radii=radii))[0]

        chain.set_prior_mixture(mu, sigma, lmbda, z)
        sums = hu.DeviceLabelSums(chain)
        store(0, chain.loglik_full())
        var = self.intercept_variance_prior
        t_loop = time.perf_counter()
        for it in range(1, n_total):
            if self.case_control_sampler_ is not None:
                self.case_control_sampler_.resample(it)
            chain.set_prior_mixture(mu, sigma, lmbda, None)      # z: the device keeps its own
            chain.sweep_positions(it, self.sweep_algo)
            chain.center()
            for k in range(n_ic):                    # sample_coefficients.py:12-88
                prop = intercept.copy()
                prop[k] = intercept[k] + isamp[k].step_size * rng.randn(1)[0]
                if k == 0:
                    ll_prop, ll_cur = chain.loglik_full([prop, intercept])
                else:       # same positions, and `intercept` is the state the last step left:
                    ll_prop, ll_cur = chain.loglik_full([prop])[0], ll      # its value is known
                ratio = ((ll_prop - (prop[k] - ip[k]) ** 2 / (2 * var)) -
                         (ll_cur - (intercept[k] - ip[k]) ** 2 / (2 * var)))
                accepted = int(not (np.log(rng.rand()) >= ratio))
                ll = ll_cur
                if accepted:
                    intercept, ll = prop, ll_prop
                isamp[k].book(accepted)
            chain.set_intercepts(intercept)
            if self.is_directed:                     # sample_coefficients.py:91-121
                x = rng.dirichlet(rsamp.step_size * radii)
                if np.any(x == 0.):
                    x += 1e-5
                    x /= np.sum(x)
                ll_cur, ll_prop = chain.loglik_full_radii(x)
                ratio = (ll_prop - ll_cur + _dirichlet_logpdf(radii, rsamp.step_size * x) -
                         _dirichlet_logpdf(x, rsamp.step_size * radii))
                accepted = int(not (np.log(rng.rand()) >= ratio))
                ll = ll_cur
                if accepted:
                    radii, ll = x, ll_prop
                    chain.set_radii(radii)
                rsamp.book(accepted)
            z, n, nk = chain.sample_labels(it, self._stack(init_w, trans_w, T))
            X = chain.get_positions()
            mu, sigma = mu.copy(), sigma.copy()
            init_w, trans_w = init_w.copy(), trans_w.copy()
            lmbda = hu.lpcm_gibbs_updates(sums, n, nk, mu, sigma, init_w, trans_w, lmbda, hp,
                                          rng, self.dirichlet_prior_)
            if miss is not None:                     # lpcm.py:676-689
                dm = X[miss[0], miss[1]] - X[miss[0], miss[2]]
                eta = intercept[0] - np.sqrt(np.sum(dm * dm, axis=1))
                y_ij = rng.binomial(1, 1. / (1. + np.exp(-eta)))
                if it > self.n_burn_:
                    self.missings_ += y_ij
            store(it, ll)
        self.loop_seconds_ = time.perf_counter() - t_loop
        if miss is not None:
            self.missings_ /= max(1, n_total - self.n_burn_)
        chain.get_samplers(self.latent_samplers)
        self.mean_variance_prior_, self.b_ = hp.mean_variance_prior, hp.b

        if self.thin is not None:                    # lpcm.py:703-716
            for name in ('Xs_', 'intercepts_', 'mus_', 'sigmas_', 'zs_', 'init_weights_',
                         'trans_weights_', 'lambdas_', 'logps_'):
                setattr(self, name, getattr(self, name)[::self.thin])
            if self.is_directed:
                self.radiis_ = self.radiis_[::self.thin]
        n_burn = min(self.n_burn_, self.logps_.shape[0] - 1)
        self.cooccurrence_probas_ = post.posterior_cooccurrences(self, chain, n_burn)
        if self.selection_type == 'map':
            # lpcm.py:724 takes the argmax over logps_[n_burn:] and uses it as an index into
            # the full trace (without adding n_burn); kept as it is
            best = int(np.argmax(self.logps_[n_burn:]))
        else:
            best, self.expected_vis_ = post.minimize_posterior_expected_vi(
                self, chain, n_burn, cooc=self.cooccurrence_probas_)
        chain.post_release()
        self.selected_id_ = best
        self.logp_ = self.logps_[best]
        self.X_, self.intercept_ = self.Xs_[best].copy(), self.intercepts_[best]
        self.lambda_ = self.lambdas_[best]
        if self.is_directed:
            self.radii_ = self.radiis_[best]
        self.z_ = self.zs_[best]
        self.init_weight_, self.trans_weight_ = self.init_weights_[best], self.trans_weights_[best]
        self.mu_, self.sigma_ = self.mus_[best].copy(), self.sigmas_[best]
        post.procrustes_align_samples(self)          # lpcm.py:742-749
        self.X_mean_ = self.Xs_[n_burn:].mean(axis=0)
        self.lambda_mean_ = self.lambdas_[n_burn:].mean(axis=0)
        self.intercepts_mean_ = self.intercepts_[n_burn:].mean(axis=0)
        if self.is_directed:
            self.radii_mean_ = self.radiis_[n_burn:].mean(axis=0)
        return self
