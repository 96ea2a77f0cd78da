"""DynamicNetworkHDPLPCM with the reference's constructor, ``fit(Y)`` and the
per-sample traces (hdp_lpcm.py:144-1083), Gibbs loop on one MI355X.

Per iteration (hdp_lpcm.py:823-1069): latent-position sweep with the AR-mixture
prior, centring, intercept (and radii) MH, label block update -- all kernels of
the engine -- then the O(TN + TK^2) conjugate / auxiliary updates in numpy
(``hdp_updates``) and the log-posterior trace.

After the loop (hdp_lpcm.py:1072-1176; ``posterior``): thinning, model selection by
minimum posterior expected VI (default), BIC or MAP size - the posterior
co-occurrence matrices and the VI criterion of every kept sample are computed on
the device - weight renormalisation, Procrustes alignment of the stored samples,
posterior means and group counts.  Out of scope: Geweke
diagnostics and forecasting.  Missing edges are rejected.
"""
import time

import numpy as np

from .engine import Chain, SamplerGrid, check_n_features
from . import hdp_updates as hu
from . import initialization as init_mod
from . import posterior as post
from . import forecast as fc
from .diagnostics import geweke_diag
from .imputer import SimpleNetworkImputer
from .metrics import FittedQuantities
from .lsm import (DynamicNetworkLSM, _ScalarMetropolis, _dirichlet_logpdf,
                  check_random_state)

__all__ = ['DynamicNetworkHDPLPCM']



def _geweke(trace, n_burn):
    """(z, p) of diagnostics.geweke_diag; (nan, nan) for traces too short for its AR fits"""
    if not np.all(np.isfinite(np.asarray(trace)[n_burn:])):     # e.g. a -inf log-posterior at the start
        return float('nan'), float('nan')
    try:
        return geweke_diag(trace, n_burn=n_burn)
    except (ValueError, TypeError, np.linalg.LinAlgError, ZeroDivisionError):
        return float('nan'), float('nan')


class DynamicNetworkHDPLPCM(FittedQuantities):
    """Constructor parameters are the reference's (hdp_lpcm.py:385-455) plus
    ``device``, ``chain_id``, ``sweep_algo`` and ``hdp_loop``:

    ``hdp_loop='device'`` runs the whole Gibbs iteration on the GPU (``dlsm_hdp_run``: every
    draw of hdp_lpcm.py:876-1023 from Philox counters, no host round trip inside an
    iteration; directed and case-control models too: their two intercept steps and the radii
    step run as in the directed LSM's device loop); ``'host'`` keeps the auxiliary / conjugate
    draws in numpy on the caller's MT19937 stream in the reference's order (the bit-level pin to
    the reference's ``_fit`` trace).  ``'auto'`` = device for undirected models, host for the
    directed ones.  The two are equal in distribution.

    After a device-resident loop the post-loop processing (hdp_lpcm.py:1085-1162: model selection,
    Procrustes alignment of every stored sample, posterior means) runs on the trace where it lies,
    and ``Xs_``, ``zs_``, ``weights_`` and ``cooccurrence_probas_`` are copied to the host when they
    are first read (320 KB per stored sample of positions at T=10, N=2000);
    ``post_processing='host'`` copies the whole trace first and processes it in numpy, as thinning
    and missing dyads always do.  Until they are read the four arrays are not in ``vars(model)``:
    ``materialize()`` reads them all (before copying or pickling the estimator),
    ``release_device_trace()`` then frees the chain's device memory; a read that fails raises
    ``AttributeError`` (chained to the engine's error), so ``hasattr`` / ``getattr`` defaults work."""

    def __init__(self, n_features=2, n_components=10, is_directed=False,
                 selection_type='vi', n_iter=5000, tune=2500, tune_interval=100,
                 burn=2500, thin=None, gamma=1.0, gamma_prior_shape=1.0,
                 gamma_prior_rate=0.1, alpha_init=1.0, alpha_init_shape=1.,
                 alpha_init_rate=1., alpha=1.0, kappa=4.0, alpha_kappa_shape=5,
                 alpha_kappa_rate=0.1, intercept_prior='auto', intercept_variance_prior=2,
                 mean_variance_prior='auto', a=2.0, b='auto', lambda_prior=0.9,
                 lambda_variance_prior=0.01, sigma_prior_std=4.0,
                 mean_variance_prior_std=4.0, step_size_X='auto', step_size_intercept=0.1,
                 step_size_radii=175000, n_control=None, n_resample_control=100, copy=True,
                 random_state=None, device=0, chain_id=0, sweep_algo=0, hdp_loop='auto',
                 post_processing='auto'):
        self.n_iter = n_iter
        self.hdp_loop = hdp_loop
        self.is_directed = is_directed
        self.n_features = n_features
        self.n_components = n_components
        self.step_size_X = step_size_X
        self.intercept_prior = intercept_prior
        self.intercept_variance_prior = intercept_variance_prior
        self.step_size_intercept = step_size_intercept
        self.mean_variance_prior = mean_variance_prior
        self.a = a
        self.b = b
        self.alpha_init = alpha_init
        self.alpha = alpha
        self.alpha_init_shape = alpha_init_shape
        self.alpha_init_rate = alpha_init_rate
        self.gamma = gamma
        self.gamma_prior_shape = gamma_prior_shape
        self.gamma_prior_rate = gamma_prior_rate
        self.kappa = kappa
        self.alpha_kappa_shape = alpha_kappa_shape
        self.alpha_kappa_rate = alpha_kappa_rate
        self.lambda_prior = lambda_prior
        self.lambda_variance_prior = lambda_variance_prior
        self.mean_variance_prior_std = mean_variance_prior_std
        self.sigma_prior_std = sigma_prior_std
        self.step_size_radii = step_size_radii
        self.tune = tune
        self.tune_interval = tune_interval
        self.burn = burn
        self.thin = thin
        self.selection_type = selection_type
        self.n_control = n_control
        self.n_resample_control = n_resample_control
        self.copy = copy
        self.random_state = random_state
        self.device = device
        self.chain_id = chain_id
        self.sweep_algo = sweep_algo
        self.post_processing = post_processing

    @property
    def n_burn_(self):
        return (self.burn or 0) + (self.tune or 0)

    # -- one-step-ahead forecasts (hdp_lpcm.py:496-626; undirected models) ----------
    def _forecast_ready(self):
        if not hasattr(self, 'X_'):
            raise ValueError('Model not fit.')
        if self.is_directed:
            raise ValueError('forecasts are implemented for undirected models '
                             '(as the reference formulas are)')
        return self.chain_

    @property
    def forecast_probas_map_(self):
        return fc.forecast_probas_map(self, self._forecast_ready())

    @property
    def forecast_probas_plugin_(self):
        return fc.forecast_probas_plugin(self, self._forecast_ready())

    @property
    def forecast_probas_marginalized_(self):
        return fc.forecast_probas_marginalized(self, self._forecast_ready())

    def forecast_probas(self, n_samples=5000):
        return fc.forecast_probas(self, self._forecast_ready(), n_samples=n_samples)

    @property
    def forecast_probas_pp_(self):
        return fc.forecast_probas_pp(self, self._forecast_ready())

    # ------------------------------------------------------------------ init
    def _init_sampler(self, Y, rng, init):
        """hdp_lpcm.py:48-141 : LSM warm start, longitudinal k-means, weights"""
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
                                    sweep_algo=self.sweep_algo, **kw).fit(Y)
            X, intercept = emb.X_.copy(), np.array(emb.intercept_, dtype=np.float64)
            radii = emb.radii_.copy() if self.is_directed else None
            emb.chain_.close()
        if init is not None and 'mu' in init:
            mu = np.array(init['mu'], dtype=np.float64)
            sigma = np.array(init['sigma'], dtype=np.float64)
            z = np.array(init['z'], dtype=np.int64)
        else:
            # the Lloyd iterations of longitudinal_kmeans run on the device (any handle will do)
            with Chain(1, N, D, 'undirected', device=self.device) as tmp:
                mu, sigma, z = init_mod.longitudinal_kmeans(X, n_clusters=K, random_state=rng,
                                                            chain=tmp)
            z = z.astype(np.int64)
        weights = np.zeros((T, K, K))
        weights[0, 0] = np.bincount(z[0], minlength=K) / N
        lmbda = np.array([self.lambda_prior], dtype=np.float64)
        beta = rng.dirichlet(np.repeat(self.gamma / K, K))
        for t in range(1, T):
            for k in range(K):
                weights[t, k] = rng.dirichlet(self.alpha * beta + self.kappa * np.eye(K)[k])
        return X, intercept, mu, sigma, z, beta, weights, lmbda, radii

    # ------------------------------------------------------------------- fit
    def fit(self, Y, init=None):
        """Sample the posterior of the HDP-LPCM given ``Y`` (T, N, N).  ``init``
        may carry starting values ``X, intercept[, radii][, mu, sigma, z]``."""
        self._prepare(Y, init)
        t_loop = time.perf_counter()
        self._run(1, self._n_total - 1)
        self.chain_.synchronize()
        self.loop_seconds_ = time.perf_counter() - t_loop     # Gibbs loop only
        return self._finish()

    def _prepare(self, Y, init=None, network_from=None):
        """Everything of ``fit`` before the Gibbs loop (hdp_lpcm.py:628-821): checks,
        starting values, hyper-priors, the chain handle and the trace arrays.
        ``network_from``: callable(chain) that loads the network into the chain some other
        way than the float64 upload (multi-GPU: the packed broadcast of multichain)."""
        # (copy=False keeps the caller's array as it is - a network_from loader never reads it)
        Y = (np.array(Y, dtype=np.float64, order='C') if self.copy or network_from is None
             else np.asarray(Y, dtype=np.float64))
        if Y.ndim != 3 or Y.shape[1] != Y.shape[2]:
            raise ValueError('Y must have shape (n_time_steps, n_nodes, n_nodes)')
        if np.any(np.isnan(Y)):
            raise ValueError('NaN entries are not supported: code missing dyads as -1')
        # missing dyads (hdp_lpcm.py:669-706): imputed once; undirected models also average
        # per-iteration Bernoulli draws of them after burn-in into ``missings_``
        self.nan_mask_, miss = None, None
        if np.any(Y == -1):
            if not self.is_directed:
                miss = np.nonzero(np.triu(Y == -1, 1))            # (t, i, j), row-major
                iu = np.nonzero(np.triu(np.ones(Y.shape, dtype=bool), 1))
                self.nan_mask_ = Y[iu] == -1
                self.missings_ = np.zeros(miss[0].shape[0])
            else:
                off = np.nonzero(~np.eye(Y.shape[1], dtype=bool)[None].repeat(Y.shape[0], 0))
                self.nan_mask_ = Y[off] == -1
            Y = SimpleNetworkImputer(strategy='random', missing_value=-1).fit_transform(Y)
        if self.selection_type not in ('vi', 'bic', 'map'):
            raise ValueError('Selection type not recognized')
        if self.n_control is not None and not self.is_directed:
            raise ValueError('The case-control likelihood currently only '
                             'supported for directed networks.')
        T, N, _ = Y.shape
        K, D = self.n_components, check_n_features(self.n_features)
        rng = check_random_state(self.random_state)
        self.Y_fit_ = Y
        if self.burn is not None:
            self.n_iter += self.burn
        if self.tune is not None:
            self.n_iter += self.tune
        n_total = self.n_iter

        (X, intercept, mu, sigma, z, beta, weights, lmbda, radii) = \
            self._init_sampler(Y, rng, init)
        if isinstance(self.step_size_X, str) and self.step_size_X == 'auto':
            self.step_size_X = 0.01 if self.is_directed else 0.1
        if isinstance(self.intercept_prior, str) and self.intercept_prior == 'auto':
            self.intercept_prior = intercept.copy()
        ip = np.atleast_1d(np.asarray(self.intercept_prior, dtype=np.float64))

        # hyper-priors (hdp_lpcm.py:760-793)
        if isinstance(self.mean_variance_prior, str) and self.mean_variance_prior == 'auto':
            mvp = (2 * (1. / N) ** (2. / D) if self.is_directed else (N ** (2. / D)) / 50.)
        else:
            mvp = self.mean_variance_prior
        hp = hu.HDPHyper(K, gamma=self.gamma, alpha_init=self.alpha_init, alpha=self.alpha,
                         kappa=self.kappa, mean_variance_prior=mvp, a=self.a,
                         lambda_prior=self.lambda_prior,
                         lambda_variance_prior=self.lambda_variance_prior,
                         gamma_prior_shape=self.gamma_prior_shape,
                         gamma_prior_rate=self.gamma_prior_rate,
                         alpha_init_shape=self.alpha_init_shape,
                         alpha_init_rate=self.alpha_init_rate,
                         alpha_kappa_shape=self.alpha_kappa_shape,
                         alpha_kappa_rate=self.alpha_kappa_rate)
        if self.mean_variance_prior_std is not None:
            hp.a0 = (self.mean_variance_prior_std ** 2 + 2) * 2
            hp.b0 = (hp.a0 - 2) * mvp * 2
        hp.b = (self.a + 2) * mvp if (isinstance(self.b, str) and self.b == 'auto') else self.b
        if self.sigma_prior_std is not None:
            hp.d0 = (self.sigma_prior_std ** 2 / hp.b) * 2
            hp.c0 = hp.b * hp.d0
        self.hyper_ = hp

        # ---- the chain -----------------------------------------------------
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
        elif network_from is not None:
            network_from(chain)
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
        # hdp_lpcm.py:731-747: intercept samplers keep the default tune_interval
        isamp = [_ScalarMetropolis(self.step_size_intercept, self.tune) for _ in range(n_ic)]
        rsamp = _ScalarMetropolis(self.step_size_radii, self.tune, dirichlet=True)
        self.intercept_samplers, self.radii_sampler = isamp, rsamp

        if self.hdp_loop not in ('auto', 'device', 'host'):
            raise ValueError("hdp_loop must be 'auto', 'device' or 'host'")
        # 'auto': the undirected model's loop runs on the device; the directed models' does when
        # it is asked for (their default stays the host-driven loop with the reference's MT19937
        # draw order, which the recorded reference traces pin)
        self.loop_kind_ = ('device-resident' if (self.hdp_loop == 'device' or
                                                 (self.hdp_loop == 'auto' and not self.is_directed))
                           else 'host-driven')
        # The device-resident loop keeps its trace in HBM; the three large arrays - Xs_
        # (320 KB per sample at T=10, N=2000), zs_, weights_ - reach the host only when somebody
        # reads them (__getattr__), and the post-loop processing runs where they lie.
        self._lazy_trace = False
        self._lazy_cooc = False
        for name in ('Xs_', 'zs_', 'weights_', 'cooccurrence_probas_'):
            self.__dict__.pop(name, None)
        if self.loop_kind_ == 'host-driven':
            self.Xs_ = np.zeros((n_total, T, N, D))
            self.zs_ = np.zeros((n_total, T, N), dtype=np.int64)
            self.weights_ = np.zeros((n_total, T, K, K))
        self.intercepts_ = np.zeros((n_total, n_ic))
        self.mus_ = np.zeros((n_total, K, D))
        self.sigmas_ = np.zeros((n_total, K))
        self.betas_ = np.zeros((n_total, K))
        self.lambdas_ = np.zeros((n_total, 1))
        self.radiis_ = np.zeros((n_total, N)) if self.is_directed else None
        self.logps_ = np.zeros(n_total)

        chain.set_prior_mixture(mu, sigma, lmbda, z)
        self._n_total, self._rng, self._ip, self._miss = n_total, rng, ip, miss
        self._sums = hu.DeviceLabelSums(chain)  # label-wise sums at the chain's X and z
        self._st = dict(X=X, intercept=intercept, mu=mu, sigma=sigma, z=z, beta=beta,
                        weights=weights, lmbda=lmbda, radii=radii)
        self._store(0, chain.loglik_full())
        if self.loop_kind_ == 'device-resident':
            chain.hdp_configure(hp, beta, weights, ip, self.intercept_variance_prior,
                                step_size_intercept=self.step_size_intercept, tune=self.tune,
                                tune_interval=100, sweep_algo=self.sweep_algo,
                                step_size_radii=self.step_size_radii, radii_tune=self.tune)
            chain.hdp_trace_alloc(n_total, logp0=float(self.logps_[0]))
        return self

    def _pull(self, first, count):
        """device-resident loop: the stored samples first .. first + count - 1 -> the host
        trace arrays"""
        if self.loop_kind_ != 'device-resident' or count <= 0:
            return
        n_total, T, N = self._n_total, self.chain_.T, self.chain_.N
        K, D = self.n_components, self.n_features
        if 'Xs_' not in self.__dict__:      # the eager path: every array on the host
            self.Xs_ = np.zeros((n_total, T, N, D))
            self.zs_ = np.zeros((n_total, T, N), dtype=np.int64)
            self.weights_ = np.zeros((n_total, T, K, K))
            tr0 = self.chain_.hdp_trace_read(0, 1)
            self.Xs_[0], self.zs_[0], self.weights_[0] = tr0['Xs'][0], tr0['zs'][0], tr0['weights'][0]
        tr = self.chain_.hdp_trace_read(first, count)
        sl = slice(first, first + count)
        self.Xs_[sl], self.intercepts_[sl], self.logps_[sl] = tr['Xs'], tr['intercepts'], tr['logps']
        self.mus_[sl], self.sigmas_[sl], self.zs_[sl] = tr['mus'], tr['sigmas'], tr['zs']
        self.betas_[sl], self.weights_[sl], self.lambdas_[sl] = (tr['betas'], tr['weights'],
                                                                 tr['lambdas'])
        self.hypers_ = getattr(self, 'hypers_', np.zeros((self._n_total, 6)))
        self.hypers_[sl] = tr['hypers']
        if self.is_directed:
            self.radiis_[sl] = self.chain_.trace_read_radii(first, count)

    def _store(self, it, ll):
        st, hp = self._st, self.hyper_
        self.intercepts_[it] = st['intercept']
        self.mus_[it], self.sigmas_[it] = st['mu'], st['sigma']
        self.betas_[it] = st['beta']
        if 'Xs_' in self.__dict__:          # (device-resident loop: row 0 of its trace is the device's)
            self.Xs_[it], self.zs_[it], self.weights_[it] = st['X'], st['z'], st['weights']
        self.lambdas_[it] = st['lmbda']
        if self.is_directed:
            self.radiis_[it] = st['radii']
        self.logps_[it] = np.ravel(ll + hu.log_posterior_terms(
            self._sums, st['intercept'], self._ip, self.intercept_variance_prior, st['mu'],
            st['sigma'], st['weights'], st['beta'], st['lmbda'], hp, radii=st['radii']))[0]

    def _run(self, first, count):
        """Gibbs iterations first .. first + count - 1 (hdp_lpcm.py:823-1069), host-driven:
        the kernels of the engine around numpy draws on the caller's MT19937 stream."""
        chain, rng, hp, ip = self.chain_, self._rng, self.hyper_, self._ip
        if self.loop_kind_ == 'device-resident':
            ccs = self.case_control_sampler_
            if ccs is None:
                chain.hdp_run(first, count)      # asynchronous: the iterations are enqueued
                return
            # the host keeps the cadence of the control resampling (case_control_likelihood.py:27-33)
            it, last = first, first + count - 1
            while it <= last:
                ccs.resample(it)
                nxt = it + 1
                while nxt <= last and (ccs.n_resample is None or ccs.n_iter % ccs.n_resample != 0):
                    ccs.n_iter += 1
                    nxt += 1
                chain.hdp_run(it, nxt - it)
                it = nxt
            return
        isamp, rsamp, sums, miss = (self.intercept_samplers, self.radii_sampler, self._sums,
                                    self._miss)
        st = self._st
        X, intercept, mu, sigma, z = st['X'], st['intercept'], st['mu'], st['sigma'], st['z']
        beta, weights, lmbda, radii = st['beta'], st['weights'], st['lmbda'], st['radii']
        n_ic = 2 if self.is_directed else 1
        var = self.intercept_variance_prior
        for it in range(first, first + count):
            if self.case_control_sampler_ is not None:
                self.case_control_sampler_.resample(it)
            chain.set_prior_mixture(mu, sigma, lmbda, None)      # z: the device keeps its own
            chain.sweep_positions(it, self.sweep_algo)
            chain.center()
            # intercepts (sample_coefficients.py:12-88), fused two-candidate passes
            for k in range(n_ic):
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
            if self.is_directed:
                x = rng.dirichlet(rsamp.step_size * radii)
                if np.any(x == 0.):
                    x += 1e-5
                    x /= np.sum(x)
                ll_cur, ll_prop = chain.loglik_full_radii(x)
                ratio = (ll_prop - ll_cur +
                         _dirichlet_logpdf(radii, rsamp.step_size * x) -
                         _dirichlet_logpdf(x, rsamp.step_size * radii))
                accepted = int(not (np.log(rng.rand()) >= ratio))
                ll = ll_cur
                if accepted:
                    radii, ll = x, ll_prop
                    chain.set_radii(radii)
                rsamp.book(accepted)
            # label block update on the device (sample_labels.py:134-190)
            z, n, nk = chain.sample_labels(it, weights)
            X = chain.get_positions()
            mu, sigma, weights = mu.copy(), sigma.copy(), weights.copy()
            beta, lmbda = hu.gibbs_updates(sums, n, nk, mu, sigma, beta, weights, lmbda, hp, rng)
            if miss is not None:                  # hdp_lpcm.py:1039-1049
                dm = X[miss[0], miss[1]] - X[miss[0], miss[2]]
                eta = intercept[0] - np.sqrt(np.sum(dm * dm, axis=1))
                y_ij = rng.binomial(1, 1. / (1. + np.exp(-eta)))
                if it > self.n_burn_:
                    self.missings_ += y_ij
            st.update(X=X, intercept=intercept, mu=mu, sigma=sigma, z=z, beta=beta,
                      weights=weights, lmbda=lmbda, radii=radii)
            self._store(it, ll)

    # ---- the large arrays of a device-resident trace, on demand --------------------------------
    _LAZY = ('Xs_', 'zs_', 'weights_', 'cooccurrence_probas_')

    def __getattr__(self, name):            # reached only when the attribute is not there
        d = self.__dict__
        try:
            if name in ('Xs_', 'zs_', 'weights_') and d.get('_lazy_trace') and d.get('chain_') is not None:
                key = {'Xs_': 'Xs', 'zs_': 'zs', 'weights_': 'weights'}[name]
                tr = d['chain_'].hdp_trace_read(0, d['_n_total'], positions=name == 'Xs_',
                                                labels=name == 'zs_', weights=name == 'weights_',
                                                small=False)
                d[name] = tr[key]
                return d[name]
            if name == 'cooccurrence_probas_' and d.get('_lazy_cooc') and d.get('chain_') is not None:
                d[name] = d['chain_'].post_get_cooccurrence()
                d['chain_'].post_release()
                d['_lazy_cooc'] = False
                return d[name]
        except Exception as e:              # noqa: BLE001  (a closed handle, a failing copy)
            # hasattr / getattr(obj, name, default) rely on AttributeError
            raise AttributeError('%s could not be read from the device-resident trace: %s'
                                 % (name, e)) from e
        raise AttributeError(name)

    def materialize(self):
        """Read every array that still lives in the device-resident trace (``Xs_``, ``zs_``,
        ``weights_``, ``cooccurrence_probas_``: at T=10, N=2000 320 KB per sample and a 320 MB
        co-occurrence tensor) into the estimator's ``__dict__`` - what ``vars()``, ``copy`` and
        ``pickle`` see.  The reference holds them as plain attributes (hdp_lpcm.py:795-818); here
        they are fetched on first access, or all at once by this call.  Returns self."""
        for name in self._LAZY:
            if name not in self.__dict__:
                try:
                    getattr(self, name)
                except AttributeError:
                    pass
        return self

    def release_device_trace(self, materialize=True):
        """Free the chain handle - the trace and every other device buffer of this fit (about
        2 GB per 5000 samples at T=10, N=2000) - after reading the lazy arrays (``materialize=False``
        drops them instead).  Forecasts and further post-processing on the device are not possible
        afterwards."""
        if materialize:
            self.materialize()
        self._lazy_trace = False
        self._lazy_cooc = False
        ch = self.__dict__.get('chain_')
        if ch is not None:
            ch.close()
        return self

    def _finish_on_device(self):
        """``_finish`` for the device-resident loop without thinning or missing dyads: model
        selection, Procrustes alignment and the posterior means run on the trace in HBM
        (hdp_lpcm.py:1085-1162); the host receives the scalar traces, the cluster parameters and
        the selected sample."""
        chain, hp, n_total = self.chain_, self.hyper_, self._n_total
        tr = chain.hdp_trace_read(0, n_total, positions=False, labels=False, weights=False)
        self.intercepts_, self.logps_ = tr['intercepts'], tr['logps']
        self.mus_, self.sigmas_, self.betas_ = tr['mus'], tr['sigmas'], tr['betas']
        self.lambdas_, self.hypers_ = tr['lambdas'], tr['hypers']
        self._trace_logliks = tr['logliks']
        self._lazy_trace = True
        cfg = chain.hdp_get_config()
        hp.gamma, hp.alpha_init, hp.alpha, hp.kappa = cfg.gamma, cfg.alpha_init, cfg.alpha, cfg.kappa
        hp.mean_variance_prior, hp.b = cfg.mean_variance_prior, cfg.b
        sm = self.intercept_samplers[0]
        sm.step_size, sm.n_accepted = cfg.i_step_size, cfg.i_n_accepted
        sm.n_steps, sm.steps_until_tune = cfg.i_n_steps, cfg.i_steps_until_tune
        last = post._trace_row(chain, n_total - 1)
        self._st.update(X=last['X'], intercept=last['intercept'], mu=last['mu'], sigma=last['sigma'],
                        z=last['z'], beta=last['beta'], weights=last['weights'], lmbda=last['lmbda'])
        chain.get_samplers(self.latent_samplers)
        self.gamma, self.alpha_init, self.alpha, self.kappa = (hp.gamma, hp.alpha_init,
                                                               hp.alpha, hp.kappa)
        self.mean_variance_prior_, self.b_ = hp.mean_variance_prior, hp.b
        n_burn = min(self.n_burn_, n_total - 1)
        post.select_model_device(self, chain, n_burn)
        # Procrustes: every stored sample (and its cluster means) onto the selected one (:1141-1146)
        chain.post_trace_align(0, n_total, self.selected_id_)
        self.mus_ = chain.hdp_trace_read(0, n_total, positions=False, labels=False,
                                         weights=False)['mus']
        self.posterior_group_ids_, self.posterior_group_counts_ = post.posterior_group_counts_from(
            self._counts_t)
        self.X_mean_ = chain.post_trace_mean(n_burn, n_total - n_burn)
        self.lambda_mean_ = self.lambdas_[n_burn:].mean(axis=0)
        self.intercepts_mean_ = self.intercepts_[n_burn:].mean(axis=0)
        with np.errstate(all='ignore'):
            self.logp_geweke_ = _geweke(self.logps_, n_burn)
            self.lambda_geweke_ = _geweke(self.lambdas_[:, 0], n_burn)
            self.intercept_geweke_ = _geweke(self.intercepts_[:, 0], n_burn)
        return self

    def _finish(self):
        """Everything of ``fit`` after the Gibbs loop (hdp_lpcm.py:1072-1176)."""
        chain, hp, n_total = self.chain_, self.hyper_, self._n_total
        if (self.loop_kind_ == 'device-resident' and self.thin is None and self._miss is None
                and self.post_processing != 'host' and not self.is_directed):
            return self._finish_on_device()
        if self.loop_kind_ == 'device-resident':
            self._pull(1, n_total - 1)
            cfg = chain.hdp_get_config()
            hp.gamma, hp.alpha_init, hp.alpha, hp.kappa = (cfg.gamma, cfg.alpha_init, cfg.alpha,
                                                           cfg.kappa)
            hp.mean_variance_prior, hp.b = cfg.mean_variance_prior, cfg.b
            sm = self.intercept_samplers[0]
            sm.step_size, sm.n_accepted = cfg.i_step_size, cfg.i_n_accepted
            sm.n_steps, sm.steps_until_tune = cfg.i_n_steps, cfg.i_steps_until_tune
            self._st.update(X=self.Xs_[-1], intercept=self.intercepts_[-1], mu=self.mus_[-1],
                            sigma=self.sigmas_[-1], z=self.zs_[-1], beta=self.betas_[-1],
                            weights=self.weights_[-1], lmbda=self.lambdas_[-1])
            if self.is_directed:
                so, rs = self.intercept_samplers[1], self.radii_sampler
                so.step_size, so.n_accepted = cfg.i_step_size_out, cfg.i_n_accepted_out
                so.n_steps, so.steps_until_tune = cfg.i_n_steps_out, cfg.i_steps_until_tune_out
                rs.step_size, rs.n_accepted = cfg.r_step_size, cfg.r_n_accepted
                rs.n_steps, rs.steps_until_tune = cfg.r_n_steps, cfg.r_steps_until_tune
                self._st.update(radii=self.radiis_[-1])
            if self._miss is not None:            # hdp_lpcm.py:1039-1049 from the stored samples
                miss, rng = self._miss, self._rng
                for it in range(1, n_total):
                    X = self.Xs_[it]
                    dm = X[miss[0], miss[1]] - X[miss[0], miss[2]]
                    eta = self.intercepts_[it, 0] - np.sqrt(np.sum(dm * dm, axis=1))
                    y_ij = rng.binomial(1, 1. / (1. + np.exp(-eta)))
                    if it > self.n_burn_:
                        self.missings_ += y_ij
        if self._miss is not None:
            self.missings_ /= max(1, n_total - self.n_burn_)       # hdp_lpcm.py:1155-1156
        chain.get_samplers(self.latent_samplers)
        self.gamma, self.alpha_init, self.alpha, self.kappa = (hp.gamma, hp.alpha_init,
                                                               hp.alpha, hp.kappa)
        self.mean_variance_prior_, self.b_ = hp.mean_variance_prior, hp.b

        # thinning (hdp_lpcm.py:1072-1083)
        if self.thin is not None:
            for name in ('Xs_', 'intercepts_', 'mus_', 'sigmas_', 'zs_', 'betas_',
                         'weights_', 'lambdas_', 'logps_'):
                setattr(self, name, getattr(self, name)[::self.thin])
            if self.is_directed:
                self.radiis_ = self.radiis_[::self.thin]
        # model selection (hdp_lpcm.py:1085-1139): BIC / MAP size / minimum posterior
        # expected VI, with the co-occurrence matrices and the VI criterion on the device
        n_burn = min(-(-self.n_burn_ // (self.thin or 1)), self.logps_.shape[0] - 1)     # ceil: hdp_lpcm.py:465
        post.select_model(self, chain, n_burn)
        # Procrustes: rotate every stored sample onto the selected one (:1141-1146)
        post.procrustes_align_samples(self)
        self.posterior_group_ids_, self.posterior_group_counts_ = post.posterior_group_counts(
            self, n_burn)
        self.X_mean_ = self.Xs_[n_burn:].mean(axis=0)
        self.lambda_mean_ = self.lambdas_[n_burn:].mean(axis=0)
        self.intercepts_mean_ = self.intercepts_[n_burn:].mean(axis=0)
        # Geweke's diagnostic of the scalar traces (hdp_lpcm.py:1165-1176)
        with np.errstate(all='ignore'):
            self.logp_geweke_ = _geweke(self.logps_, n_burn)
            self.lambda_geweke_ = _geweke(self.lambdas_[:, 0], n_burn)
            if self.is_directed:
                self.intercept_in_geweke_ = _geweke(self.intercepts_[:, 0], n_burn)
                self.intercept_out_geweke_ = _geweke(self.intercepts_[:, 1], n_burn)
            else:
                self.intercept_geweke_ = _geweke(self.intercepts_[:, 0], n_burn)
        return self
