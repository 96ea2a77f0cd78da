"""DynamicNetworkLSM with the reference's constructor, ``fit(Y)`` and fitted
attributes (lsm.py:100-625), running the Gibbs loop on one MI355X.

Host code is orchestration only.  For the undirected model the whole loop
(lsm.py:474-572) is device resident: sweep, Procrustes, centring, intercept MH
fused with the log-posterior trace, samples stored in a device trace that is
read back once.  Directed models (exact and case-control) drive the same
kernels from the host because the radii step draws a Dirichlet proposal with
the numpy stream (metropolis.py:57-82).

Differences from the reference that a user can see:
  * random numbers: the latent-position sweep uses the engine's Philox streams
    (keyed by a seed drawn from ``random_state``) and the even/odd-t scan order,
    so chains are equal in distribution, not sample for sample;
  * missing edges (-1 coded dyads) are imputed once before the chain, as the reference
    effectively does (see ``imputer``); NaN entries are rejected;
  * ``fit(Y, init=...)`` accepts starting values and skips the init pipeline.
"""
import time

import numpy as np
from scipy.special import gammaln, xlogy

from .engine import Chain, SamplerGrid, check_n_features
from . import initialization as init_mod
from .imputer import SimpleNetworkImputer
from .metrics import FittedQuantities

__all__ = ['DynamicNetworkLSM']


def check_random_state(seed):
    if seed is None or seed is np.random:
        return np.random.mtrand._rand
    if isinstance(seed, (int, np.integer)):
        return np.random.RandomState(seed)
    if isinstance(seed, np.random.RandomState):
        return seed
    raise ValueError('%r cannot be used to seed a numpy.random.RandomState' % seed)


class _ScalarMetropolis(object):
    """Host mirror of metropolis.py:85-136 for the scalar / Dirichlet blocks
    the directed models update from the host."""

    def __init__(self, step_size, tune, tune_interval=100, dirichlet=False):
        self.step_size, self.tune, self.tune_interval = step_size, tune, tune_interval
        self.steps_until_tune = tune_interval
        self.n_accepted = 0
        self.n_steps = 0
        self.dirichlet = dirichlet

    def book(self, accepted):
        self.n_accepted += accepted
        self.n_steps += 1
        if self.tune is not None:
            if self.n_steps < self.tune and self.steps_until_tune == 0:
                r = self.n_accepted / self.tune_interval
                s = self.step_size
                if self.dirichlet:      # metropolis.py:23-37
                    s *= (10.0 if r < 0.001 else 2 if r < 0.05 else 1.1 if r < 0.25 else
                          0.1 if r > 0.95 else 0.5 if r > 0.75 else 0.9 if r > 0.4 else 1)
                else:                   # metropolis.py:5-20
                    s *= (0.1 if r < 0.001 else 0.5 if r < 0.05 else 0.9 if r < 0.25 else
                          10.0 if r > 0.95 else 2.0 if r > 0.75 else 1.1 if r > 0.4 else 1)
                self.step_size = s
                self.n_accepted = 0
                self.steps_until_tune = self.tune_interval
            else:
                self.steps_until_tune -= 1


def latent_prior_terms(X, tau_sq, sigma_sq):
    """lsm.py:604-613"""
    diff = X[1:] - X[:-1]
    return -(0.5 * np.sum(X[0] * X[0]) / tau_sq + 0.5 * np.sum(diff * diff) / sigma_sq)


def _dirichlet_logpdf(x, alpha):
    """scipy.stats.dirichlet.logpdf(x, alpha) without its input checks (the proposal
    density ratio of metropolis.py:70-76)"""
    return gammaln(np.sum(alpha)) - np.sum(gammaln(alpha)) + np.sum(xlogy(alpha - 1.0, x))


class DynamicNetworkLSM(FittedQuantities):
    """Latent space model for dynamic networks (Sewell & Chen) on MI355X.

    Constructor parameters are the reference's (lsm.py:234-268) plus
    ``device`` (GPU index), ``chain_id`` (Philox stream of this chain) and
    ``sweep_algo`` (0 auto; 1 .. 5 as ``dlsm_sweep_positions`` in include/dynetlsm_hip.h)."""

    def __init__(self, n_features=2, is_directed=False, n_iter=5000, tune=2500,
                 tune_interval=100, burn=2500, intercept_prior='auto',
                 intercept_variance_prior=2.0, tau_sq=2.0, sigma_sq=0.1, step_size_X=0.1,
                 step_size_intercept=0.1, step_size_radii=175000, n_control=None,
                 n_resample_control=100, copy=True, random_state=None, device=0,
                 chain_id=0, sweep_algo=0, directed_loop='device'):
        self.directed_loop = directed_loop
        self.n_iter = n_iter
        self.is_directed = is_directed
        self.n_features = n_features
        self.tau_sq = tau_sq
        self.sigma_sq = sigma_sq
        self.step_size_X = step_size_X
        self.intercept_prior = intercept_prior
        self.intercept_variance_prior = intercept_variance_prior
        self.step_size_intercept = step_size_intercept
        self.step_size_radii = step_size_radii
        self.tune = tune
        self.tune_interval = tune_interval
        self.burn = burn
        self.n_control = n_control
        self.n_resample_control = n_resample_control
        self.copy = copy
        self.random_state = random_state
        self.device = device
        self.chain_id = chain_id
        self.sweep_algo = sweep_algo

    @property
    def n_burn_(self):
        return (self.burn or 0) + (self.tune or 0)

    # -- fit ------------------------------------------------------------------
    def fit(self, Y, init=None):
        """Sample the posterior given the dynamic network ``Y`` (T, N, N),
        float64 binary adjacency matrices.  ``init`` may give starting values
        ``dict(X=..., intercept=..., radii=...)``; otherwise they come from the
        GMDS / conditional-MLE pipeline as in the reference (lsm.py:386-407)."""
        Y = np.array(Y, dtype=np.float64, copy=self.copy, order='C')
        if Y.ndim != 3 or Y.shape[1] != Y.shape[2]:
            raise ValueError('Y must have shape (n_time_steps, n_nodes, n_nodes)')
        if np.any(np.isnan(Y)):
            raise ValueError('NaN entries are not supported: code missing dyads as -1')
        if np.any(Y == -1):          # lsm.py:345-359
            Y = SimpleNetworkImputer(strategy='random', missing_value=-1).fit_transform(Y)
        T, N, _ = Y.shape
        D = check_n_features(self.n_features)
        rng = check_random_state(self.random_state)
        self.Y_fit_ = Y
        if self.n_control is not None and not self.is_directed:
            raise ValueError('The case-control likelihood currently only '
                             'supported for directed networks.')

        n_iter_procrustes = 0
        if self.tune is not None:
            self.n_iter += self.tune       # the reference mutates n_iter too (lsm.py:362-368)
            n_iter_procrustes += self.tune
        if self.burn is not None:
            self.n_iter += self.burn
            n_iter_procrustes += self.burn
        n_total = self.n_iter

        model = ('undirected' if not self.is_directed else
                 'case_control' if self.n_control is not None else 'directed')
        exact_model = 'directed' if self.is_directed else 'undirected'
        seed = int(rng.randint(0, 2 ** 31 - 1)) | (int(rng.randint(0, 2 ** 31 - 1)) << 31)

        # ---- starting values ------------------------------------------------
        radii = None
        if init is not None:
            X = np.array(init['X'], dtype=np.float64)
            intercept = np.atleast_1d(np.asarray(init['intercept'], dtype=np.float64)).copy()
            if self.is_directed:
                radii = np.array(init['radii'], dtype=np.float64)
        else:
            with Chain(T, N, D, exact_model, device=self.device) as c0:
                c0.upload_network(Y)
                X = init_mod.generalized_mds(c0, is_directed=self.is_directed,
                                             random_state=rng)
                if self.is_directed:
                    radii = init_mod.initialize_radii(Y)
                    b_in, b_out = init_mod.directed_intercept_mle(c0, X, radii)
                    intercept = np.array([b_in, b_out])
                else:
                    scale, b = init_mod.scale_intercept_mle(c0, X)
                    intercept = np.array([b])
                    X = X * np.exp(scale)
        X = X - np.mean(X, axis=(0, 1))
        if isinstance(self.tau_sq, str) and self.tau_sq == 'auto':
            self.tau_sq = np.mean(X[0] * X[0])
        if isinstance(self.intercept_prior, str) and self.intercept_prior == 'auto':
            self.intercept_prior = intercept.copy()
        ip = np.atleast_1d(np.asarray(self.intercept_prior, dtype=np.float64))

        # ---- the chain ------------------------------------------------------
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
        chain.set_prior_random_walk(self.tau_sq, self.sigma_sq)
        self.latent_samplers = SamplerGrid(T, N, self.step_size_X, tune=self.tune,
                                           tune_interval=self.tune_interval)
        chain.set_samplers(self.latent_samplers)
        ll0 = chain.loglik_full()
        logp0 = self._log_prior(X, intercept, ip) + ll0

        t_loop = time.perf_counter()
        if not self.is_directed:
            self._fit_undirected(chain, n_total, n_iter_procrustes, logp0, ip)
        elif self.directed_loop == 'device':
            self._fit_directed_device(chain, n_total, n_iter_procrustes, logp0, ip)
        else:
            self._fit_directed(chain, rng, X, intercept, radii, n_total, n_iter_procrustes,
                               logp0, ip)
        chain.get_samplers(self.latent_samplers)
        self.loop_seconds_ = time.perf_counter() - t_loop     # Gibbs loop only
        self._set_map(n_total)
        return self

    def _log_prior(self, X, intercept, ip):
        """lsm.py:604-623"""
        lp = latent_prior_terms(X, self.tau_sq, self.sigma_sq)
        diff = intercept - ip
        return lp - np.sum(0.5 * (diff * diff) / self.intercept_variance_prior)

    def _fit_undirected(self, chain, n_total, n_iter_procrustes, logp0, ip):
        # lsm.py:465-467: the undirected intercept sampler ignores tune_interval
        chain.lsm_configure(ip, self.intercept_variance_prior,
                            step_size_intercept=self.step_size_intercept, tune=self.tune,
                            tune_interval=100, n_iter_procrustes=n_iter_procrustes,
                            sweep_algo=self.sweep_algo)
        chain.trace_alloc(n_total, logp0=float(logp0))
        first = min(n_iter_procrustes, n_total - 1)
        if first > 0:
            chain.lsm_run(1, first)
        if n_total - 1 > first:
            _, _, lps = chain.trace_read(0, n_iter_procrustes + 1, positions=False)
            prev_map = int(np.argmax(lps))          # lsm.py:496
            chain.lsm_run(first + 1, n_total - 1 - first, procrustes_ref=prev_map)
        self.Xs_, self.intercepts_, self.logps_ = chain.trace_read(0, n_total)
        cfg = chain.lsm_get_config()
        self.intercept_samplers = [_ScalarMetropolis(cfg.i_step_size[0], self.tune)]
        self.intercept_samplers[0].n_accepted = cfg.i_n_accepted[0]
        self.intercept_samplers[0].n_steps = cfg.i_n_steps[0]
        self.intercept_samplers[0].steps_until_tune = cfg.i_steps_until_tune[0]

    def _fit_directed_device(self, chain, n_total, n_iter_procrustes, logp0, ip):
        """lsm.py:474-572 for the directed models, iterations enqueued on the device
        (``dlsm_lsm_run``): sweep, Procrustes / centring, the two intercept steps and the radii
        step with Philox draws.  The host only keeps the cadence of the control resampling
        (case_control_likelihood.py:27-33) and picks the Procrustes reference (lsm.py:496)."""
        chain.lsm_configure(ip, self.intercept_variance_prior,
                            step_size_intercept=self.step_size_intercept, tune=self.tune,
                            tune_interval=self.tune_interval, n_iter_procrustes=n_iter_procrustes,
                            sweep_algo=self.sweep_algo, step_size_radii=self.step_size_radii,
                            radii_tune=None)
        chain.trace_alloc(n_total, logp0=float(logp0))
        ccs = self.case_control_sampler_
        prev_map = -1

        def run(first, last):                     # iterations first .. last
            it = first
            while it <= last:
                if ccs is not None:
                    ccs.resample(it)
                    nxt = it + 1                  # next iteration that resamples
                    # n_resample_control=None: never resample (case_control_likelihood.py:28)
                    while nxt <= last and (ccs.n_resample is None or
                                           ccs.n_iter % ccs.n_resample != 0):
                        ccs.n_iter += 1
                        nxt += 1
                    chain.lsm_run(it, nxt - it, procrustes_ref=prev_map)
                    it = nxt
                else:
                    chain.lsm_run(it, last - it + 1, procrustes_ref=prev_map)
                    it = last + 1
        first = min(n_iter_procrustes, n_total - 1)
        if first > 0:
            run(1, first)
        if n_total - 1 > first:
            _, _, lps = chain.trace_read(0, n_iter_procrustes + 1, positions=False)
            prev_map = int(np.argmax(lps))          # lsm.py:496
            run(first + 1, n_total - 1)
        self.Xs_, self.intercepts_, self.logps_ = chain.trace_read(0, n_total)
        self.radiis_ = chain.trace_read_radii(0, n_total)
        cfg = chain.lsm_get_config()
        self.intercept_samplers = []
        for k in range(2):
            sm = _ScalarMetropolis(cfg.i_step_size[k], self.tune, self.tune_interval)
            sm.n_accepted, sm.n_steps = cfg.i_n_accepted[k], cfg.i_n_steps[k]
            sm.steps_until_tune = cfg.i_steps_until_tune[k]
            self.intercept_samplers.append(sm)
        self.radii_sampler = _ScalarMetropolis(cfg.r_step_size, None, dirichlet=True)
        self.radii_sampler.n_accepted, self.radii_sampler.n_steps = (cfg.r_n_accepted,
                                                                     cfg.r_n_steps)

    def _fit_directed(self, chain, rng, X, intercept, radii, n_total, n_iter_procrustes,
                      logp0, ip):
        T, N, D = X.shape
        self.Xs_ = np.zeros((n_total, T, N, D))
        self.intercepts_ = np.zeros((n_total, 2))
        self.radiis_ = np.zeros((n_total, N))
        self.logps_ = np.zeros(n_total)
        self.Xs_[0], self.intercepts_[0], self.radiis_[0], self.logps_[0] = (
            X, intercept, radii, logp0)
        isamp = [_ScalarMetropolis(self.step_size_intercept, self.tune, self.tune_interval)
                 for _ in range(2)]
        rsamp = _ScalarMetropolis(self.step_size_radii, None, dirichlet=True)
        self.intercept_samplers, self.radii_sampler = isamp, rsamp
        var = self.intercept_variance_prior
        intercept = intercept.copy()
        radii = radii.copy()
        for it in range(1, n_total):
            if self.case_control_sampler_ is not None:
                self.case_control_sampler_.resample(it)
            chain.sweep_positions(it, self.sweep_algo)
            if it > n_iter_procrustes:
                prev_map = int(np.argmax(self.logps_[:n_iter_procrustes + 1]))
                chain.procrustes(self.Xs_[prev_map])
            chain.center()
            # sample_coefficients.py:18-75 : two scalar RW-MH steps, each a fused
            # two-candidate pass
            for k in range(2):
                prop = intercept.copy()
                prop[k] = intercept[k] + isamp[k].step_size * rng.randn(1)[0]
                if k == 0:
                    ll_prop, ll_cur = chain.loglik_full([prop, intercept])
                else:       # same positions, and `intercept` is the state the last step left
                    ll_prop, ll_cur = chain.loglik_full([prop])[0], ll_state
                ratio = ((ll_prop - (prop[k] - ip[k]) ** 2 / (2 * var)) -
                         (ll_cur - (intercept[k] - ip[k]) ** 2 / (2 * var)))
                accepted = int(not (np.log(rng.rand()) >= ratio))
                ll_state = ll_cur
                if accepted:
                    intercept, ll_state = prop, ll_prop
                isamp[k].book(accepted)
            chain.set_intercepts(intercept)
            # sample_coefficients.py:91-121 + metropolis.py:57-82
            x = rng.dirichlet(rsamp.step_size * radii)
            if np.any(x == 0.):
                x += 1e-5
                x /= np.sum(x)
            ll_cur, ll_prop = chain.loglik_full_radii(x)
            ratio = ll_prop - ll_cur
            ratio += (_dirichlet_logpdf(radii, rsamp.step_size * x) -
                      _dirichlet_logpdf(x, rsamp.step_size * radii))
            accepted = int(not (np.log(rng.rand()) >= ratio))
            ll = ll_cur
            if accepted:
                radii, ll = x, ll_prop
                chain.set_radii(radii)
            rsamp.book(accepted)
            Xc = chain.get_positions()
            self.Xs_[it], self.intercepts_[it], self.radiis_[it] = Xc, intercept, radii
            self.logps_[it] = ll + self._log_prior(Xc, intercept, ip)

    def _set_map(self, n_total):
        """MAP bookkeeping of lsm.py:554-566, replayed over the trace."""
        best = 0
        logp = self.logps_[0]
        for it in range(1, n_total):
            if self.tune and it == (self.tune + (self.burn or 0)):
                best, logp = it, self.logps_[it]
            elif self.logps_[it] > logp:
                best, logp = it, self.logps_[it]
        self.logp_ = logp
        self.X_ = self.Xs_[best]
        self.intercept_ = self.intercepts_[best]
        if self.is_directed:
            self.radii_ = self.radiis_[best]
        self.map_index_ = best

    def logp(self, Y, X, intercept, radii=None, dist=None):
        """lsm.py:576-625 for arbitrary arguments (one GPU pass)."""
        model = ('undirected' if not self.is_directed else 'directed')
        T, N, D = X.shape
        with Chain(T, N, D, model, device=self.device) as c:
            c.upload_network(np.ascontiguousarray(Y, dtype=np.float64))
            c.set_positions(X)
            if self.is_directed:
                c.set_radii(radii)
            ic = np.atleast_1d(np.asarray(intercept, dtype=np.float64))
            ll = c.loglik_full([ic])[0]
        ip = np.atleast_1d(np.asarray(self.intercept_prior, dtype=np.float64))
        return ll + self._log_prior(X, ic, ip)
