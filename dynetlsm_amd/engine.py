"""Python face of one device-resident MCMC chain (one ``dlsm_chain`` handle).

Thin: argument checking, dtype/contiguity normalisation and error mapping.
All arithmetic happens in the HIP library; nothing here computes a likelihood.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (EngineError, LsmConfig, HdpConfig, c_double_p, c_i32_p, c_i64_p,
                   UNDIRECTED, DIRECTED, DIRECTED_CASE_CONTROL)

__all__ = ['Chain', 'SamplerGrid', 'EngineError', 'MAX_FEATURES', 'check_n_features']

# The kernels take the latent dimension as a template parameter, instantiated for 1..8 (csrc/device_common.hpp
# DLSM_D_MAX; the reference takes any n_features, lsm.py:235,254; its examples and the paper use 2).
MAX_FEATURES = 8


def check_n_features(n_features):
    """ValueError naming the limit, before any device call (capi.hip's DISPATCH_D would answer
    DLSM_E_LIMIT from the first kernel launch only)"""
    d = int(n_features)
    if d != n_features or not 1 <= d <= MAX_FEATURES:
        raise ValueError('n_features=%r is not supported by the MI355X engine: its kernels are '
                         'compiled for 1 <= n_features <= %d latent dimensions' % (n_features, MAX_FEATURES))
    return d


def _f64(a, shape=None, name='array'):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError('%s has shape %s, expected %s' % (name, a.shape, shape))
    return a


def _i64(a, shape=None, name='array'):
    a = np.ascontiguousarray(a, dtype=np.int64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError('%s has shape %s, expected %s' % (name, a.shape, shape))
    return a


def _i32(a, shape=None, name='array'):
    a = np.ascontiguousarray(a, dtype=np.int32)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError('%s has shape %s, expected %s' % (name, a.shape, shape))
    return a


def _p(a):
    if a.dtype == np.float64:
        return a.ctypes.data_as(c_double_p)
    if a.dtype == np.int64:
        return a.ctypes.data_as(c_i64_p)
    if a.dtype == np.int32:
        return a.ctypes.data_as(c_i32_p)
    raise TypeError(a.dtype)


class SamplerGrid(object):
    """State of the T x N random-walk Metropolis samplers (one per (t, node)),
    the struct-of-arrays form of the reference's ``latent_samplers`` list of
    ``Metropolis`` objects (metropolis.py:85-94, lsm.py:451-457)."""

    def __init__(self, T, N, step_size=0.1, tune=500, tune_interval=100):
        self.step_size = np.full((T, N), float(step_size))
        self.n_accepted = np.zeros((T, N), dtype=np.int32)
        self.n_steps = np.zeros((T, N), dtype=np.int32)
        self.steps_until_tune = np.full((T, N), tune_interval, dtype=np.int32)
        self.tune = tune
        self.tune_interval = tune_interval

    @classmethod
    def from_objects(cls, samplers):
        """Build from a list (T) of lists (N) of Metropolis-like objects."""
        T, N = len(samplers), len(samplers[0])
        s0 = samplers[0][0]
        g = cls(T, N, s0.step_size, s0.tune, s0.tune_interval)
        for t in range(T):
            for j in range(N):
                s = samplers[t][j]
                g.step_size[t, j] = s.step_size
                g.n_accepted[t, j] = s.n_accepted
                g.n_steps[t, j] = s.n_steps
                g.steps_until_tune[t, j] = s.steps_until_tune
        return g

    def to_objects(self, samplers):
        for t in range(len(samplers)):
            for j in range(len(samplers[0])):
                s = samplers[t][j]
                s.step_size = float(self.step_size[t, j])
                s.n_accepted = int(self.n_accepted[t, j])
                s.n_steps = int(self.n_steps[t, j])
                s.steps_until_tune = int(self.steps_until_tune[t, j])


class Chain(object):
    """One chain on one MI355X.

    Parameters mirror ``dlsm_create``: ``model`` is 'undirected', 'directed' or
    'case_control'; (seed, chain_id) key the Philox streams.
    """
    MODELS = {'undirected': UNDIRECTED, 'directed': DIRECTED,
              'case_control': DIRECTED_CASE_CONTROL}

    def __init__(self, T, N, D=2, model='undirected', seed=0, chain_id=0, device=0):
        check_n_features(D)
        self._L = _lib.load()
        self._h = _lib.handle_t()
        self.T, self.N, self.D = int(T), int(N), int(D)
        self.model = self.MODELS[model] if isinstance(model, str) else int(model)
        self.n_intercepts = 1 if self.model == UNDIRECTED else 2
        rc = self._L.dlsm_create(int(device), self.T, self.N, self.D, self.model,
                                 C.c_uint64(int(seed) & (2 ** 64 - 1)), int(chain_id),
                                 C.byref(self._h))
        if rc != 0:
            msg = self._L.dlsm_last_error(None)
            self._h = None
            raise EngineError(rc, msg.decode() if msg else 'dlsm_create failed')
        self.K = 0
        self.C = 0
        self.seed, self.chain_id = int(seed) & (2 ** 64 - 1), int(chain_id)    # the Philox key

    # -- plumbing ---------------------------------------------------------
    def _ck(self, rc):
        if rc != 0:
            msg = self._L.dlsm_last_error(self._h)
            raise EngineError(rc, msg.decode() if msg else '?')

    def close(self):
        if getattr(self, '_h', None):
            self._L.dlsm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def synchronize(self):
        self._ck(self._L.dlsm_synchronize(self._h))

    # -- network -----------------------------------------------------------
    def upload_network(self, Y):
        Y = _f64(Y, (self.T, self.N, self.N), 'Y')
        self._ck(self._L.dlsm_upload_network(self._h, _p(Y)))

    def network_packed_words(self):
        """uint32 words of the bit-packed network as the chain holds it"""
        n = C.c_int64(0)
        self._ck(self._L.dlsm_network_packed_words(self._h, C.byref(n)))
        return int(n.value)

    def get_network_packed(self, ptr, n_words):
        """copy the packed network into ``ptr`` (device or host address, e.g.
        ``tensor.data_ptr()`` of an int32 tensor of ``n_words`` elements)"""
        self._ck(self._L.dlsm_get_network_packed(self._h, C.c_void_p(int(ptr)), int(n_words)))

    def set_network_packed(self, ptr, n_words):
        """take the packed network from ``ptr`` (as written by ``get_network_packed`` of a
        chain of the same shape and model); the layout's invariants are checked"""
        self._ck(self._L.dlsm_set_network_packed(self._h, C.c_void_p(int(ptr)), int(n_words)))

    def upload_edges(self, in_edges, out_edges, degree):
        ie = _i64(in_edges); oe = _i64(out_edges)
        dg = _i64(degree, (self.T, self.N, 2), 'degree')
        if ie.shape[:2] != (self.T, self.N) or oe.shape[:2] != (self.T, self.N):
            raise ValueError('edge lists must be (T, N, max_degree)')
        self._ck(self._L.dlsm_upload_edges(self._h, _p(ie), ie.shape[2], _p(oe),
                                           oe.shape[2], _p(dg)))

    def set_controls(self, control_nodes_in, control_nodes_out):
        ci = _i64(control_nodes_in); co = _i64(control_nodes_out)
        if ci.shape != co.shape or ci.shape[:2] != (self.T, self.N):
            raise ValueError('control node arrays must both be (T, N, n_control)')
        self.C = ci.shape[2]
        self._ck(self._L.dlsm_set_controls(self._h, _p(ci), _p(co), self.C))

    def get_controls(self):
        ci = np.zeros((self.T, self.N, self.C), dtype=np.int64)
        co = np.zeros_like(ci)
        self._ck(self._L.dlsm_get_controls(self._h, _p(ci), _p(co)))
        return ci, co

    def resample_controls(self, it, n_control):
        self._ck(self._L.dlsm_resample_controls(self._h, int(it), int(n_control)))
        self.C = int(n_control)

    # -- state -------------------------------------------------------------
    def set_positions(self, X):
        X = _f64(X, (self.T, self.N, self.D), 'X')
        self._ck(self._L.dlsm_set_positions(self._h, _p(X)))

    def get_positions(self):
        X = np.zeros((self.T, self.N, self.D))
        self._ck(self._L.dlsm_get_positions(self._h, _p(X)))
        return X

    def set_intercepts(self, b):
        b = _f64(np.atleast_1d(b).ravel(), (self.n_intercepts,), 'intercept')
        self._ck(self._L.dlsm_set_intercepts(self._h, _p(b), self.n_intercepts))

    def get_intercepts(self):
        b = np.zeros(self.n_intercepts)
        self._ck(self._L.dlsm_get_intercepts(self._h, _p(b), self.n_intercepts))
        return b

    def set_radii(self, radii):
        r = _f64(radii, (self.N,), 'radii')
        self._ck(self._L.dlsm_set_radii(self._h, _p(r)))

    def get_radii(self):
        r = np.zeros(self.N)
        self._ck(self._L.dlsm_get_radii(self._h, _p(r)))
        return r

    def set_squared(self, squared):
        self._ck(self._L.dlsm_set_squared(self._h, int(bool(squared))))

    def set_samplers(self, grid):
        sh = (self.T, self.N)
        st = _f64(grid.step_size, sh, 'step_size')
        na = _i32(grid.n_accepted, sh); ns = _i32(grid.n_steps, sh)
        un = _i32(grid.steps_until_tune, sh)
        tune = -1 if grid.tune is None else int(grid.tune)
        self._ck(self._L.dlsm_set_samplers(self._h, _p(st), _p(na), _p(ns), _p(un),
                                           tune, int(grid.tune_interval)))

    def get_samplers(self, grid):
        """read the device's sampler state back into ``grid`` (in place)"""
        sh = (self.T, self.N)
        st = np.zeros(sh); na = np.zeros(sh, dtype=np.int32)
        ns = np.zeros(sh, dtype=np.int32); un = np.zeros(sh, dtype=np.int32)
        self._ck(self._L.dlsm_get_samplers(self._h, _p(st), _p(na), _p(ns), _p(un)))
        grid.step_size[...] = st; grid.n_accepted[...] = na
        grid.n_steps[...] = ns; grid.steps_until_tune[...] = un
        return grid

    def set_prior_random_walk(self, tau_sq, sigma_sq):
        self._ck(self._L.dlsm_set_prior_random_walk(self._h, float(tau_sq),
                                                    float(sigma_sq)))

    def set_prior_mixture(self, mu, sigma, lmbda, z):
        """``z=None`` keeps the labels the device already holds (those of the last
        ``sample_labels`` or ``set_prior_mixture``)."""
        sigma = _f64(sigma)
        K = sigma.shape[0]
        mu = _f64(mu, (K, self.D), 'mu')
        zp = None if z is None else _p(_i64(z, (self.T, self.N), 'z'))
        lm = float(np.asarray(lmbda).ravel()[0])
        self._ck(self._L.dlsm_set_prior_mixture(self._h, _p(mu), _p(sigma), lm, zp, K))
        self.K = K

    # -- kernels -----------------------------------------------------------
    def loglik_full(self, intercepts=None):
        """network log-likelihood at the current X; ``intercepts`` (m, n_ic)
        evaluates m candidates in fused passes, None the current intercept."""
        if intercepts is None:
            out = np.zeros(1)
            self._ck(self._L.dlsm_loglik_full(self._h, 1, None, _p(out)))
            return float(out[0])
        ic = _f64(np.atleast_2d(intercepts))
        if ic.shape[1] != self.n_intercepts:
            raise ValueError('intercepts must be (m, %d)' % self.n_intercepts)
        out = np.zeros(ic.shape[0])
        self._ck(self._L.dlsm_loglik_full(self._h, ic.shape[0], _p(ic), _p(out)))
        return out

    def loglik_full_radii(self, radii_alt):
        r = _f64(radii_alt, (self.N,), 'radii_alt')
        out = np.zeros(2)
        self._ck(self._L.dlsm_loglik_full_radii(self._h, _p(r), _p(out)))
        return out

    def loglik_partial(self, t, j, x=None, with_prior=False):
        out = np.zeros(1)
        xp = None
        if x is not None:
            x = _f64(x, (self.D,), 'x')
            xp = _p(x)
        self._ck(self._L.dlsm_loglik_partial(self._h, int(t), int(j), xp,
                                             int(with_prior), _p(out)))
        return float(out[0])

    def loglik_partial_all(self, with_prior=False):
        out = np.zeros((self.T, self.N))
        self._ck(self._L.dlsm_loglik_partial_all(self._h, int(with_prior), _p(out)))
        return out

    def sweep_positions(self, it, algo=0):
        self._ck(self._L.dlsm_sweep_positions(self._h, int(it), int(algo)))

    def resolve_sweep_algo(self, algo=0):
        """the sweep algorithm ``algo`` resolves to on this chain (0 = auto)"""
        rc = self._L.dlsm_resolve_sweep_algo(self._h, int(algo))
        if rc < 0:
            self._ck(rc)
        return rc

    def center(self):
        self._ck(self._L.dlsm_center(self._h))

    def procrustes(self, X_ref):
        X_ref = _f64(X_ref, (self.T, self.N, self.D), 'X_ref')
        R = np.zeros((self.D, self.D))
        self._ck(self._L.dlsm_procrustes(self._h, _p(X_ref), _p(R)))
        return R

    def gaussian_likelihood(self, node, normalize=True):
        out = np.zeros((self.T, self.K))
        self._ck(self._L.dlsm_gaussian_likelihood(self._h, int(node), int(normalize),
                                                  _p(out)))
        return out

    def sample_labels(self, it, w):
        K = self.K
        w = _f64(w, (self.T, K, K), 'w')
        z = np.zeros((self.T, self.N), dtype=np.int64)
        n = np.zeros((self.T, K, K))
        nk = np.zeros((self.T, K), dtype=np.int64)
        self._ck(self._L.dlsm_sample_labels(self._h, int(it), _p(w), _p(z), _p(n), _p(nk)))
        return z, n, nk

    # -- device-resident LSM loop -------------------------------------------
    def hdp_label_sums(self, stage, mu=None, sigma=None, lmbda=0.0, w=None, a=0.0, b=0.0):
        """Label-wise sums of the HDP-LPCM conjugate updates on the device (stage 0 means,
        1 residuals, 2 lambda, 3 log-posterior node terms); see ``dlsm_hdp_label_sums``."""
        K = self.K
        nv = self.D if stage == 0 else (2 if stage == 2 else 1)
        out = np.empty((self.T, K, nv) if nv > 1 else (self.T, K))
        mu = None if mu is None else _f64(mu, (K, self.D), 'mu')
        sigma = None if sigma is None else _f64(sigma, (K,), 'sigma')
        w = None if w is None else _f64(w, (self.T, K, K), 'w')
        self._ck(self._L.dlsm_hdp_label_sums(
            self._h, int(stage), None if mu is None else _p(mu),
            None if sigma is None else _p(sigma), float(np.ravel(lmbda)[0]),
            None if w is None else _p(w), float(a), float(b), _p(out)))
        return out

    def lsm_configure(self, intercept_prior, intercept_variance_prior,
                      step_size_intercept=0.1, tune=None, tune_interval=100,
                      n_iter_procrustes=0, sweep_algo=0, state=None, step_size_radii=175000.,
                      radii_tune=None, radii_tune_interval=100):
        cfg = LsmConfig()
        ip = np.atleast_1d(np.asarray(intercept_prior, dtype=np.float64)).ravel()
        for k in range(2):
            cfg.intercept_prior[k] = ip[k] if k < ip.size else 0.0
            cfg.i_step_size[k] = float(step_size_intercept)
            cfg.i_n_accepted[k] = 0
            cfg.i_n_steps[k] = 0
            cfg.i_steps_until_tune[k] = int(tune_interval)
        if state is not None:       # (step, n_accepted, n_steps, until) per sampler
            for k, s in enumerate(state):
                cfg.i_step_size[k], cfg.i_n_accepted[k] = s[0], s[1]
                cfg.i_n_steps[k], cfg.i_steps_until_tune[k] = s[2], s[3]
        cfg.intercept_variance_prior = float(intercept_variance_prior)
        cfg.i_tune = -1 if tune is None else int(tune)
        cfg.i_tune_interval = int(tune_interval)
        cfg.n_iter_procrustes = int(n_iter_procrustes)
        cfg.sweep_algo = int(sweep_algo)
        cfg.r_step_size = float(step_size_radii)
        cfg.r_n_accepted, cfg.r_n_steps = 0, 0
        cfg.r_steps_until_tune = int(radii_tune_interval)
        cfg.r_tune = -1 if radii_tune is None else int(radii_tune)
        cfg.r_tune_interval = int(radii_tune_interval)
        self._ck(self._L.dlsm_lsm_configure(self._h, C.byref(cfg)))

    def lsm_get_config(self):
        cfg = LsmConfig()
        self._ck(self._L.dlsm_lsm_get_config(self._h, C.byref(cfg)))
        return cfg

    def trace_alloc(self, n_total, logp0=0.0):
        self._ck(self._L.dlsm_trace_alloc(self._h, int(n_total), float(logp0)))
        self._trace_n = int(n_total)

    def lsm_run(self, first, count, procrustes_ref=-1):
        """enqueue iterations first..first+count-1 (asynchronous)"""
        self._ck(self._L.dlsm_lsm_run(self._h, int(first), int(count),
                                      int(procrustes_ref)))

    def trace_read(self, first, count, positions=True):
        Xs = np.zeros((count, self.T, self.N, self.D)) if positions else None
        ics = np.zeros((count, 2))
        lps = np.zeros(count)
        self._ck(self._L.dlsm_trace_read(self._h, int(first), int(count),
                                         _p(Xs) if positions else None, _p(ics),
                                         _p(lps)))
        return Xs, ics[:, :self.n_intercepts], lps

    # -- device-resident HDP-LPCM loop -------------------------
    def hdp_configure(self, hp, beta, weights, intercept_prior, intercept_variance_prior,
                      step_size_intercept=0.1, tune=None, tune_interval=100, sweep_algo=0,
                      state=None, step_size_radii=175000., radii_tune=None, radii_tune_interval=100):
        """``hp``: an object with the HDP-LPCM's hyper-parameters as attributes (gamma,
        alpha_init, alpha, kappa, mean_variance_prior, b, a, a0, b0, c0, d0 - None switches an
        update off -, lambda_prior, lambda_variance_prior, *_prior_shape / *_rate); beta (K,),
        weights (T, K, K).  The mixture prior (mu, sigma, lmbda, z) must be set."""
        K = self.K
        cfg = HdpConfig()
        for name in ('gamma', 'alpha_init', 'alpha', 'kappa', 'mean_variance_prior', 'b', 'a',
                     'lambda_prior', 'lambda_variance_prior', 'gamma_prior_shape',
                     'gamma_prior_rate', 'alpha_init_shape', 'alpha_init_rate',
                     'alpha_kappa_shape', 'alpha_kappa_rate'):
            setattr(cfg, name, float(np.ravel(getattr(hp, name))[0]))
        cfg.has_a0 = int(hp.a0 is not None)
        cfg.has_c0 = int(hp.c0 is not None)
        cfg.a0, cfg.b0 = (float(hp.a0), float(hp.b0)) if hp.a0 is not None else (0.0, 0.0)
        cfg.c0, cfg.d0 = (float(hp.c0), float(hp.d0)) if hp.c0 is not None else (0.0, 0.0)
        cfg.intercept_prior = float(np.ravel(intercept_prior)[0])
        cfg.intercept_variance_prior = float(intercept_variance_prior)
        cfg.i_step_size = float(step_size_intercept)
        cfg.i_n_accepted, cfg.i_n_steps = 0, 0
        cfg.i_steps_until_tune = int(tune_interval)
        if state is not None:       # (step, n_accepted, n_steps, until)
            cfg.i_step_size, cfg.i_n_accepted = float(state[0]), int(state[1])
            cfg.i_n_steps, cfg.i_steps_until_tune = int(state[2]), int(state[3])
        cfg.i_tune = -1 if tune is None else int(tune)
        cfg.i_tune_interval = int(tune_interval)
        cfg.sweep_algo = int(sweep_algo)
        if self.model != UNDIRECTED:        # intercept_out and the radii sampler (hdp_lpcm.py:731-747)
            ip = np.ravel(np.asarray(intercept_prior, dtype=np.float64))
            cfg.intercept_prior_out = float(ip[1] if ip.size > 1 else ip[0])
            cfg.i_step_size_out = float(step_size_intercept)
            cfg.i_n_accepted_out, cfg.i_n_steps_out = 0, 0
            cfg.i_steps_until_tune_out = int(tune_interval)
            cfg.r_step_size = float(step_size_radii)
            cfg.r_n_accepted, cfg.r_n_steps = 0, 0
            cfg.r_steps_until_tune = int(radii_tune_interval)
            cfg.r_tune = -1 if radii_tune is None else int(radii_tune)
            cfg.r_tune_interval = int(radii_tune_interval)
        beta = _f64(beta, (K,), 'beta')
        weights = _f64(weights, (self.T, K, K), 'weights')
        self._ck(self._L.dlsm_hdp_configure(self._h, C.byref(cfg), _p(beta), _p(weights)))

    def hdp_get_config(self):
        cfg = HdpConfig()
        self._ck(self._L.dlsm_hdp_get_config(self._h, C.byref(cfg)))
        return cfg

    def hdp_trace_alloc(self, n_total, logp0=0.0):
        self._ck(self._L.dlsm_hdp_trace_alloc(self._h, int(n_total), float(logp0)))

    def hdp_run(self, first, count):
        """enqueue Gibbs iterations first..first+count-1 of the HDP-LPCM (asynchronous)"""
        self._ck(self._L.dlsm_hdp_run(self._h, int(first), int(count)))

    def hdp_trace_read(self, first, count, positions=True, labels=True, weights=True, small=True):
        """dict of the stored samples first..first+count-1: ``Xs`` (positions), ``zs`` (labels),
        ``weights`` - the three large arrays, each optional - and the small ones (``small``)"""
        T, N, D, K = self.T, self.N, self.D, self.K
        out = {}
        if small:
            out.update(intercepts=np.zeros((count, 2)), logps=np.zeros(count),
                       mus=np.zeros((count, K, D)), sigmas=np.zeros((count, K)),
                       betas=np.zeros((count, K)), lambdas=np.zeros((count, 1)),
                       hypers=np.zeros((count, 6)))
        if weights:
            out['weights'] = np.zeros((count, T, K, K))
        if positions:
            out['Xs'] = np.zeros((count, T, N, D))
        if labels:
            out['zs'] = np.zeros((count, T, N), dtype=np.int64)

        def ptr(name):
            return _p(out[name]) if name in out else None
        self._ck(self._L.dlsm_hdp_trace_read(
            self._h, int(first), int(count), ptr('Xs'), ptr('intercepts'), ptr('logps'), ptr('mus'),
            ptr('sigmas'), ptr('zs'), ptr('betas'), ptr('weights'), ptr('lambdas'), ptr('hypers')))
        if small and self.model == UNDIRECTED:
            # (undirected device loop: the second slot carries the network log-likelihood of the
            # stored state, NaN where it is not known)
            out['logliks'] = out['intercepts'][:, 1].copy()
            out['intercepts'] = out['intercepts'][:, :1]
        elif small:
            out['logliks'] = np.full(count, np.nan)
        return out

    def hdp_trace_write(self, first, Xs=None, intercepts=None, logps=None, mus=None, sigmas=None,
                        zs=None, betas=None, weights=None, lambdas=None):
        """rows first .. of the device-resident trace from host arrays (the mirror of
        ``hdp_trace_read``); every given array has the same leading length"""
        T, N, D, K = self.T, self.N, self.D, self.K
        given = [a for a in (Xs, intercepts, logps, mus, sigmas, zs, betas, weights, lambdas)
                 if a is not None]
        count = int(np.shape(given[0])[0])

        def f(a, shape):
            return None if a is None else _p(_f64(np.reshape(a, (count,) + shape), (count,) + shape))
        zz = None if zs is None else _i64(zs, (count, T, N), 'zs')
        self._ck(self._L.dlsm_hdp_trace_write(
            self._h, int(first), count, f(Xs, (T, N, D)),
            None if intercepts is None else _p(_f64(
                np.reshape(intercepts, (count, -1))[:, :self.n_intercepts].copy())),
            f(logps, ()), f(mus, (K, D)), f(sigmas, (K,)), None if zz is None else _p(zz),
            f(betas, (K,)), f(weights, (T, K, K)), f(lambdas, ())))

    def hdp_queues(self):
        """queues the last ``hdp_run`` used: 2 when the intercept's likelihood pass ran beside the
        label update and the conjugate draws (undirected model, the process's only live chain)"""
        q = C.c_int(0)
        self._ck(self._L.dlsm_hdp_queues(self._h, C.byref(q)))
        return int(q.value)

    def hdp_get_aux(self):
        """auxiliary variables of the last iteration: m, m_bar, w_over, n, nk"""
        T, K = self.T, self.K
        m = np.zeros((T, K, K), dtype=np.int64); mb = np.zeros(K)
        wo = np.zeros((max(T - 1, 0), K), dtype=np.int64)
        n = np.zeros((T, K, K), dtype=np.int64); nk = np.zeros((T, K), dtype=np.int64)
        self._ck(self._L.dlsm_hdp_get_aux(self._h, _p(m), _p(mb), _p(wo), _p(n), _p(nk)))
        return dict(m=m, m_bar=mb, w_over=wo, n=n, nk=nk)

    def trace_read_radii(self, first, count):
        out = np.zeros((count, self.N))
        self._ck(self._L.dlsm_trace_read_radii(self._h, int(first), int(count), _p(out)))
        return out

    # -- starting values -----------------------------------
    def init_shortest_paths(self):
        self._ck(self._L.dlsm_init_shortest_paths(self._h))

    def init_get_dissimilarity(self, t):
        out = np.empty((self.N, self.N))
        self._ck(self._L.dlsm_init_get_dissimilarity(self._h, int(t), _p(out)))
        return out

    def init_smacof(self, t, X0, max_iter=300, eps=1e-6):
        """SMACOF runs from the starting configurations ``X0`` (n_init, N, D);
        returns (X[n_init, N, D], stress[n_init], n_iter[n_init])."""
        X0 = np.ascontiguousarray(X0, dtype=np.float64)
        if X0.ndim == 2:
            X0 = X0[None]
        X0 = _f64(X0, (X0.shape[0], self.N, self.D), 'X0')
        n_init = X0.shape[0]
        X = np.empty_like(X0)
        stress = np.empty(n_init)
        n_iter = np.empty(n_init, dtype=np.int32)
        self._ck(self._L.dlsm_init_smacof(self._h, int(t), n_init, _p(X0), int(max_iter),
                                          float(eps), _p(X), _p(stress), _p(n_iter)))
        return X, stress, n_iter

    def init_gmds_step(self, t, X_prev, lmbda=10.0, max_lanczos=256, tol=1e-12):
        """One Sarkar-Moore step; returns (X_t, evals[D], info) with info =
        {'n_lanczos', 'residual'}."""
        X_prev = _f64(X_prev, (self.N, self.D), 'X_prev')
        X = np.empty_like(X_prev)
        evals = np.empty(self.D)
        nl = np.zeros(1, dtype=np.int32)
        res = np.zeros(1)
        self._ck(self._L.dlsm_init_gmds_step(self._h, int(t), _p(X_prev), float(lmbda),
                                             int(max_lanczos), float(tol), _p(X), _p(evals),
                                             _p(nl), _p(res)))
        return X, evals, {'n_lanczos': int(nl[0]), 'residual': float(res[0])}

    def init_mle_sums(self, p0, p1):
        out = np.empty(3)
        self._ck(self._L.dlsm_init_mle_sums(self._h, float(p0), float(p1), _p(out)))
        return out

    def init_release(self):
        self._ck(self._L.dlsm_init_release(self._h))

    # -- post-loop processing ---------------------------------
    def init_kmeans_lloyd(self, Xc, centers_init, max_iter=300, tol=0.0):
        """scikit-learn's Lloyd loop on the centred (N, F) matrix from the given seeding; returns
        (centers, labels, n_iter) or None when a cluster emptied (the caller finishes on the host)"""
        Xc = _f64(Xc)
        N, F = Xc.shape
        K = int(np.shape(centers_init)[0])
        c0 = _f64(centers_init, (K, F), 'centers_init')
        cen = np.empty((K, F)); lab = np.empty(N, dtype=np.int32)
        nit = np.zeros(1, dtype=np.int32); empty = np.zeros(1, dtype=np.int32)
        self._ck(self._L.dlsm_init_kmeans_lloyd(self._h, _p(Xc), N, F, K, _p(c0), int(max_iter),
                                                float(tol), _p(cen), _p(lab), _p(nit), _p(empty)))
        if empty[0]:
            return None
        return cen, lab, int(nit[0])

    def post_cooccurrence(self, zs, K, want_matrix=True):
        """co-occurrence probabilities of the kept samples ``zs`` (S, T, N); returns the
        (T, N, N) matrices (or None) and keeps them on the device for the VI sums"""
        zs = _i64(zs, (np.shape(zs)[0], self.T, self.N), 'zs')
        out = np.empty((self.T, self.N, self.N)) if want_matrix else None
        self._ck(self._L.dlsm_post_cooccurrence(self._h, _p(zs), zs.shape[0], int(K),
                                                None if out is None else _p(out)))
        self._post_S = zs.shape[0]
        return out

    def post_expected_vi_sums(self):
        out = np.empty((self.T, self._post_S))
        self._ck(self._L.dlsm_post_expected_vi_sums(self._h, _p(out)))
        return out

    def post_release(self):
        self._ck(self._L.dlsm_post_release(self._h))

    # the same on the device-resident trace of hdp_run (rows first .. first + count - 1)
    def post_trace_label_counts(self, first, count):
        """(count, T, K) int32: nodes per label, time and stored sample"""
        nk = np.empty((count, self.T, self.K), dtype=np.int32)
        self._ck(self._L.dlsm_post_trace_label_counts(self._h, int(first), int(count),
                                                      nk.ctypes.data_as(c_i32_p)))
        return nk

    def post_trace_cooccurrence(self, first, count, want_matrix=False):
        """co-occurrence probabilities of the stored samples: returns (matrix or None, row sums
        (T, N)); the matrices stay on the device for the VI sums / ``post_get_cooccurrence``"""
        out = np.empty((self.T, self.N, self.N)) if want_matrix else None
        rs = np.empty((self.T, self.N))
        self._ck(self._L.dlsm_post_trace_cooccurrence(self._h, int(first), int(count),
                                                      None if out is None else _p(out), _p(rs)))
        self._post_S = int(count)
        return out, rs

    def post_get_cooccurrence(self):
        out = np.empty((self.T, self.N, self.N))
        self._ck(self._L.dlsm_post_get_cooccurrence(self._h, _p(out)))
        return out

    def post_trace_align(self, first, count, ref_row):
        """rotate the stored positions and cluster means of the rows onto row ``ref_row``"""
        self._ck(self._L.dlsm_post_trace_align(self._h, int(first), int(count), int(ref_row)))

    def post_trace_mean(self, first, count):
        out = np.empty((self.T, self.N, self.D))
        self._ck(self._L.dlsm_post_trace_mean(self._h, int(first), int(count), _p(out)))
        return out

    def post_latent_marginal_loglik(self, init_w, trans_w, mu, sigma, lmbda, row=-1):
        """approx_bic.py:54-76 at trace row ``row`` (-1: the chain's current positions)"""
        Ka = int(np.shape(sigma)[0])
        init_w = _f64(init_w, (Ka,), 'init_w')
        trans_w = _f64(trans_w, (self.T, Ka, Ka), 'trans_w')
        mu = _f64(mu, (Ka, self.D), 'mu')
        sigma = _f64(sigma, (Ka,), 'sigma')
        out = np.zeros(1)
        self._ck(self._L.dlsm_post_latent_marginal_loglik(
            self._h, int(row), _p(init_w), _p(trans_w), _p(mu), _p(sigma),
            float(np.ravel(lmbda)[0]), Ka, _p(out)))
        return float(out[0])

    # -- one-step-ahead forecasts ------------------------------
    def forecast_mean_probas(self, Xs, intercepts, zero_diag=False):
        Xs = np.ascontiguousarray(Xs, dtype=np.float64)
        S = Xs.shape[0]
        Xs = _f64(Xs, (S, self.N, self.D), 'Xs')
        b = _f64(np.broadcast_to(np.ravel(intercepts), (S,)) if np.size(intercepts) == 1
                 else np.ravel(intercepts), (S,), 'intercepts')
        out = np.empty((self.N, self.N))
        self._ck(self._L.dlsm_forecast_mean_probas(self._h, _p(Xs), _p(b), S, int(zero_diag),
                                                   _p(out)))
        return out

    def forecast_marginal(self, x, W, intercepts):
        W = np.ascontiguousarray(W, dtype=np.float64)
        S = W.shape[0]
        x = _f64(x, (self.N, self.D), 'x')
        W = _f64(W, (S, self.N), 'W')
        b = _f64(np.ravel(intercepts), (S,), 'intercepts')
        out = np.empty((self.N, self.N))
        self._ck(self._L.dlsm_forecast_marginal(self._h, _p(x), _p(W), _p(b), S, _p(out)))
        return out

    def profile_enable(self, on=True):
        self._ck(self._L.dlsm_profile_enable(self._h, int(on)))

    def profile_read(self, kernel):
        ms = C.c_double(0.0)
        n = C.c_int(0)
        self._ck(self._L.dlsm_profile_read(self._h, int(kernel), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_read_eval_stamps(self):
        """(mean in-kernel duration in us, launches) of k_spec_eval while profiling"""
        us = C.c_double(0.0)
        n = C.c_int(0)
        self._ck(self._L.dlsm_profile_read_eval_stamps(self._h, C.byref(us), C.byref(n)))
        return us.value, n.value

    def timer_start(self):
        self._ck(self._L.dlsm_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_double(0.0)
        self._ck(self._L.dlsm_timer_stop(self._h, C.byref(ms)))
        return ms.value
