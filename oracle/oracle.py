"""CPU ORACLE for the DynetLSM Gibbs hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module, and only as the checker / reported CPU baseline.
The product package ``dynetlsm_amd`` never imports it and has no CPU fallback.

Two layers:

* ctypes bindings to ``liboracle.so`` (``dynetlsm_oracle.c``): the scalar C
  restatement of every likelihood on the path plus the sweep / label update in
  the ENGINE's scan order with the engine's Philox draws.
* a small pure-Python restatement of the Metropolis sweep and the label block
  update that is generic over the source of random draws.  Driven by a numpy
  ``RandomState`` in the reference's draw order it reproduces the reference
  traces (pinned by ``tests/golden``); driven by Philox draws in the engine's
  scan order it must equal the C restatement.  That is the chain of trust
  reference -> python(MT19937) -> python(Philox) == C(Philox) -> HIP engine.

Reference citations (file:line) are relative to joshloyal/dynetlsm v0.1.0.
Parity is PINNED by tests/test_oracle_golden.py against tests/golden/*.npz.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_double_p = C.POINTER(C.c_double)
c_i64_p = C.POINTER(C.c_int64)
c_i32_p = C.POINTER(C.c_int32)

STREAM_SWEEP_NORMAL, STREAM_SWEEP_UNIFORM, STREAM_INTERCEPT, STREAM_LABELS, \
    STREAM_CONTROLS = range(5)


def build(force=False):
    so = os.path.join(_HERE, 'liboracle.so')
    src = os.path.join(_HERE, 'dynetlsm_oracle.c')
    if force or not os.path.exists(so) or \
            os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s', '-B', 'liboracle.so'])
    return so


class Chain(C.Structure):
    """mirror of ``orc_chain``"""
    _fields_ = [
        ('T', C.c_int), ('N', C.c_int), ('D', C.c_int), ('model', C.c_int),
        ('squared', C.c_int), ('Y', c_double_p),
        ('in_edges', c_i64_p), ('Din', C.c_int),
        ('out_edges', c_i64_p), ('Dout', C.c_int),
        ('degree', c_i64_p), ('ctrl_in', c_i64_p), ('ctrl_out', c_i64_p),
        ('C', C.c_int),
        ('X', c_double_p), ('intercept', C.c_double * 2), ('radii', c_double_p),
        ('prior_kind', C.c_int), ('tau_sq', C.c_double), ('sigma_sq', C.c_double),
        ('mu', c_double_p), ('sigma', c_double_p), ('lmbda', C.c_double),
        ('z', c_i64_p), ('K', C.c_int),
        ('step_size', c_double_p), ('n_accepted', c_i32_p), ('n_steps', c_i32_p),
        ('steps_until_tune', c_i32_p), ('tune', C.c_int),
        ('tune_interval', C.c_int),
        ('seed', C.c_uint64), ('chain', C.c_uint32), ('iter', C.c_uint32),
    ]


class ScalarSampler(C.Structure):
    _fields_ = [('step_size', C.c_double), ('n_accepted', C.c_int32),
                ('n_steps', C.c_int32), ('steps_until_tune', C.c_int32),
                ('tune', C.c_int), ('tune_interval', C.c_int)]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_partial_loglikelihood.restype = C.c_double
        L.orc_partial_loglikelihood.argtypes = [
            c_double_p, c_double_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int]
        L.orc_directed_partial_loglikelihood.restype = C.c_double
        L.orc_directed_partial_loglikelihood.argtypes = [
            c_double_p, c_double_p, c_double_p, C.c_int, C.c_int, C.c_double,
            C.c_double, C.c_int, C.c_int]
        L.orc_approx_directed_partial_loglikelihood.restype = C.c_double
        L.orc_approx_directed_partial_loglikelihood.argtypes = [
            c_double_p, c_double_p, c_i64_p, C.c_int, c_i64_p, C.c_int, c_i64_p,
            c_i64_p, c_i64_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
            C.c_int, C.c_int, C.c_int]
        L.orc_loglik_undirected.restype = C.c_double
        L.orc_loglik_undirected.argtypes = [
            c_double_p, c_double_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]
        L.orc_loglik_directed.restype = C.c_double
        L.orc_loglik_directed.argtypes = [
            c_double_p, c_double_p, c_double_p, C.c_int, C.c_int, C.c_int,
            C.c_double, C.c_double, C.c_int]
        L.orc_approx_loglik_directed.restype = C.c_double
        L.orc_approx_loglik_directed.argtypes = [
            c_double_p, c_double_p, c_i64_p, C.c_int, c_i64_p, c_i64_p, C.c_int,
            C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]
        L.orc_gaussian_likelihood.restype = None
        L.orc_gaussian_likelihood.argtypes = [
            c_double_p, C.c_int, c_double_p, c_double_p, C.c_double, C.c_int,
            C.c_int, C.c_int, C.c_int, c_double_p]
        L.orc_node_logp.restype = C.c_double
        L.orc_node_logp.argtypes = [C.POINTER(Chain), C.c_int, C.c_int, c_double_p]
        L.orc_sweep_positions.restype = None
        L.orc_sweep_positions.argtypes = [C.POINTER(Chain)]
        L.orc_center.restype = None
        L.orc_center.argtypes = [c_double_p, C.c_int, C.c_int, C.c_int]
        L.orc_sample_labels.restype = None
        L.orc_sample_labels.argtypes = [
            c_double_p, c_double_p, c_double_p, C.c_double, c_double_p, C.c_int,
            C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32,
            c_i64_p, c_double_p, c_i64_p]
        L.orc_lsm_log_prior.restype = C.c_double
        L.orc_lsm_log_prior.argtypes = [
            c_double_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
            c_double_p, C.c_int, c_double_p, C.c_double]
        L.orc_lsm_iteration_undirected.restype = C.c_double
        L.orc_lsm_iteration_undirected.argtypes = [
            C.POINTER(Chain), C.POINTER(ScalarSampler), C.c_double, C.c_double]
        L.orc_philox4x32.restype = None
        L.orc_philox4x32.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32,
                                     C.c_uint32, C.c_uint32,
                                     C.POINTER(C.c_uint32 * 4)]
        L.orc_philox_uniform2.restype = None
        L.orc_philox_uniform2.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32,
                                          C.c_uint32, C.c_uint32,
                                          C.POINTER(C.c_double * 2)]
        _LIB = L
    return _LIB


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(c_double_p)


def _i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(c_i64_p)


# --------------------------------------------------------------------------
# likelihoods (function seam: same argument meaning as the reference)
# --------------------------------------------------------------------------
def partial_loglikelihood(Y, X, intercept, node_id, squared=False):
    """static_network_fast.pyx:17-44"""
    Y, Yp = _f64(Y); X, Xp = _f64(X)
    return lib().orc_partial_loglikelihood(Yp, Xp, X.shape[0], X.shape[1],
                                           float(intercept), node_id, int(squared))


def directed_partial_loglikelihood(Y, X, radii, intercept_in, intercept_out,
                                   node_id, squared=False):
    """directed_likelihoods_fast.pyx:46-80"""
    Y, Yp = _f64(Y); X, Xp = _f64(X); radii, rp = _f64(radii)
    return lib().orc_directed_partial_loglikelihood(
        Yp, Xp, rp, X.shape[0], X.shape[1], intercept_in, intercept_out, node_id,
        int(squared))


def approx_directed_partial_loglikelihood(X, radii, in_edges, out_edges, degree,
                                          control_nodes_in, control_nodes_out,
                                          intercept_in, intercept_out, node_id,
                                          squared=False, ref_compat=False):
    """directed_likelihoods_fast.pyx:83-182"""
    X, Xp = _f64(X); radii, rp = _f64(radii)
    ie, iep = _i64(in_edges); oe, oep = _i64(out_edges); dg, dgp = _i64(degree)
    ci, cip = _i64(control_nodes_in); co, cop = _i64(control_nodes_out)
    return lib().orc_approx_directed_partial_loglikelihood(
        Xp, rp, iep, ie.shape[1], oep, oe.shape[1], dgp, cip, cop, ci.shape[1],
        X.shape[0], X.shape[1], intercept_in, intercept_out, node_id, int(squared),
        int(ref_compat))


def dynamic_network_loglikelihood_undirected(Y, X, intercept, squared=False):
    """network_likelihoods.py:26-33"""
    Y, Yp = _f64(Y); X, Xp = _f64(X)
    T, N, D = X.shape
    return lib().orc_loglik_undirected(Yp, Xp, T, N, D,
                                       float(np.asarray(intercept).ravel()[0]),
                                       int(squared))


def dynamic_network_loglikelihood_directed(Y, X, intercept_in, intercept_out,
                                           radii, squared=False):
    """network_likelihoods.py:16-22 -> directed_likelihoods_fast.pyx:185-205"""
    Y, Yp = _f64(Y); X, Xp = _f64(X); radii, rp = _f64(radii)
    T, N, D = X.shape
    return lib().orc_loglik_directed(Yp, Xp, rp, T, N, D, intercept_in,
                                     intercept_out, int(squared))


def approx_directed_network_loglikelihood(X, radii, in_edges, out_edges, degree,
                                          control_nodes, intercept_in,
                                          intercept_out, squared=False):
    """directed_likelihoods_fast.pyx:208-270 (in_edges is unused there too)"""
    X, Xp = _f64(X); radii, rp = _f64(radii)
    oe, oep = _i64(out_edges); dg, dgp = _i64(degree); co, cop = _i64(control_nodes)
    T, N, D = X.shape
    return lib().orc_approx_loglik_directed(Xp, rp, oep, oe.shape[2], dgp, cop,
                                            co.shape[2], T, N, D, intercept_in,
                                            intercept_out, int(squared))


def compute_gaussian_likelihood(X, mu, sigma, lmbda, normalize=True):
    """gaussian_likelihood_fast.pyx:30-54 ; X is (T, D)"""
    X, Xp = _f64(X); mu, mp = _f64(mu); sigma, sp = _f64(sigma)
    T, D = X.shape
    K = sigma.shape[0]
    out = np.zeros((T, K))
    lib().orc_gaussian_likelihood(Xp, D, mp, sp, float(np.asarray(lmbda).ravel()[0]), T, D, K,
                                  int(normalize), out.ctypes.data_as(c_double_p))
    return out


# --------------------------------------------------------------------------
# Philox4x32-10 in numpy (vectorised; independent of the C code)
# --------------------------------------------------------------------------
def philox4x32(seed, c0, c1, c2, c3):
    c0, c1, c2, c3 = [np.asarray(v, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
                      for v in np.broadcast_arrays(c0, c1, c2, c3)]
    k0 = np.uint64(seed & 0xFFFFFFFF)
    k1 = np.uint64((seed >> 32) & 0xFFFFFFFF)
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    s32 = np.uint64(32)
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        n0 = (p1 >> s32) ^ c1 ^ k0
        n1 = p1 & mask
        n2 = (p0 >> s32) ^ c3 ^ k1
        n3 = p0 & mask
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return c0, c1, c2, c3


def _u53(hi, lo):
    k = (hi >> np.uint64(5)).astype(np.float64) * 67108864.0 + \
        (lo >> np.uint64(6)).astype(np.float64)
    return (k + 1.0) * (1.0 / 9007199254740992.0)


def philox_uniform2(seed, c0, c1, c2, c3):
    r0, r1, r2, r3 = philox4x32(seed, c0, c1, c2, c3)
    return _u53(r0, r1), _u53(r2, r3)


def stream_word(chain, stream):
    return ((chain << 8) | stream) & 0xFFFFFFFF


class PhiloxDraws(object):
    """The engine's draws for the sweep (keyed by iteration, t, j)."""
    def __init__(self, seed, chain, it):
        self.seed, self.chain, self.it = seed, chain, it

    def randn(self, t, j, D):
        z = np.zeros(D)
        for d in range(0, D, 2):
            u1, u2 = philox_uniform2(self.seed, j, t | ((d // 2) << 16), self.it,
                                     stream_word(self.chain, STREAM_SWEEP_NORMAL))
            r = np.sqrt(-2.0 * np.log(u1))
            a = 6.283185307179586476925286766559 * u2
            z[d] = r * np.cos(a)
            if d + 1 < D:
                z[d + 1] = r * np.sin(a)
        return z

    def rand(self, t, j):
        u1, _ = philox_uniform2(self.seed, j, t, self.it,
                                stream_word(self.chain, STREAM_SWEEP_UNIFORM))
        return float(u1)


class MTDraws(object):
    """The reference's draws: one numpy RandomState consumed in call order
    (metropolis.py:44 ``randn(d)`` then :49 ``rand()``)."""
    def __init__(self, rng):
        self.rng = rng

    def randn(self, t, j, D):
        return self.rng.randn(D)

    def rand(self, t, j):
        return self.rng.rand()


# --------------------------------------------------------------------------
# Metropolis state (struct of arrays; one sampler per (t, node))
# --------------------------------------------------------------------------
class SamplerGrid(object):
    """metropolis.py:85-94 for a T x N grid of random-walk samplers."""
    def __init__(self, T, N, step_size=0.1, tune=500, tune_interval=100):
        self.step_size = np.full((T, N), float(step_size))
        self.n_accepted = np.zeros((T, N), dtype=np.int32)
        self.n_steps = np.zeros((T, N), dtype=np.int32)
        self.steps_until_tune = np.full((T, N), tune_interval, dtype=np.int32)
        self.tune = tune            # None == no adaptation
        self.tune_interval = tune_interval

    def copy(self):
        g = SamplerGrid.__new__(SamplerGrid)
        g.step_size = self.step_size.copy()
        g.n_accepted = self.n_accepted.copy()
        g.n_steps = self.n_steps.copy()
        g.steps_until_tune = self.steps_until_tune.copy()
        g.tune, g.tune_interval = self.tune, self.tune_interval
        return g


def tune_step_size_random_walk(step_size, acc_rate):
    """metropolis.py:5-20"""
    if acc_rate < 0.001:
        return step_size * 0.1
    elif acc_rate < 0.05:
        return step_size * 0.5
    elif acc_rate < 0.25:
        return step_size * 0.9
    elif acc_rate > 0.95:
        return step_size * 10.0
    elif acc_rate > 0.75:
        return step_size * 2.0
    elif acc_rate > 0.4:
        return step_size * 1.1
    return step_size


def _bookkeeping(g, t, j, accepted):
    """metropolis.py:110-136"""
    g.n_accepted[t, j] += accepted
    g.n_steps[t, j] += 1
    if g.tune is not None:
        if g.n_steps[t, j] < g.tune and g.steps_until_tune[t, j] == 0:
            rate = g.n_accepted[t, j] / g.tune_interval
            g.step_size[t, j] = tune_step_size_random_walk(g.step_size[t, j], rate)
            g.n_accepted[t, j] = 0
            g.steps_until_tune[t, j] = g.tune_interval
        else:
            g.steps_until_tune[t, j] -= 1


class ChainState(object):
    """Owns numpy arrays + the ctypes ``orc_chain`` view of them."""
    def __init__(self, X, samplers, Y=None, intercept=(0.0,), radii=None,
                 model=0, squared=False, case_control=None,
                 tau_sq=2.0, sigma_sq=0.1,
                 mu=None, sigma=None, lmbda=None, z=None,
                 seed=0, chain=0, it=0):
        self.X = np.ascontiguousarray(X, dtype=np.float64).copy()
        T, N, D = self.X.shape
        self.samplers = samplers
        c = Chain()
        c.T, c.N, c.D, c.model, c.squared = T, N, D, model, int(squared)
        self._keep = []
        if Y is not None:
            self.Y, c.Y = _f64(Y)
        if case_control is not None:
            ie, c.in_edges = _i64(case_control['in_edges'])
            oe, c.out_edges = _i64(case_control['out_edges'])
            dg, c.degree = _i64(case_control['degree'])
            ci, c.ctrl_in = _i64(case_control['control_nodes_in'])
            co, c.ctrl_out = _i64(case_control['control_nodes_out'])
            c.Din, c.Dout, c.C = ie.shape[2], oe.shape[2], ci.shape[2]
            self._keep += [ie, oe, dg, ci, co]
        c.X = self.X.ctypes.data_as(c_double_p)
        ic = np.asarray(intercept, dtype=np.float64).ravel()
        c.intercept[0] = ic[0]
        c.intercept[1] = ic[1] if ic.size > 1 else 0.0
        if radii is not None:
            self.radii, c.radii = _f64(radii)
        if mu is not None:
            c.prior_kind = 1
            self.mu, c.mu = _f64(mu)
            self.sigma, c.sigma = _f64(sigma)
            self.z, c.z = _i64(z)
            c.lmbda = float(np.asarray(lmbda).ravel()[0])
            c.K = self.sigma.shape[0]
        else:
            c.prior_kind = 0
            c.tau_sq, c.sigma_sq = float(tau_sq), float(sigma_sq)
        g = samplers
        c.step_size = g.step_size.ctypes.data_as(c_double_p)
        c.n_accepted = g.n_accepted.ctypes.data_as(c_i32_p)
        c.n_steps = g.n_steps.ctypes.data_as(c_i32_p)
        c.steps_until_tune = g.steps_until_tune.ctypes.data_as(c_i32_p)
        c.tune = -1 if g.tune is None else int(g.tune)
        c.tune_interval = int(g.tune_interval)
        c.seed, c.chain, c.iter = seed, chain, it
        self.c = c

    @property
    def intercept(self):
        return np.array([self.c.intercept[0], self.c.intercept[1]])

    def node_logp(self, t, j, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        return lib().orc_node_logp(C.byref(self.c), t, j,
                                   x.ctypes.data_as(c_double_p))

    def sweep_c(self):
        """engine scan order + Philox draws, all in C"""
        lib().orc_sweep_positions(C.byref(self.c))
        return self.X

    def sweep_py(self, draws, order='reference'):
        """generic restatement of sample_latent_positions.py:92-146 / :149-206
        + metropolis.py:40-54.  order: 'reference' (t-major) or 'engine'
        (even-t slices then odd-t slices)."""
        T, N, D = self.X.shape
        ts = list(range(T)) if order == 'reference' else \
            list(range(0, T, 2)) + list(range(1, T, 2))
        g = self.samplers
        for t in ts:
            for j in range(N):
                x0 = self.X[t, j].copy()
                x = x0 + g.step_size[t, j] * draws.randn(t, j, D)
                ratio = self.node_logp(t, j, x) - self.node_logp(t, j, x0)
                u = draws.rand(t, j)
                accepted = 1
                if np.log(u) >= ratio:
                    x = x0
                    accepted = 0
                self.X[t, j] = x
                _bookkeeping(g, t, j, accepted)
        return self.X


def center(X):
    """lsm.py:501 / hdp_lpcm.py:852"""
    return X - np.mean(X, axis=(0, 1))


def procrustes_rotation(X_ref, X):
    """procrustes.py:20-35 (scipy.linalg.orthogonal_procrustes restated):
    R = argmin ||X R - X_ref||_F over orthogonal R; returns X R, R."""
    A = X.reshape(-1, X.shape[-1])
    B = X_ref.reshape(-1, X.shape[-1])
    u, _, vt = np.linalg.svd(B.T.dot(A).T)
    R = u.dot(vt)
    return A.dot(R).reshape(X.shape), R


# --------------------------------------------------------------------------
# label block update
# --------------------------------------------------------------------------
def sample_labels_block_mt(X, mu, sigma, lmbda, w, rng):
    """sample_labels.py:134-190 driven by a numpy RandomState (reference
    draw order: one ``uniform(0, cdf[-1])`` per (node, t))."""
    T, N, D = X.shape
    K = sigma.shape[0]
    n = np.zeros((T, K, K))
    nk = np.zeros((T, K), dtype=np.int64)
    resp = np.zeros((T, N, K), dtype=np.int64)
    z = np.zeros((T, N), dtype=np.int64)
    bm = np.ones((T, K))
    pm = np.zeros((T, K))
    for i in range(N):
        L = compute_gaussian_likelihood(X[:, i], mu, sigma, lmbda, normalize=False)
        for t in range(T - 1, 0, -1):
            pm[t] = L[t] * bm[t]
            bm[t - 1] = np.dot(w[t], pm[t])
            bm[t - 1] /= np.sum(bm[t - 1])
        pm[0] = L[0] * bm[0]
        for t in range(T):
            probas = w[0, 0] * pm[0] if t == 0 else w[t, z[t - 1, i]] * pm[t]
            cdf = np.cumsum(probas)
            u = rng.uniform(0, cdf[-1])
            z[t, i] = np.sum(u > cdf)
            if t == 0:
                n[0, 0, z[t, i]] += 1
            else:
                n[t, z[t - 1, i], z[t, i]] += 1
            resp[t, i, z[t, i]] = 1
            nk[t, z[t, i]] += 1
    return z, n, nk, resp


def sample_labels_block_philox(X, mu, sigma, lmbda, w, seed, chain, it):
    """a12 with the engine's draws, in C."""
    X, Xp = _f64(X); mu, mp = _f64(mu); sigma, sp = _f64(sigma); w, wp = _f64(w)
    T, N, D = X.shape
    K = sigma.shape[0]
    z = np.zeros((T, N), dtype=np.int64)
    n = np.zeros((T, K, K))
    nk = np.zeros((T, K), dtype=np.int64)
    lib().orc_sample_labels(Xp, mp, sp, float(np.asarray(lmbda).ravel()[0]), wp,
                            T, N, D, K, seed, chain, it,
                            z.ctypes.data_as(c_i64_p), n.ctypes.data_as(c_double_p),
                            nk.ctypes.data_as(c_i64_p))
    return z, n, nk


# --------------------------------------------------------------------------
# case-control bookkeeping (case_control_likelihood.py:37-73), numpy restatement
# --------------------------------------------------------------------------
def case_control_init(Y):
    """degrees_[T,N,2] (col 0 in, col 1 out), zero padded in/out edge lists."""
    T, N, _ = Y.shape
    deg = np.zeros((T, N, 2), dtype=np.int64)
    deg[:, :, 0] = Y.sum(axis=1)
    deg[:, :, 1] = Y.sum(axis=2)
    in_edges = np.zeros((T, N, int(deg[:, :, 0].max())), dtype=np.int64)
    out_edges = np.zeros((T, N, int(deg[:, :, 1].max())), dtype=np.int64)
    for t in range(T):
        for i in range(N):
            idx = np.where(Y[t, i, :] == 1)[0]
            out_edges[t, i, :idx.shape[0]] = idx
            idx = np.where(Y[t, :, i] == 1)[0]
            in_edges[t, i, :idx.shape[0]] = idx
    return deg, in_edges, out_edges


def lsm_log_prior(X, tau_sq, sigma_sq, intercept, intercept_prior, intercept_var):
    """lsm.py:604-623"""
    X, Xp = _f64(X)
    ic, icp = _f64(np.atleast_1d(intercept))
    ip, ipp = _f64(np.atleast_1d(intercept_prior))
    T, N, D = X.shape
    return lib().orc_lsm_log_prior(Xp, T, N, D, tau_sq, sigma_sq, icp, ic.size,
                                   ipp, intercept_var)


def lsm_iteration_undirected(state, isamp, intercept_prior, intercept_var):
    """one engine-schedule LSM iteration in C; returns the log-posterior trace"""
    return lib().orc_lsm_iteration_undirected(
        C.byref(state.c), C.byref(isamp), intercept_prior, intercept_var)


# --------------------------------------------------------------------------
# reference-order (MT19937) LSM loop: lsm.py:474-572 restated on top of the
# pieces above.  Used only to pin the oracle against the reference's fit().
# --------------------------------------------------------------------------
class ScalarMetropolis(object):
    """metropolis.py:85-136 for a scalar / vector random-walk block."""
    def __init__(self, step_size=0.1, tune=500, tune_interval=100):
        self.step_size, self.tune, self.tune_interval = step_size, tune, tune_interval
        self.steps_until_tune = tune_interval
        self.n_accepted = 0
        self.n_steps = 0

    def book(self, accepted, rule=tune_step_size_random_walk):
        self.n_accepted += accepted
        self.n_steps += 1
        if self.tune is not None:
            if self.n_steps < self.tune and self.steps_until_tune == 0:
                self.step_size = rule(self.step_size,
                                      self.n_accepted / self.tune_interval)
                self.n_accepted = 0
                self.steps_until_tune = self.tune_interval
            else:
                self.steps_until_tune -= 1

    def step_rw(self, x0, logp, rng):
        """metropolis.py:40-54"""
        x = x0 + self.step_size * rng.randn(x0.shape[0])
        ratio = logp(x) - logp(x0)
        u = rng.rand()
        accepted = 1
        if np.log(u) >= ratio:
            x, accepted = x0, 0
        self.book(accepted)
        return x

    def step_dirichlet(self, x0, logp, rng, reg=1e-5):
        """metropolis.py:57-82"""
        from scipy.stats import dirichlet
        x = rng.dirichlet(self.step_size * x0)
        if np.any(x == 0.):
            x += reg
            x /= np.sum(x)
        ratio = logp(x) - logp(x0)
        ratio += (dirichlet.logpdf(x0, self.step_size * x) -
                  dirichlet.logpdf(x, self.step_size * x0))
        u = rng.rand()
        accepted = 1
        if np.log(u) >= ratio:
            x, accepted = x0, 0
        self.book(accepted)   # tune is None for the radii sampler (lsm.py:470)
        return x


def lsm_reference_loop(Y, X0, intercept0, rng, n_total, n_iter_procrustes,
                       tau_sq, sigma_sq, intercept_prior, intercept_var,
                       samplers, isamplers, radii0=None, radii_sampler=None,
                       case_control=None, logp0=None):
    """lsm.py:474-572 for a fully observed network; returns the traces
    (Xs, intercepts, radiis, logps).  ``case_control`` = dict of the reference
    sampler's arrays (no resampling inside the loop)."""
    T, N, D = X0.shape
    directed = radii0 is not None
    Xs = np.zeros((n_total, T, N, D)); Xs[0] = X0
    ics = np.zeros((n_total, intercept0.shape[0])); ics[0] = intercept0
    rds = np.zeros((n_total, N)) if directed else None
    if directed:
        rds[0] = radii0
    logps = np.zeros(n_total)
    if logp0 is not None:
        logps[0] = logp0

    def full_loglik(X, ic, radii):
        if not directed:
            return dynamic_network_loglikelihood_undirected(Y, X, ic[0])
        if case_control is not None:
            return approx_directed_network_loglikelihood(
                X, radii, case_control['in_edges'], case_control['out_edges'],
                case_control['degree'], case_control['control_nodes_out'],
                ic[0], ic[1])
        return dynamic_network_loglikelihood_directed(Y, X, ic[0], ic[1], radii)

    for it in range(1, n_total):
        X = Xs[it - 1].copy()
        ic = ics[it - 1].copy()
        radii = rds[it - 1].copy() if directed else None
        model = 0 if not directed else (2 if case_control is not None else 1)
        st = ChainState(X, samplers, Y=Y, intercept=ic, radii=radii, model=model,
                        case_control=case_control, tau_sq=tau_sq,
                        sigma_sq=sigma_sq)
        X = st.sweep_py(MTDraws(rng), order='reference').copy()
        if it > n_iter_procrustes:
            prev_map = np.argmax(logps[:(n_iter_procrustes + 1)])
            X, _ = procrustes_rotation(Xs[prev_map], X)
        X = center(X)
        # sample_coefficients.py:12-88
        if directed:
            def lp_in(x):
                return (full_loglik(X, np.array([x[0], ic[1]]), radii) -
                        (x[0] - intercept_prior[0]) ** 2 / (2 * intercept_var))
            ic[0] = isamplers[0].step_rw(np.array([ic[0]]), lp_in, rng)[0]

            def lp_out(x):
                return (full_loglik(X, np.array([ic[0], x[0]]), radii) -
                        (x[0] - intercept_prior[1]) ** 2 / (2 * intercept_var))
            ic[1] = isamplers[1].step_rw(np.array([ic[1]]), lp_out, rng)[0]
            # sample_coefficients.py:91-121
            radii = radii_sampler.step_dirichlet(
                radii, lambda r: full_loglik(X, ic, r), rng)
        else:
            def lp(x):
                return (full_loglik(X, x, None) -
                        (x[0] - intercept_prior[0]) ** 2 / (2 * intercept_var))
            ic = isamplers[0].step_rw(ic, lp, rng)
        logps[it] = full_loglik(X, ic, radii) + lsm_log_prior(
            X, tau_sq, sigma_sq, ic, intercept_prior, intercept_var)
        Xs[it], ics[it] = X, ic
        if directed:
            rds[it] = radii
    return Xs, ics, rds, logps


# --------------------------------------------------------------------------
# engine-schedule (Philox) iteration of the DIRECTED LSM: what the device loop of
# dlsm_lsm_run does for model 1 / 2 (lsm.py:474-572 with the engine's draws):
# sweep (C oracle), Procrustes / centring, intercept_in, intercept_out, radii.
# --------------------------------------------------------------------------
STREAM_RADII = 5


def _box_muller(u0, u1):
    r = np.sqrt(-2.0 * np.log(u0))
    a = 6.283185307179586476925286766559 * u1
    return r * np.cos(a), r * np.sin(a)


def philox_gamma(seed, chain, i, it, a):
    """Gamma(a, 1) by Marsaglia & Tsang with the engine's counters: attempt k of node i
    uses counter (i, 2k, it) for the normal and (i, 2k + 1, it) for the uniforms"""
    aa = a + 1.0 if a < 1.0 else a
    d = aa - 1.0 / 3.0
    cc = 1.0 / np.sqrt(9.0 * d)
    for att in range(4096):
        u0, u1 = philox_uniform2(seed, i, 2 * att, it, stream_word(chain, STREAM_RADII))
        z0, _ = _box_muller(float(u0), float(u1))
        w0, w1 = philox_uniform2(seed, i, 2 * att + 1, it, stream_word(chain, STREAM_RADII))
        w0, w1 = float(w0), float(w1)
        t = 1.0 + cc * z0
        if t <= 0.0:
            continue
        v = t * t * t
        x2 = z0 * z0
        if w0 < 1.0 - 0.0331 * x2 * x2 or np.log(w0) < 0.5 * x2 + d * (1.0 - v + np.log(v)):
            out = d * v
            if a < 1.0:
                out *= w1 ** (1.0 / a)
            return out
    return 0.0


def tune_step_size_dirichlet(step_size, acc_rate):
    """metropolis.py:23-37"""
    if acc_rate < 0.001:
        return step_size * 10.0
    if acc_rate < 0.05:
        return step_size * 2.0
    if acc_rate < 0.25:
        return step_size * 1.1
    if acc_rate > 0.95:
        return step_size * 0.1
    if acc_rate > 0.75:
        return step_size * 0.5
    if acc_rate > 0.4:
        return step_size * 0.9
    return step_size


def directed_coefficient_steps(state, it, loglik, isamp, rsamp, intercept_prior, intercept_var):
    """intercept_in, intercept_out (sample_coefficients.py:12-75) and radii (:91-121) steps of one
    directed iteration at ``state``'s positions, with the engine's Philox draws; updates the
    state's intercepts and radii in place and returns the network log-likelihood of the new state"""
    from scipy.special import gammaln
    seed, chain = state.c.seed, state.c.chain
    X = state.X
    b = state.intercept.copy()
    radii = state.radii
    v = intercept_var
    for k in range(2):
        u0, u1 = philox_uniform2(seed, k, 0, it, stream_word(chain, STREAM_INTERCEPT))
        z0, _ = _box_muller(float(u0), float(u1))
        prop = b.copy()
        prop[k] = b[k] + isamp[k].step_size * z0
        ll_prop, ll_cur = loglik(X, prop, radii), loglik(X, b, radii)
        ratio = ((ll_prop - (prop[k] - intercept_prior[k]) ** 2 / (2 * v)) -
                 (ll_cur - (b[k] - intercept_prior[k]) ** 2 / (2 * v)))
        lu, _ = philox_uniform2(seed, k, 1, it, stream_word(chain, STREAM_INTERCEPT))
        accepted = int(not (np.log(float(lu)) >= ratio))
        if accepted:
            b = prop
        isamp[k].book(accepted)
    state.c.intercept[0], state.c.intercept[1] = b[0], b[1]
    N = radii.shape[0]
    step = rsamp.step_size
    g = np.array([philox_gamma(seed, chain, i, it, step * radii[i]) for i in range(N)])
    x = g * (1.0 / g.sum())
    if np.any(x == 0.):
        x = x + 1e-5
        x = x / np.sum(x)
    q = ((gammaln(step * x.sum()) - gammaln(step * x).sum() + ((step * x - 1) * np.log(radii)).sum()) -
         (gammaln(step * radii.sum()) - gammaln(step * radii).sum() +
          ((step * radii - 1) * np.log(x)).sum()))
    ll_cur, ll_alt = loglik(X, b, radii), loglik(X, b, x)
    lu, _ = philox_uniform2(seed, 0xFFFFFFFF, 0, it, stream_word(chain, STREAM_RADII))
    accepted = int(not (np.log(float(lu)) >= (ll_alt - ll_cur) + q))
    llf = ll_cur
    if accepted:
        radii[:] = x
        llf = ll_alt
    rsamp.book(accepted, rule=tune_step_size_dirichlet)
    return llf


def lsm_iteration_directed(state, it, loglik, isamp, rsamp, intercept_prior, intercept_var,
                           X_ref=None):
    """One iteration on ``state`` (ChainState, model 1 or 2; its X, intercept and radii are
    updated in place); ``loglik(X, intercept, radii)`` is the model's network log-likelihood;
    isamp[2], rsamp are ScalarMetropolis objects.  Returns the log-posterior trace value."""
    state.c.iter = it
    state.sweep_c()
    X = state.X
    if X_ref is not None:
        X[:] = procrustes_rotation(X_ref, X)[0]
    X[:] = center(X)
    llf = directed_coefficient_steps(state, it, loglik, isamp, rsamp, intercept_prior, intercept_var)
    return llf + lsm_log_prior(X, state.c.tau_sq, state.c.sigma_sq, state.intercept, intercept_prior,
                               intercept_var)
