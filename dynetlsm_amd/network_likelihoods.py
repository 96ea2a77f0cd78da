"""Function seam: the reference's likelihood functions, same names, argument
meaning and dtype rules, evaluated by the HIP engine.

Each call stages its arguments on the GPU, so these are for parity tests and
drop-in monkey-patching, not for speed: the fast path is ``Chain``.

Reference typed-memoryview semantics are kept: ``float64`` / ``int64`` only
(another dtype raises ``ValueError``), C-contiguity required where the
reference requires it, Python ``float`` / ``ndarray`` returned.
"""
import numpy as np

from .engine import Chain

__all__ = ['partial_loglikelihood', 'directed_partial_loglikelihood',
           'approx_directed_partial_loglikelihood',
           'dynamic_network_loglikelihood_undirected',
           'dynamic_network_loglikelihood_directed',
           'directed_network_loglikelihood_fast',
           'approx_directed_network_loglikelihood',
           'compute_gaussian_likelihood']


def _need(a, dtype, ndim, name, contiguous=False):
    a = np.asarray(a) if not isinstance(a, np.ndarray) else a
    if a.dtype != dtype:
        raise ValueError("Buffer dtype mismatch, expected '%s' but got '%s' (%s)"
                         % (np.dtype(dtype).name, a.dtype.name, name))
    if a.ndim != ndim:
        raise ValueError('Buffer has wrong number of dimensions (expected %d, '
                         'got %d) (%s)' % (ndim, a.ndim, name))
    if contiguous and not a.flags['C_CONTIGUOUS']:
        raise ValueError('ndarray is not C-contiguous (%s)' % name)
    return a


def partial_loglikelihood(Y, X, intercept, node_id, squared=False):
    """static_network_fast.pyx:17-44"""
    Y = _need(Y, np.float64, 2, 'Y'); X = _need(X, np.float64, 2, 'X')
    N, D = X.shape
    with Chain(1, N, D, 'undirected') as c:
        c.upload_network(Y[None]); c.set_positions(X[None])
        c.set_intercepts([float(np.asarray(intercept).ravel()[0])])
        c.set_squared(squared)
        return c.loglik_partial(0, int(node_id))


def directed_partial_loglikelihood(Y, X, radii, intercept_in, intercept_out,
                                   node_id, squared=False):
    """directed_likelihoods_fast.pyx:46-80 (Y and X must be C-contiguous)"""
    Y = _need(Y, np.float64, 2, 'Y', True); X = _need(X, np.float64, 2, 'X', True)
    radii = _need(radii, np.float64, 1, 'radii')
    N, D = X.shape
    with Chain(1, N, D, 'directed') as c:
        c.upload_network(Y[None]); c.set_positions(X[None]); c.set_radii(radii)
        c.set_intercepts([intercept_in, intercept_out]); c.set_squared(squared)
        return c.loglik_partial(0, int(node_id))


def approx_directed_partial_loglikelihood(X, radii, in_edges, out_edges, degree,
                                          control_nodes_in, control_nodes_out,
                                          intercept_in, intercept_out, node_id,
                                          squared=False):
    """directed_likelihoods_fast.pyx:83-182.  The second control loop tests the
    OUT list's own -1 sentinel (the reference tests the IN list's, :160-167)."""
    X = _need(X, np.float64, 2, 'X'); radii = _need(radii, np.float64, 1, 'radii')
    ie = _need(in_edges, np.int64, 2, 'in_edges')
    oe = _need(out_edges, np.int64, 2, 'out_edges')
    dg = _need(degree, np.int64, 2, 'degree')
    ci = _need(control_nodes_in, np.int64, 2, 'control_nodes_in')
    co = _need(control_nodes_out, np.int64, 2, 'control_nodes_out')
    N, D = X.shape
    with Chain(1, N, D, 'case_control') as c:
        c.upload_edges(ie[None], oe[None], dg[None]); c.set_controls(ci[None], co[None])
        c.set_positions(X[None]); c.set_radii(radii)
        c.set_intercepts([intercept_in, intercept_out]); c.set_squared(squared)
        return c.loglik_partial(0, int(node_id))


def dynamic_network_loglikelihood_undirected(Y, X, intercept, squared=False,
                                             dist=None):
    """network_likelihoods.py:26-33.  ``dist`` (the reference's cached distance
    matrix) is accepted and ignored: distances are recomputed from X on chip."""
    Y = _need(Y, np.float64, 3, 'Y'); X = _need(X, np.float64, 3, 'X')
    T, N, D = X.shape
    with Chain(T, N, D, 'undirected') as c:
        c.upload_network(Y); c.set_positions(X); c.set_squared(squared)
        b = float(np.asarray(intercept).ravel()[0])
        return np.float64(c.loglik_full([[b]])[0])


def dynamic_network_loglikelihood_directed(Y, X, intercept_in, intercept_out, radii,
                                           squared=False, dist=None):
    """network_likelihoods.py:16-22"""
    Y = _need(Y, np.float64, 3, 'Y', True); X = _need(X, np.float64, 3, 'X')
    radii = _need(radii, np.float64, 1, 'radii')
    T, N, D = X.shape
    with Chain(T, N, D, 'directed') as c:
        c.upload_network(Y); c.set_positions(X); c.set_radii(radii)
        c.set_squared(squared)
        return float(c.loglik_full([[intercept_in, intercept_out]])[0])


def directed_network_loglikelihood_fast(Y, X, radii, intercept_in, intercept_out,
                                        squared=False):
    """directed_likelihoods_fast.pyx:185-205 with X in place of the cached dist"""
    return dynamic_network_loglikelihood_directed(Y, X, intercept_in, intercept_out,
                                                  radii, squared=squared)


def approx_directed_network_loglikelihood(X, radii, in_edges, out_edges, degree,
                                          control_nodes, intercept_in, intercept_out,
                                          squared=False):
    """directed_likelihoods_fast.pyx:208-270"""
    X = _need(X, np.float64, 3, 'X'); radii = _need(radii, np.float64, 1, 'radii')
    ie = _need(in_edges, np.int64, 3, 'in_edges')
    oe = _need(out_edges, np.int64, 3, 'out_edges')
    dg = _need(degree, np.int64, 3, 'degree')
    co = _need(control_nodes, np.int64, 3, 'control_nodes')
    T, N, D = X.shape
    with Chain(T, N, D, 'case_control') as c:
        c.upload_edges(ie, oe, dg); c.set_controls(co, co)
        c.set_positions(X); c.set_radii(radii); c.set_squared(squared)
        return float(c.loglik_full([[intercept_in, intercept_out]])[0])


def compute_gaussian_likelihood(X, mu, sigma, lmbda, normalize=True):
    """gaussian_likelihood_fast.pyx:30-54 ; X is the (T, D) path of one node"""
    X = _need(X, np.float64, 2, 'X'); mu = _need(mu, np.float64, 2, 'mu')
    sigma = _need(sigma, np.float64, 1, 'sigma')
    T, D = X.shape
    Xp = np.zeros((T, 2, D)); Xp[:, 0] = X        # engine needs N >= 2
    with Chain(T, 2, D, 'undirected') as c:
        c.set_positions(Xp)
        c.set_prior_mixture(mu, sigma, lmbda, np.zeros((T, 2), dtype=np.int64))
        return c.gaussian_likelihood(0, normalize)
