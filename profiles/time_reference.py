"""Time the REFERENCE itself (joshloyal/dynetlsm, imported from /root/reference as
tests/golden/make_golden.py does) on the BASELINE.json configurations, in the BUILD
CONTAINER: the reference cannot travel to the GPU box, so its own numbers are measured
here, on this container's host cores, and committed as profiles/r02_reference_timing.json.

    python profiles/time_reference.py [c1] [c2] [c3] [c4]

  c1  DynamicNetworkLSM(n_iter=500, tune=None, burn=None, random_state=42).fit(monks): the
      whole fit() and the Gibbs loop alone (lsm.py:474-572)
  c2  T=10 N=2000 undirected: one sweep of sample_latent_positions (lsm.py:478-487) + one
      dynamic_network_loglikelihood_undirected; an iteration = sweep + 3 evaluations
      (sample_intercepts 2, logp 1)
  c3  the same network with the AR-mixture prior: sample_latent_positions_mixture,
      sample_labels_block, one evaluation; iteration = sweep + labels + 3 evaluations
  c4  directed case-control T=5 N=10 000, 100 controls: approx_directed_partial_loglikelihood
      per call (x 2 T N per sweep) and approx_directed_network_loglikelihood (x 7 per
      iteration); the Python closure overhead of the sweep is NOT included (lower bound)
The reference is single-threaded (GIL held, no prange): 1 core.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def c1(ref):
    from dynetlsm import DynamicNetworkLSM
    Y = np.load(os.path.join(ROOT, 'tests', 'golden', 'monks.npz'))['Y_undirected']
    t0 = time.perf_counter()
    m = DynamicNetworkLSM(n_iter=500, tune=None, burn=None, random_state=42)
    m.fit(Y)
    total = time.perf_counter() - t0
    # the Gibbs loop alone: a second fit of 1 iteration measures everything around it
    t0 = time.perf_counter()
    DynamicNetworkLSM(n_iter=1, tune=None, burn=None, random_state=42).fit(Y)
    setup = time.perf_counter() - t0
    loop = total - setup
    return dict(config='C1 DynamicNetworkLSM monks T=3 N=18, 500 iterations',
                fit_seconds=round(total, 3), loop_seconds=round(loop, 3),
                it_per_s=round(499 / loop, 1))


def _samplers(T, N, step):
    from dynetlsm.metropolis import Metropolis
    return [[Metropolis(step_size=step, tune=None) for _ in range(N)] for _ in range(T)]


def c2(ref, T=10, N=2000):
    from dynetlsm.sample_latent_positions import sample_latent_positions
    from dynetlsm.network_likelihoods import dynamic_network_loglikelihood_undirected
    from dynetlsm_amd.synthetic import synthetic_lsm_network
    net = synthetic_lsm_network(T, N, 2, density=0.03, seed=0)
    Y, X = net['Y'], net['X_init'].copy()
    b = np.array([net['intercept']])
    rng = np.random.RandomState(1)
    t0 = time.perf_counter()
    sample_latent_positions(Y, X, b, 2.0, 0.1, _samplers(T, N, 0.1), random_state=rng)
    sweep = time.perf_counter() - t0
    t0 = time.perf_counter()
    ll = dynamic_network_loglikelihood_undirected(Y, X, b)
    ev = time.perf_counter() - t0
    it = sweep + 3 * ev
    return dict(config='C2 DynamicNetworkLSM synthetic undirected T=%d N=%d d=2' % (T, N),
                sweep_seconds=round(sweep, 3), loglik_eval_seconds=round(ev, 3),
                iteration_seconds=round(it, 3), it_per_s=round(1.0 / it, 5),
                loglik=float(np.ravel(ll)[0]))


def c3(ref, T=10, N=2000, K=20):
    from dynetlsm.sample_latent_positions import sample_latent_positions_mixture
    from dynetlsm.sample_labels import sample_labels_block
    from dynetlsm.network_likelihoods import dynamic_network_loglikelihood_undirected
    from dynetlsm_amd.synthetic import synthetic_hdp_network
    net = synthetic_hdp_network(T, N, 2, density=0.03, seed=0)
    Y, X = net['Y'], net['X_init'].copy()
    b = np.array([net['intercept']])
    rs = np.random.RandomState(5)
    mu = np.zeros((K, 2)); mu[:6] = net['mu_true']; mu[6:] = 3.0 * rs.randn(K - 6, 2)
    sigma = np.full(K, float(net['sigma_true'].mean()))
    z = net['z_true'].copy()
    w = rs.dirichlet(np.ones(K), size=(T, K))
    lmbda = np.array([0.8])
    rng = np.random.RandomState(1)
    t0 = time.perf_counter()
    sample_latent_positions_mixture(Y, X, b, mu, sigma, lmbda, z, _samplers(T, N, 0.1),
                                    random_state=rng)
    sweep = time.perf_counter() - t0
    t0 = time.perf_counter()
    sample_labels_block(X, mu, sigma, lmbda, w, random_state=rng)
    lab = time.perf_counter() - t0
    t0 = time.perf_counter()
    dynamic_network_loglikelihood_undirected(Y, X, b)
    ev = time.perf_counter() - t0
    it = sweep + lab + 3 * ev
    return dict(config='C3 DynamicNetworkHDPLPCM synthetic T=%d N=%d d=2 K=%d' % (T, N, K),
                sweep_seconds=round(sweep, 3), labels_seconds=round(lab, 3),
                loglik_eval_seconds=round(ev, 3), iteration_seconds=round(it, 3),
                it_per_s=round(1.0 / it, 5),
                note='the O(T K^2) auxiliary / conjugate draws (hdp_lpcm.py:876-1023) are not '
                     'included: lower bound on the iteration')


def c4(ref, T=5, N=10000, C=100):
    from dynetlsm.directed_likelihoods_fast import (approx_directed_partial_loglikelihood,
                                                    approx_directed_network_loglikelihood)
    from test_gpu_full_size import _sparse_directed
    X, radii, degree, in_edges, out_edges = _sparse_directed(T, N, 20, 0)
    rng = np.random.RandomState(2)
    ctrl_in = rng.randint(0, N, size=(T, N, C)).astype(np.int64)
    ctrl_out = rng.randint(0, N, size=(T, N, C)).astype(np.int64)
    n_call = 20000
    nodes = rng.randint(0, N, n_call)
    t0 = time.perf_counter()
    for j in nodes:
        approx_directed_partial_loglikelihood(
            X[0], radii=radii, in_edges=in_edges[0], out_edges=out_edges[0], degree=degree[0],
            control_nodes_in=ctrl_in[0], control_nodes_out=ctrl_out[0], intercept_in=1.0,
            intercept_out=0.5, node_id=int(j), squared=False)
    call = (time.perf_counter() - t0) / n_call
    t0 = time.perf_counter()
    approx_directed_network_loglikelihood(
        X, radii=radii, in_edges=in_edges, out_edges=out_edges, degree=degree,
        control_nodes=ctrl_out, intercept_in=1.0, intercept_out=0.5, squared=False)
    ev = time.perf_counter() - t0
    sweep = 2 * T * N * call
    it = sweep + 7 * ev
    return dict(config='C4 directed case-control T=%d N=%d n_control=%d' % (T, N, C),
                partial_call_us=round(1e6 * call, 3), sweep_seconds_kernel_calls_only=round(sweep, 3),
                loglik_eval_seconds=round(ev, 4), iteration_seconds_lower_bound=round(it, 3),
                it_per_s_upper_bound=round(1.0 / it, 4),
                note='2 T N partial calls (incl. their Python call overhead, not the closure / '
                     'Metropolis overhead of the sweep) + 7 evaluations per iteration')


if __name__ == '__main__':
    from make_golden import import_reference
    ref = import_reference()
    which = sys.argv[1:] or ['c1', 'c2', 'c3', 'c4']
    out = dict(host_cores=os.cpu_count(), threads_used=1,
               where='build container (the reference never travels to the GPU box)')
    for w in which:
        out[w] = globals()[w](ref)
        print(w, json.dumps(out[w]), flush=True)
    path = os.path.join(ROOT, 'profiles', 'r02_reference_timing.json')
    if set(which) >= {'c1', 'c2', 'c3', 'c4'}:
        with open(path, 'w') as f:
            json.dump(out, f, indent=1)
        print('wrote', path)
