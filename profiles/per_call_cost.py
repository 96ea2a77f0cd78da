"""What a call of the device-resident loops costs beyond its iterations (round-4 verdict: the driver's
window is ONE call of 20 iterations and reads 3 - 10 % below a 200-iteration call).  Host-clock time of
calls of K = 5 .. 320 iterations between synchronisations, least-squares line t = a + b K per model:
`a` is the per-call cost (launches the first / last iteration does not share, queue fork / join, the
synchronisation's wake-up), `b` the iteration.
    python profiles/per_call_cost.py [lsm] [hdp] [cc]        (on the GPU box)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import dynetlsm_amd as da                                   # noqa: E402

what = sys.argv[1:] or ['lsm', 'hdp', 'cc']
KS = (5, 10, 20, 40, 80, 160, 320)
REPS = 7


def fit_line(run, sync, first):
    """run(first_iteration, count) enqueues; returns per-K medians and the fitted (a, b) in microseconds"""
    it = first
    run(it, 50); it += 50
    sync()
    med = {}
    for K in KS:
        ts = []
        for _ in range(REPS):
            sync()
            t0 = time.perf_counter()
            run(it, K)
            sync()
            ts.append(time.perf_counter() - t0)
            it += K
        med[K] = float(np.median(ts)) * 1e6
    A = np.array([[1.0, K] for K in KS])
    y = np.array([med[K] for K in KS])
    a, b = np.linalg.lstsq(A, y, rcond=None)[0]
    return med, float(a), float(b), it


n_rows = 60 + REPS * sum(KS) + 10

if 'lsm' in what:
    from dynetlsm_amd.synthetic import synthetic_lsm_network
    net = synthetic_lsm_network(T=10, N=2000, D=2, density=0.03, seed=0)
    ch = da.Chain(10, 2000, 2, 'undirected', seed=1, chain_id=0)
    ch.upload_network(net['Y']); ch.set_positions(net['X_init']); ch.set_intercepts([net['intercept']])
    ch.set_prior_random_walk(2.0, 0.1)
    ch.set_samplers(da.SamplerGrid(10, 2000, 0.1, tune=None))
    ch.lsm_configure([net['intercept']], 2.0, step_size_intercept=0.1, tune=None, n_iter_procrustes=0)
    ch.trace_alloc(n_rows, logp0=0.0)
    med, a, b, _ = fit_line(lambda f, c: ch.lsm_run(f, c, procrustes_ref=0), ch.synchronize, 1)
    print(json.dumps({'model': 'lsm C2', 'us_per_call': round(a, 1), 'us_per_iteration': round(b, 2),
                      'it_per_s_asymptotic': round(1e6 / b, 1), 'it_per_s_at_20': round(20e6 / med[20], 1),
                      'median_us_by_K': {k: round(v, 1) for k, v in med.items()}}))
    ch.close()

if 'hdp' in what:
    from dynetlsm_amd.synthetic import synthetic_hdp_network
    net = synthetic_hdp_network(T=10, N=2000, D=2, density=0.03, seed=0)
    rs = np.random.RandomState(5)
    mu0 = np.zeros((20, 2)); mu0[:6] = net['mu_true']; mu0[6:] = 3.0 * rs.randn(14, 2)
    m = da.DynamicNetworkHDPLPCM(n_iter=n_rows, tune=None, burn=None, n_components=20, random_state=1,
                                 selection_type='map')
    m.copy = False
    m._prepare(net['Y'], init=dict(X=net['X_init'], intercept=[net['intercept']], mu=mu0,
                                   sigma=np.full(20, float(net['sigma_true'].mean())), z=net['z_true']))
    med, a, b, _ = fit_line(lambda f, c: m._run(f, c), m.chain_.synchronize, 1)
    print(json.dumps({'model': 'hdp C3', 'queues': m.chain_.hdp_queues(), 'us_per_call': round(a, 1),
                      'us_per_iteration': round(b, 2), 'it_per_s_asymptotic': round(1e6 / b, 1),
                      'it_per_s_at_20': round(20e6 / med[20], 1),
                      'median_us_by_K': {k: round(v, 1) for k, v in med.items()}}))
    m.chain_.close()

if 'cc' in what:
    from dynetlsm_amd.synthetic import synthetic_sparse_directed
    T, N, Cn = 5, 10000, 100
    X, radii, degree, in_edges, out_edges = synthetic_sparse_directed(T, N, 20, 0)
    ch = da.Chain(T, N, 2, 'case_control', seed=20240229, chain_id=0)
    ch.upload_edges(in_edges, out_edges, degree)
    ch.resample_controls(0, Cn)
    ch.set_positions(X); ch.set_radii(radii); ch.set_intercepts([1.0, 0.5])
    ch.set_prior_random_walk(1e-4, 1e-5)
    ch.set_samplers(da.SamplerGrid(T, N, step_size=0.002, tune=None))
    ch.lsm_configure([1.0, 0.5], 2.0, step_size_intercept=0.1, tune=None, n_iter_procrustes=0,
                     step_size_radii=175000., radii_tune=None)
    ch.trace_alloc(n_rows, logp0=0.0)
    med, a, b, _ = fit_line(lambda f, c: ch.lsm_run(f, c, procrustes_ref=0), ch.synchronize, 1)
    print(json.dumps({'model': 'cc C4 (no control resampling inside the calls)', 'us_per_call': round(a, 1),
                      'us_per_iteration': round(b, 2), 'it_per_s_asymptotic': round(1e6 / b, 1),
                      'it_per_s_at_20': round(20e6 / med[20], 1),
                      'median_us_by_K': {k: round(v, 1) for k, v in med.items()}}))
    ch.close()
