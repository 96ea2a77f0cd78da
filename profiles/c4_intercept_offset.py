"""Where does config 4's chain put its intercepts, and why?  (round-5 verdict, weak 2: at T=5 N=10 000 the
case-control chains agree with each other but sit ~10 posterior sd above the generating (0.3, 0.7).)

On ONE network drawn from the model at a size where the exact directed model is affordable (T=5, N=2000 by default)
three pairs of chains, same starting point, same priors, same step-size tuning:
  exact       the directed model itself (k_loglik_directed; directed_likelihoods_fast.pyx:185-205)
  exhaustive  the case-control model with n_control = N - 1: every non-neighbour is a control, the estimator is the
              exact likelihood (directed_likelihoods_fast.pyx:208-270 with weight (N - deg - 1) / n_control = 1)
  cc100       the case-control model with n_control = 100 (config 4's setting)
prints posterior mean / sd / Monte Carlo error of both intercepts per chain.

    python profiles/c4_intercept_offset.py [N [n_burn [n_keep]]]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch  # noqa: F401,E402
from dynetlsm_amd import Chain, SamplerGrid  # noqa: E402
from dynetlsm_amd.synthetic import synthetic_directed_from_model  # noqa: E402
from mcmc_diag import mcse, split_rhat, effective_n  # noqa: E402

T = 5
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n_burn = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
n_keep = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
SEED = 20240229


def run(kind, net, cid):
    w = net['width']
    n_total = 1 + n_burn + n_keep
    rs = np.random.RandomState(1)
    X0 = net['X'] + 0.05 * w * rs.randn(*net['X'].shape)
    model = 'directed' if kind == 'exact' else 'case_control'
    ch = Chain(T, N, 2, model, seed=SEED, chain_id=cid)
    C = None
    if kind == 'exact':
        Y = np.zeros((T, N, N))
        for t in range(T):
            for i in range(N):
                Y[t, i, net['out_edges'][t, i, :net['degree'][t, i, 1]]] = 1.0
        ch.upload_network(Y)
    else:
        C = N - 1 if kind == 'exhaustive' else 100
        ch.upload_edges(net['in_edges'], net['out_edges'], net['degree'])
        ch.resample_controls(0, C)
    ch.set_positions(X0); ch.set_radii(net['radii']); ch.set_intercepts(net['intercepts'])
    ch.set_prior_random_walk(w * w, (0.1 * w) ** 2)
    ch.set_samplers(SamplerGrid(T, N, step_size=0.02 * w, tune=n_burn, tune_interval=100))
    ch.lsm_configure(net['intercepts'], 2.0, step_size_intercept=0.01, tune=n_burn, tune_interval=100,
                     n_iter_procrustes=0, sweep_algo=0, step_size_radii=175000., radii_tune=n_burn,
                     radii_tune_interval=100)
    ch.trace_alloc(n_total, logp0=0.0)
    t0 = time.perf_counter()
    it = 1
    while it < n_total:
        nxt = min(n_total, (it // 100 + 1) * 100)
        if C is not None and it % 100 == 0 and kind != 'exhaustive':
            ch.resample_controls(it, C)
        ch.lsm_run(it, nxt - it, procrustes_ref=0)
        it = nxt
    ch.synchronize()
    secs = time.perf_counter() - t0
    _, ics, lps = ch.trace_read(1 + n_burn, n_keep, positions=False)
    ch.close()
    return ics.copy(), lps.copy(), secs


net = synthetic_directed_from_model(T, N, 20.0, seed=0)
print(json.dumps({'N': N, 'T': T, 'mean_out_degree': float(net['degree'][:, :, 1].mean()), 'width': net['width'],
                  'generating_intercepts': list(net['intercepts']), 'n_burn': n_burn, 'n_keep': n_keep}))
for kind in (sys.argv[4].split(',') if len(sys.argv) > 4 else ('exact', 'exhaustive', 'cc100')):
    tr = [run(kind, net, cid) for cid in (0, 1)]
    b_in = np.stack([t[0][:, 0] for t in tr]); b_out = np.stack([t[0][:, 1] for t in tr])
    out = {'kind': kind, 'seconds_per_chain': [round(t[2], 1) for t in tr]}
    for name, x in (('b_in', b_in), ('b_out', b_out)):
        out[name] = {'mean': [round(float(v.mean()), 4) for v in x], 'sd': [round(float(v.std()), 4) for v in x],
                     'mcse': [round(float(mcse(v, maxlags=2000)), 4) for v in x],
                     'ess': [int(effective_n(v)) for v in x], 'split_rhat': round(float(split_rhat(x)), 3)}
    out['logp_split_rhat'] = round(float(split_rhat(np.stack([t[1] for t in tr]))), 4)
    print(json.dumps(out), flush=True)
