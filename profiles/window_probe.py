"""Why does the FIRST 20-step window behind the warm-up read 2 % below the later ones (LSM, config 2)?
Host clock and device events (the handle's timer: events on the chain's stream) around windows that follow
different predecessors.   python profiles/window_probe.py   (on the GPU box)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import dynetlsm_amd as da                                   # noqa: E402
from dynetlsm_amd.synthetic import synthetic_lsm_network    # noqa: E402

net = synthetic_lsm_network(T=10, N=2000, D=2, density=0.03, seed=0)
ch = da.Chain(10, 2000, 2, 'undirected', seed=1, chain_id=0)
ch.upload_network(net['Y']); ch.set_positions(net['X_init']); ch.set_intercepts([net['intercept']])
ch.set_prior_random_walk(2.0, 0.1)
ch.set_samplers(da.SamplerGrid(10, 2000, 0.1, tune=None))
ch.lsm_configure([net['intercept']], 2.0, step_size_intercept=0.1, tune=None, n_iter_procrustes=0)
ch.trace_alloc(4000, logp0=0.0)
it = [1]


def run(k):
    ch.lsm_run(it[0], k, procrustes_ref=0)
    it[0] += k


def window(k, label):
    ch.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    ch.timer_start()
    run(k)
    ms_dev = ch.timer_stop()            # (synchronises on the closing event)
    ch.synchronize(); torch.cuda.synchronize()
    host = time.perf_counter() - t0
    return {'label': label, 'host_it_per_s': round(k / host, 1), 'device_it_per_s': round(k / (ms_dev * 1e-3), 1),
            'host_minus_device_us': round(1e6 * host - 1e3 * ms_dev, 1)}


out = []
run(300); run(5)
out.append(window(20, 'behind 300 + 5 steps (the bench\'s first window)'))
for i in range(3):
    out.append(window(20, 'behind a 20-step window'))
run(5)
out.append(window(20, 'behind 5 steps again'))
ch.synchronize(); time.sleep(0.05)
out.append(window(20, 'behind 50 ms of idle device'))
out.append(window(20, 'behind a 20-step window'))
run(300)
out.append(window(20, 'behind 300 steps'))
out.append(window(200, '200 steps'))
for o in out:
    print(json.dumps(o))
ch.close()
