"""Forecast accumulations at config C3's size on the device, against the
reference's numpy formulation of the same accumulation on a bounded sample.

    python profiles/forecast_timing.py
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import dynetlsm_amd as da                                  # noqa: E402

N, D, S_MC, S_POST = 2000, 2, 5000, 2500
rng = np.random.RandomState(0)
res = dict(config='N=%d d=%d; forecast_probas n_samples=%d; marginal forecast over %d kept samples'
                  % (N, D, S_MC, S_POST))
c = da.Chain(2, N, D, 'undirected')
X = rng.randn(512, N, D)
b = np.full(512, 0.5)
c.forecast_mean_probas(X[:8], b[:8])                        # warm-up
t0 = time.perf_counter()
for _ in range(S_MC // 512 + 1):
    P = c.forecast_mean_probas(X, b, zero_diag=True)
res['device_forecast_probas_accumulation_s'] = time.perf_counter() - t0
t0 = time.perf_counter()
for s in range(4):
    d = np.sqrt(((X[s][:, None] - X[s][None]) ** 2).sum(-1))
    Q = 1 / (1 + np.exp(-(b[s] - d)))
res['host_forecast_probas_accumulation_s_extrapolated'] = (time.perf_counter() - t0) / 4 * S_MC
x = rng.randn(N, D)
W = rng.gamma(1.0, 1.0, size=(S_POST, N))
bb = rng.randn(S_POST) * 0.1
t0 = time.perf_counter()
M = c.forecast_marginal(x, W, bb)
res['device_marginal_forecast_s'] = time.perf_counter() - t0
d = np.sqrt(((x[:, None] - x[None]) ** 2).sum(-1))
t0 = time.perf_counter()
for s in range(4):
    ww = np.outer(W[s], W[s])
    num = ww / (1 + np.exp(-(bb[s] - d)))
res['host_marginal_forecast_s_extrapolated'] = (time.perf_counter() - t0) / 4 * S_POST
c.close()
out = os.path.join(ROOT, 'gpurun_out', 'forecast_timing.json')
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res))
