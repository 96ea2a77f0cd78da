"""Per-wavefront entry / exit stamps and hardware ids of the last k_loglik_undirected launch
(engine built with -DDLSM_PIPE_TIMING): how evenly the launch's wavefronts are spread over the
SIMDs and how long the busiest one works.
    python profiles/loglik_timing.py tmp_timing/libtiming.so [out.json]
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynetlsm_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from dynetlsm_amd import Chain, SamplerGrid  # noqa: E402
from dynetlsm_amd.synthetic import synthetic_lsm_network  # noqa: E402

T, N, D = 10, 2000, 2
net = synthetic_lsm_network(T, N, D, density=0.03, seed=0)
ch = Chain(T, N, D, 'undirected', seed=20240229, chain_id=0, device=0)
ch.upload_network(net['Y'])
ch.set_positions(net['X_init'])
ch.set_intercepts([float(net['intercept'])])
ch.set_prior_random_walk(2.0, 0.1)
ch.set_samplers(SamplerGrid(T, N, step_size=0.1, tune=None))
ch.lsm_configure([float(net['intercept'])], 2.0, step_size_intercept=0.1, tune=None,
                 n_iter_procrustes=0, sweep_algo=4)
ch.trace_alloc(64, logp0=0.0)
ch.lsm_run(1, 40, procrustes_ref=0)
ch.synchronize()
L = _lib.load()
w = np.zeros((8192, 3), dtype=np.uint64)
L.dlsm_debug_loglik_timing.restype = C.c_int
L.dlsm_debug_loglik_timing.argtypes = [C.c_void_p]
assert L.dlsm_debug_loglik_timing(w.ctypes.data) == 0
w = w[w[:, 0] > 0].astype(np.int64)
t0 = w[:, 0].min()
ent = (w[:, 0] - t0) * 0.01
ext = (w[:, 1] - t0) * 0.01
hw = w[:, 2] & 0xFFFFFFFF
# HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13] (+ xcc in XCC_ID)
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
xcc = (w[:, 2] >> 32) & 15                          # XCC_ID[3:0]
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
out = {'wavefronts': int(w.shape[0]), 'span_us': float(ext.max()),
       'entry_us_p50_max': [float(np.median(ent)), float(ent.max())],
       'duration_us_p50_max': [float(np.median(ext - ent)), float((ext - ent).max())],
       'exit_us_percentiles_10_50_90_100': [float(np.percentile(ext, q)) for q in (10, 50, 90, 100)]}
cnt = np.bincount(key)
nz = cnt[cnt > 0]
out['simds_seen'] = int(nz.size)
out['waves_per_simd_histogram'] = {int(k): int((nz == k).sum()) for k in np.unique(nz)}
# last exit of a SIMD against its wave count, and the work of its waves (sum of loop trips is
# the same for all: durations differ through sharing only)
last = np.zeros(cnt.size); np.maximum.at(last, key, ext)
out['last_exit_us_by_wave_count'] = {int(k): [round(float(last[cnt == k].min()), 2), round(float(last[cnt == k].max()), 2)]
                                     for k in np.unique(nz)}
print(json.dumps(out))
if len(sys.argv) > 2:
    np.savez_compressed(os.path.splitext(sys.argv[2])[0] + '_raw.npz', w=w)
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
