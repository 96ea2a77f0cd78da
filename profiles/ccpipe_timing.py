"""Phase stamps of the sparse pipelined case-control sweep (k_ccpipe_step, algo 5) at config 4:
resolver workgroups (stamps 0 entry, 2 state and lists loaded + cross sums handed over,
3 first ballot, 4 fixed point reached, 5 exit; 1 / 7 the helper workgroup's entry / announcement) and evaluator wavefronts (entry / exit), last sweep; engine built with
-DDLSM_PIPE_TIMING.
    python profiles/ccpipe_timing.py tmp_timing/libtiming.so [out.json]
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
from dynetlsm_amd.synthetic import synthetic_sparse_directed  # noqa: E402

T, N, Cn = 5, 10000, 100
X, radii, degree, in_edges, out_edges = synthetic_sparse_directed(T, N, 20, 0)
ch = Chain(T, N, 2, 'case_control', seed=20240229, chain_id=0, device=0)
ch.upload_edges(in_edges.astype(np.int64), out_edges.astype(np.int64), degree.astype(np.int64))
ch.resample_controls(0, Cn)
ch.set_positions(X); ch.set_radii(radii); ch.set_intercepts([1.0, 0.5])
ch.set_prior_random_walk(1e-4, 1e-5)
ch.set_samplers(SamplerGrid(T, N, step_size=0.002, tune=None))
ch.lsm_configure([1.0, 0.5], 2.0, step_size_intercept=0.1, tune=None, n_iter_procrustes=0,
                 sweep_algo=5, step_size_radii=175000., radii_tune=None)
ch.trace_alloc(64, logp0=0.0)
ch.lsm_run(1, 30, procrustes_ref=0)
ch.synchronize()
L = _lib.load()
res = np.zeros((32, 16, 8), dtype=np.uint64)
items = np.zeros((32, 4096, 2), dtype=np.uint64)
L.dlsm_debug_ccpipe_timing.restype = C.c_int
L.dlsm_debug_ccpipe_timing.argtypes = [C.c_void_p, C.c_void_p]
assert L.dlsm_debug_ccpipe_timing(res.ctypes.data, items.ctypes.data) == 0
out = []
for l in range(32):
    r = res[l].astype(np.int64); it = items[l].astype(np.int64)
    r = r[r[:, 0] > 0]; it = it[it[:, 0] > 0]
    if r.size == 0 and it.size == 0:
        continue
    t0 = min([x[:, 0].min() for x in (r, it) if x.size])
    row = {'launch': l, 'resolvers': int(r.shape[0]), 'evaluator_wavefronts': int(it.shape[0])}
    if r.size:
        rel = (r[:, :6] - t0) * 0.01
        row['resolver_us_median'] = [round(float(np.median(rel[:, i])), 2) for i in range(6)]
        row['resolver_us_max'] = [round(float(rel[:, i].max()), 2) for i in range(6)]
        row['resolver_fixed_point_passes_wave0'] = [int(v) for v in r[:, 6]]
        if (r[:, 7] > 0).any():         # the cross-sum helpers (ccpipe_cross_helper): entry, sums announced
            hh = r[r[:, 7] > 0]
            row['helper_entry_exit_us_median'] = [round(float(np.median((hh[:, 1] - t0) * 0.01)), 2),
                                                  round(float(np.median((hh[:, 7] - t0) * 0.01)), 2)]
    if it.size:
        rel = (it - t0) * 0.01
        row['evaluator_entry_us_p50_max'] = [round(float(np.median(rel[:, 0])), 2), round(float(rel[:, 0].max()), 2)]
        row['evaluator_exit_us_p50_max'] = [round(float(np.median(rel[:, 1])), 2), round(float(rel[:, 1].max()), 2)]
        row['evaluator_exit_us_p90_p99_p999'] = [round(float(np.percentile(rel[:, 1], q)), 2) for q in (90, 99, 99.9)]
        dur = rel[:, 1] - rel[:, 0]
        row['evaluator_item_us_p50_p90_p99_max'] = [round(float(np.percentile(dur, q)), 2) for q in (50, 90, 99, 100)]
    out.append(row)
    print(json.dumps(row))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
