"""Phase stamps of the persistent sweep (k_pipe_persist, algo 7) at config 2, from an engine
built with -DDLSM_PIPE_TIMING (100 MHz constant clock):

    hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DDLSM_PIPE_TIMING \
          -o tmp_timing/libtiming.so dynetlsm_amd/csrc/capi.hip
    python profiles/persist_timing.py tmp_timing/libtiming.so [out.json]

Resolver (slice t, batch b): wait start, wait end (records of the batch complete, neighbours
final), solve done and stores drained, flag published.  Evaluator workgroup, round r
(wavefront 0): round start, poll matched + barrier, item done and drained; ticket.
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
                 n_iter_procrustes=0, sweep_algo=7)
ch.trace_alloc(64, logp0=0.0)
ch.lsm_run(1, 40, procrustes_ref=0)
ch.synchronize()

L = _lib.load()
res = np.zeros((32, 24, 4), dtype=np.uint64)
ev = np.zeros((256, 24, 5), dtype=np.uint64)
L.dlsm_debug_persist_timing.restype = C.c_int
L.dlsm_debug_persist_timing.argtypes = [C.c_void_p, C.c_void_p]
rc = L.dlsm_debug_persist_timing(res.ctypes.data, ev.ctypes.data)
assert rc == 0, rc
res = res.astype(np.int64); ev = ev.astype(np.int64)
t0 = min(res[res[:, :, 0] > 0][:, 0].min(), ev[ev[:, :, 0] > 0][:, 0].min())
out = {'resolver': [], 'evaluator_rounds': []}
nbat = (N + 127) // 128
print('resolver: per batch, medians over the slices of a parity (us from the launch start): '
      'wait start, wait end, solved, published | wait, solve')
for par in (0, 1):
    for b in range(nbat):
        r = res[par:T:2, b, :]
        rel = (r - t0) * 0.01
        med = np.median(rel, axis=0)
        row = {'parity': par, 'batch': b, 'us': [round(float(x), 2) for x in med],
               'wait_us': round(float(np.median(rel[:, 1] - rel[:, 0])), 2),
               'solve_us': round(float(np.median(rel[:, 2] - rel[:, 1])), 2),
               'publish_us': round(float(np.median(rel[:, 3] - rel[:, 2])), 2)}
        out['resolver'].append(row)
        print(json.dumps(row))
per = np.diff(np.median((res[0:T:2, :nbat, 3] - t0) * 0.01, axis=0))
print('even slices: period between published batches, us:', [round(float(x), 2) for x in per])
out['even_period_us'] = [round(float(x), 2) for x in per]
print('evaluator workgroups: per round medians (us): round start, go, item done | wait, item')
for r in range(24):
    e = ev[:, r, :]
    ok = (e[:, 0] > 0) & (e[:, 4] > 0)
    if not ok.any():
        continue
    rel = (e[ok, :3] - t0) * 0.01
    row = {'round': r, 'workgroups': int(ok.sum()),
           'us': [round(float(x), 2) for x in np.median(rel, axis=0)],
           'wait_us': round(float(np.median(rel[:, 1] - rel[:, 0])), 2),
           'item_us': round(float(np.median(rel[:, 2] - rel[:, 1])), 2),
           'item_us_max': round(float((rel[:, 2] - rel[:, 1]).max()), 2)}
    out['evaluator_rounds'].append(row)
    print(json.dumps(row))
# the item's own phases (pipe_eval_item) and the solve's (pipe_resolve), as profiles/pipe_timing.py reads them
items = np.zeros((24, 4096, 6), dtype=np.uint64)
rres = np.zeros((24, 32, 5), dtype=np.uint64)
L.dlsm_debug_pipe_timing.restype = C.c_int
L.dlsm_debug_pipe_timing.argtypes = [C.c_void_p, C.c_void_p]
assert L.dlsm_debug_pipe_timing(items.ctypes.data, rres.ctypes.data) == 0
items = items.astype(np.int64); rres = rres.astype(np.int64)
print('item phases per round (us after the entry of the item; median / max over wavefronts): first trip done, '
      'last prefetched trip done, record stored, first H operands, exit')
out['item_phases'] = []
for r in range(3, 14):
    it = items[r]
    it = it[(it[:, 0] >= t0) & (it[:, 5] > 0)]
    if not it.size:
        continue
    rel = np.where(it[:, 1:] > 0, (it[:, 1:] - it[:, :1]) * 0.01, np.nan)
    row = {'round': r, 'wavefronts': int(it.shape[0]),
           'median_us': [round(float(np.nanmedian(rel[:, i])), 2) for i in range(5)],
           'max_us': [round(float(np.nanmax(rel[:, i])), 2) for i in range(5)]}
    out['item_phases'].append(row)
    print(json.dumps(row))
print('solve phases per batch (us after entry; median over slices): block + records in LDS, cross block applied, '
      'fixed point, exit')
out['solve_phases'] = []
for b in range(nbat):
    rs = rres[b][:T]
    rs = rs[rs[:, 0] >= t0]
    if not rs.size:
        continue
    rel = (rs[:, 1:] - rs[:, :1]) * 0.01
    row = {'batch': b, 'median_us': [round(float(np.median(rel[:, i])), 2) for i in range(4)],
           'max_us': [round(float(rel[:, i].max()), 2) for i in range(4)]}
    out['solve_phases'].append(row)
    print(json.dumps(row))
end = max(res[:, :, 3].max(), ev[:, :, 2].max())
print('launch span us:', (end - t0) * 0.01)
out['span_us'] = float((end - t0) * 0.01)
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
