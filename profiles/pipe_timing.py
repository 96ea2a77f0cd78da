"""Where the time of a pipelined-sweep launch (k_pipe_step, algo 4) goes, from in-kernel phase
stamps: an engine built with -DDLSM_PIPE_TIMING records the 100 MHz constant clock
(s_memrealtime) at the phase boundaries of every evaluator wavefront and resolver workgroup of
the last sweep; this script runs config 2 on it and prints, per launch, the medians over
wavefronts relative to the launch's earliest stamp.

    hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DDLSM_PIPE_TIMING \
          -o tmp_timing/libtiming.so dynetlsm_amd/csrc/capi.hip
    python profiles/pipe_timing.py tmp_timing/libtiming.so [out.json [sweep algo: 4 | 6]]

Evaluator stamps (round 6, kernels_pipe_lds.hpp; reported in this order): kernel start (the wavefront's first
instruction), rows + table staged (behind the workgroup's barrier; a serving wavefront: behind its cross product),
first trip of 64 neighbours done, last trip done, record stored (= exit: the H factors left inside their trips).
(DLSM_PIPE_LDS=0, pipe_eval_item: slot order entry behind the table's barrier, first trip, last prefetched trip,
record stored, kernel start, exit.)  Resolver stamps: 0 entry, 1 H block + records in LDS, 2 cross products
applied, 3 fixed point reached, 4 exit.
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch  # noqa: F401  (its HIP runtime first, see dynetlsm_amd/_lib.py)

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
                 n_iter_procrustes=0, sweep_algo=int(sys.argv[3]) if len(sys.argv) > 3 else 4)
ch.trace_alloc(64, logp0=0.0)
ch.lsm_run(1, 40, procrustes_ref=0)
ch.synchronize()

L = _lib.load()
items = np.zeros((24, 4096, 6), dtype=np.uint64)
res = np.zeros((24, 32, 5), dtype=np.uint64)
L.dlsm_debug_pipe_timing.restype = C.c_int
L.dlsm_debug_pipe_timing.argtypes = [C.c_void_p, C.c_void_p]
rc = L.dlsm_debug_pipe_timing(items.ctypes.data, res.ctypes.data)
assert rc == 0, rc

out = []
for l in range(24):
    it = items[l].astype(np.int64)
    rs = res[l].astype(np.int64)
    it = it[it[:, 0] > 0]
    rs = rs[rs[:, 0] > 0]
    if it.size == 0 and rs.size == 0:
        continue
    t0 = min([x[:, 0].min() for x in (it, rs) if x.size])
    row = {'launch': l, 'evaluator_wavefronts': int(it.shape[0]), 'resolver_workgroups': int(rs.shape[0])}
    if it.size:
        if not os.environ.get('DLSM_PIPE_LDS') == '0':
            it = it[:, [0, 4, 1, 2, 3, 5]]          # the LDS evaluators' slots in time order
        # a stamp of 0 -> nan
        rel = np.where(it > 0, (it - t0) * 0.01, np.nan)
        row['evaluator_us_median'] = [round(float(np.nanmedian(rel[:, i])), 2) for i in range(6)]
        row['evaluator_us_max'] = [round(float(np.nanmax(rel[:, i])), 2) for i in range(6)]
        row['evaluator_us_min'] = [round(float(np.nanmin(rel[:, i])), 2) for i in range(6)]
    if rs.size:
        rel = (rs - t0) * 0.01
        row['resolver_us_median'] = [round(float(np.median(rel[:, i])), 2) for i in range(5)]
        row['resolver_us_max'] = [round(float(rel[:, i].max()), 2) for i in range(5)]
    ends = [x[:, -1].max() for x in (it, rs) if x.size]
    row['span_us'] = round(float((max(ends) - t0) * 0.01), 2)
    out.append(row)
    print(json.dumps(row))
# per wavefront: how long each phase takes, by the wavefront's rank on its SIMD (wavefronts w, w + 4,
# w + 8, w + 12 of a workgroup share SIMD w % 4: rank = w / 4)
it = items[8].astype(np.int64)
ok = it[:, 0] > 0
t0 = it[ok, 0].min()
names = ['entry', 'first trip done', 'last prefetched trip done', 'record stored', 'H operands here', 'exit']
if not os.environ.get('DLSM_PIPE_LDS') == '0':
    names = ['kernel start', 'rows + table staged', 'first trip done', 'last trip done', 'record stored', '= exit']
    it = it[:, [0, 4, 1, 2, 3, 5]]
for rank in range(4):
    sel = ok & ((np.arange(4096) % 16) // 4 == rank)
    rel = np.where(it[sel] > 0, (it[sel] - t0) * 0.01, np.nan)
    dur = np.diff(rel, axis=1)
    print('launch 8, SIMD rank %d: stamps us (median) %s ; phase durations us (median) %s'
          % (rank, [round(float(np.nanmedian(rel[:, i])), 2) for i in range(6)],
             [round(float(np.nanmedian(dur[:, i])), 2) for i in range(5)]))
    out.append({'launch8_simd_rank': rank, 'stamps_us_median': [round(float(np.nanmedian(rel[:, i])), 2) for i in range(6)],
                'phase_us_median': [round(float(np.nanmedian(dur[:, i])), 2) for i in range(5)], 'phases': names})
# the dispatch ramp of one full launch: entry and exit of the workgroups in blockIdx order
it = items[8].astype(np.int64)
ok = it[:, 0] > 0
t0 = it[ok, 0].min()
wg_entry = [round(float((it[w * 16:(w + 1) * 16, 0][ok[w * 16:(w + 1) * 16]].mean() - t0) * 0.01), 2)
            for w in range(4096 // 16) if ok[w * 16:(w + 1) * 16].any()]
wg_exit = [round(float((it[w * 16:(w + 1) * 16, 5][ok[w * 16:(w + 1) * 16]].max() - t0) * 0.01), 2)
           for w in range(4096 // 16) if ok[w * 16:(w + 1) * 16].any()]
print('launch 8, evaluator workgroups in blockIdx order: mean entry us')
print(wg_entry)
print('launch 8, evaluator workgroups in blockIdx order: last exit us')
print(wg_exit)
out.append({'launch8_workgroup_entry_us': wg_entry, 'launch8_workgroup_exit_us': wg_exit})
if len(sys.argv) > 2:
    np.savez_compressed(os.path.splitext(sys.argv[2])[0] + '_raw.npz', items=items, res=res)
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
