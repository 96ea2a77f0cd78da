"""Entry / exit stamps of every workgroup of the six launches behind the label update of the
device-resident HDP-LPCM iteration (stage 1-3, hypers, logp sums, finalize) at config 3: which
role of a multi-role launch is its long pole (engine built with -DDLSM_PIPE_TIMING).
    python profiles/hdp_tail_timing.py tmp_timing/libtiming.so [out.json]
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
from dynetlsm_amd import DynamicNetworkHDPLPCM  # noqa: E402
from dynetlsm_amd.synthetic import synthetic_hdp_network  # noqa: E402

T, N, D, K = 10, 2000, 2, 20
net = synthetic_hdp_network(T=T, N=N, D=D, density=0.03, seed=0)
rs = np.random.RandomState(5)
mu0 = np.zeros((K, D)); mu0[:6] = net['mu_true']; mu0[6:] = 3.0 * rs.randn(K - 6, D)
sg0 = np.full(K, float(net['sigma_true'].mean()))
m = DynamicNetworkHDPLPCM(n_iter=40, tune=None, burn=None, n_components=K, random_state=3,
                          selection_type='map')
m._prepare(net['Y'], init=dict(X=net['X_init'], intercept=[net['intercept']], mu=mu0, sigma=sg0,
                               z=net['z_true']), network_from=None)
m._run(1, 30)
m.chain_.synchronize()
L = _lib.load()
w = np.zeros((6, 512, 2), dtype=np.uint64)
L.dlsm_debug_hdp_tail_timing.restype = C.c_int
L.dlsm_debug_hdp_tail_timing.argtypes = [C.c_void_p]
assert L.dlsm_debug_hdp_tail_timing(w.ctypes.data) == 0
w = w.astype(np.int64)
names = ['stage1', 'stage2', 'stage3', 'hypers', 'logp_sums', 'finalize']
n_tab = (T * K * K + 7) // 8 if False else None
out = {}
for i, nm in enumerate(names):
    a = w[i]
    ok = a[:, 0] > 0
    if not ok.any():
        continue
    t0 = a[ok, 0].min()
    ent = (a[ok, 0] - t0) * 0.01
    ext = (a[ok, 1] - t0) * 0.01
    idx = np.nonzero(ok)[0]
    order = np.argsort(-ext)[:6]
    out[nm] = {'workgroups': int(ok.sum()), 'span_us': round(float(ext.max()), 2),
               'exit_us_p50': round(float(np.median(ext)), 2),
               'slowest_workgroups[(blockIdx, entry, exit)]': [(int(idx[j]), round(float(ent[j]), 2), round(float(ext[j]), 2)) for j in order]}
    print(nm, json.dumps(out[nm]))
# phases of the globals' workgroup of stage 2, relative to its entry
ph = np.zeros((16, 2), dtype=np.uint64)
try:
    L.dlsm_debug_hdp_globals_phases.restype = C.c_int
    L.dlsm_debug_hdp_globals_phases.argtypes = [C.c_void_p]
    if L.dlsm_debug_hdp_globals_phases(ph.ctypes.data) == 0:
        p0 = ph[[0, 2, 3, 4], 0].astype(np.int64)
        key = 'stage2_globals_phases_us[entry, binomials + column sums + pre-drawn variates done, beta drawn, exit]'
        out[key] = [round(float(v - p0[0]) * 0.01, 2) for v in p0]
        print('globals phases', out[key])
except AttributeError:
    pass
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
