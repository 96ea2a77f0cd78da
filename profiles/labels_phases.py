"""Phase stamps of the label kernel (k_sample_labels_mfma; DLSM_LABELS_KERNEL=wave: k_sample_labels) at config 3 (engine built with -DDLSM_PIPE_TIMING): per
wavefront (= node) entry, transition matrices staged in LDS, the T x K table built, backward
messages done, forward sampling done.
    python profiles/labels_phases.py tmp_timing/libtiming.so [out.json]
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
w = np.zeros((4096, 6), dtype=np.uint64)
L.dlsm_debug_labels_timing.restype = C.c_int
L.dlsm_debug_labels_timing.argtypes = [C.c_void_p]
assert L.dlsm_debug_labels_timing(w.ctypes.data) == 0
w = w[:, :5].astype(np.int64)
w = w[w[:, 0] != 0]          # one row per wavefront (per node) or per workgroup (16 nodes)
t0 = w[:, 0].min()
rel = (w - t0) * 0.01
names = ['entry', 'w staged', 'table built', 'backward done', 'forward done']
out = {n: {'p50': round(float(np.median(rel[:, i])), 2), 'max': round(float(rel[:, i].max()), 2)}
       for i, n in enumerate(names)}
out['rows'] = int(len(w))
print(json.dumps(out))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
