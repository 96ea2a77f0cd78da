"""Wall time of a whole DynamicNetworkHDPLPCM.fit(Y) - starting values, Gibbs loop, model selection,
Procrustes alignment - at config 3's size, by phase.
    python profiles/end_to_end_fit.py [n_iter]      (on the GPU box)
"""
import cProfile
import json
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import dynetlsm_amd as da                                   # noqa: E402
from dynetlsm_amd.synthetic import synthetic_hdp_network    # noqa: E402

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
net = synthetic_hdp_network(T=10, N=2000, D=2, density=0.03, seed=0)
m = da.DynamicNetworkHDPLPCM(n_iter=n_iter, tune=n_iter // 2, burn=n_iter // 2, n_components=20,
                             random_state=0)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
m.fit(net['Y'])
pr.disable()
total = time.perf_counter() - t0
st = pstats.Stats(pr)
rows = {}
for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
    if name in ('_prepare', '_run', '_finish', '_init_sampler', 'select_model', 'procrustes_align_samples',
                '_pull', 'generalized_mds', 'longitudinal_kmeans', 'posterior_group_counts'):
        rows[name] = round(ct, 3)
out = dict(config='DynamicNetworkHDPLPCM(n_iter=%d, tune=%d, burn=%d, n_components=20).fit(Y), T=10 N=2000'
                  % (n_iter, n_iter // 2, n_iter // 2), total_iterations=m.logps_.shape[0],
           total_seconds=round(total, 2), gibbs_loop_seconds=round(m.loop_seconds_, 3),
           cumulative_seconds_by_function=rows, n_clusters_selected=int(len(np.unique(m.z_))),
           lambda_mean=float(m.lambda_mean_[0]))
print(json.dumps(out))
