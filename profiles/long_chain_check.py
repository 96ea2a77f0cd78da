"""Sanity run of a long device-resident HDP-LPCM chain (6000 iterations, T=5, N=300, K_max=10, 4 true
clusters): finite trace, recovery of the generating blending coefficient / intercept / clustering.
    python profiles/long_chain_check.py      (on the GPU box)
"""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
import torch
import dynetlsm_amd as da
from dynetlsm_amd.synthetic import synthetic_hdp_network
net = synthetic_hdp_network(T=5, N=300, D=2, density=0.08, seed=3, n_clusters=4)
K = 10
rs = np.random.RandomState(1)
mu0 = np.zeros((K, 2)); mu0[:4] = net['mu_true']; mu0[4:] = 3 * rs.randn(K - 4, 2)
t0 = time.time()
m = da.DynamicNetworkHDPLPCM(n_iter=4000, tune=1000, burn=1000, n_components=K, random_state=0,
                             selection_type='map')
m.fit(net['Y'], init=dict(X=net['X_init'], intercept=[net['intercept']], mu=mu0,
                          sigma=np.full(K, 0.1), z=net['z_true']))
print('loop', m.loop_kind_, 'seconds', round(m.loop_seconds_, 2), 'it/s', round(5999 / m.loop_seconds_, 1))
print('finite logps', np.isfinite(m.logps_[1:]).all(), 'lambda mean', m.lambdas_[2000:].mean(),
      'intercept mean', m.intercepts_[2000:].mean(), 'true', net['intercept'])
ncl = np.array([len(np.unique(z)) for z in m.zs_[2000:]])
print('clusters used mean', ncl.mean(), 'min', ncl.min(), 'max', ncl.max())
print('hypers last', m.hypers_[-1])
from sklearn.metrics import adjusted_rand_score
print('ARI of the last sample vs truth', adjusted_rand_score(net['z_true'].ravel(), m.zs_[-1].ravel()))
print('sigma range', m.sigmas_[-1].min(), m.sigmas_[-1].max())
