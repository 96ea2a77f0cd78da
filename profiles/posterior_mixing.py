"""How fast the scalar traces of configs 2 / 3 mix (split R-hat, ESS) as the chains get longer:
the calibration run behind tests/test_gpu_posterior_full_size.py.
    python profiles/posterior_mixing.py        (on the GPU box)
"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import dynetlsm_amd as da
from dynetlsm_amd.synthetic import synthetic_lsm_network, synthetic_hdp_network
from mcmc_diag import split_rhat, effective_n, pooled_mean_and_se

T, N, D = 10, 2000, 2
out = {}
net = synthetic_lsm_network(T=T, N=N, D=D, density=0.03, seed=0)
n_burn, n_keep = 1000, 16000
chains = []
buf = None
for cid in (0, 1):
    c = da.Chain(T, N, D, 'undirected', seed=20240229, chain_id=cid)
    if buf is None:
        c.upload_network(net['Y'])
        n = c.network_packed_words(); buf = np.zeros(n, dtype=np.uint32); c.get_network_packed(buf.ctypes.data, n)
    else:
        c.set_network_packed(buf.ctypes.data, buf.size)
    c.set_positions(net['X_init']); c.set_intercepts([net['intercept']])
    c.set_prior_random_walk(2.0, 0.1)
    c.set_samplers(da.SamplerGrid(T, N, step_size=0.1, tune=None))
    c.lsm_configure([net['intercept']], 2.0, step_size_intercept=0.1, tune=n_burn, n_iter_procrustes=0, sweep_algo=0)
    c.trace_alloc(1 + n_burn + n_keep, logp0=0.0)
    c.lsm_run(1, n_burn + n_keep, procrustes_ref=-1)
    chains.append(c)
tr = []
for c in chains:
    c.synchronize()
    _, ics, lps = c.trace_read(1 + n_burn, n_keep, positions=False)
    tr.append((ics[:, 0], lps))
    print('C2 intercept step', c.lsm_get_config().i_step_size[0])
for nk in (3000, 6000, 16000):
    ic = np.stack([t[0][:nk] for t in tr]); lp = np.stack([t[1][:nk] for t in tr])
    print('C2 keep %5d: Rhat ic %.4f lp %.4f ESS ic %s lp %s mean ic %.5f (true %.5f) sd %.5f'
          % (nk, split_rhat(ic), split_rhat(lp), [round(effective_n(x)) for x in ic],
             [round(effective_n(x)) for x in lp], ic.mean(), net['intercept'], ic.std()))
for c in chains:
    c.close()

net = synthetic_hdp_network(T=T, N=N, D=D, density=0.03, seed=0)
K = 20
rs = np.random.RandomState(5)
mu0 = np.zeros((K, D)); mu0[:6] = net['mu_true']; mu0[6:] = 3.0 * rs.randn(K - 6, D)
sg0 = np.full(K, float(net['sigma_true'].mean()))
fits = []
for cid in (0, 1):
    m = da.DynamicNetworkHDPLPCM(n_iter=16000, tune=500, burn=500, n_components=K, n_features=D,
                                 random_state=11 + cid, chain_id=cid, selection_type='map')
    t0 = time.time()
    m.fit(net['Y'], init=dict(X=net['X_init'], intercept=[net['intercept']], mu=mu0, sigma=sg0, z=net['z_true']))
    print('C3 fit %.2f s loop %.2f' % (time.time() - t0, m.loop_seconds_))
    fits.append(m)
nb = fits[0].n_burn_
for nk in (3000, 6000, 16000):
    lam = np.stack([m.lambdas_[nb:nb + nk, 0] for m in fits]); ic = np.stack([m.intercepts_[nb:nb + nk, 0] for m in fits])
    lp = np.stack([m.logps_[nb:nb + nk] for m in fits])
    ncl = np.stack([(m.chain_.post_trace_label_counts(nb, nk) > 0).any(axis=1).sum(axis=1) for m in fits]).astype(float)
    print('C3 keep %5d: Rhat lam %.4f ic %.4f lp %.4f ncl %.4f | ESS lam %s ic %s lp %s | lam %.5f ic %.5f (true %.5f) ncl mean %.3f'
          % (nk, split_rhat(lam), split_rhat(ic), split_rhat(lp), split_rhat(ncl), [round(effective_n(x)) for x in lam],
             [round(effective_n(x)) for x in ic], [round(effective_n(x)) for x in lp], lam.mean(), ic.mean(), net['intercept'], ncl.mean()))
    print('   logp means', lp.mean(axis=1), 'sd', lp.std(axis=1))
