"""Round 5 exploration behind tests/test_gpu_cold_start_full_size.py and tests/test_gpu_c4_full_size.py:
how long do cold-start chains (fit(Y) without init=, lsm.py:386-407 / hdp_lpcm.py:48-141) need at
BASELINE's sizes, what do they recover, and how do two case-control chains at config 4 mix?
    python profiles/posterior_cold_start.py [c3] [c2] [c4]      (on the GPU box)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: F401,E402
import dynetlsm_amd as da                                   # noqa: E402
from mcmc_diag import effective_n, split_rhat               # noqa: E402

what = sys.argv[1:] or ['c3', 'c2', 'c4']


def aligned_rms(X, X_true):
    """RMS distance after the best common rotation / reflection + shift over all (t, i)"""
    A = X.reshape(-1, X.shape[-1]) - X.reshape(-1, X.shape[-1]).mean(axis=0)
    B = X_true.reshape(-1, X.shape[-1]) - X_true.reshape(-1, X.shape[-1]).mean(axis=0)
    U, _, Vt = np.linalg.svd(A.T @ B)
    return float(np.sqrt(((A @ (U @ Vt) - B) ** 2).sum(axis=1).mean())), float(np.sqrt((B ** 2).sum(axis=1).mean()))


def windows(tr, name, nb):
    out = {}
    n = tr.shape[1]
    for k in (2000, 4000, 8000, 16000):
        if nb + k <= n:
            out['%s_rhat_%d' % (name, k)] = round(split_rhat(tr[:, nb:nb + k]), 4)
    out['%s_ess' % name] = [round(effective_n(x[nb:]), 0) for x in tr]
    out['%s_mean' % name] = [round(float(x[nb:].mean()), 5) for x in tr]
    out['%s_sd' % name] = [round(float(x[nb:].std()), 5) for x in tr]
    return out


if 'c3' in what:
    from sklearn.metrics import adjusted_rand_score
    from dynetlsm_amd.synthetic import synthetic_hdp_network
    net = synthetic_hdp_network(T=10, N=2000, D=2, density=0.03, seed=0)
    n_iter, tune, burn = 16000, 2500, 2500
    fits = []
    for kind in ('cold', 'truth'):
        m = da.DynamicNetworkHDPLPCM(n_iter=n_iter, tune=tune, burn=burn, n_components=20,
                                     random_state=3 if kind == 'cold' else 4,
                                     chain_id=0 if kind == 'cold' else 1)
        t0 = time.perf_counter()
        if kind == 'cold':
            m.fit(net['Y'])
        else:
            rs = np.random.RandomState(5)
            mu0 = np.zeros((20, 2)); mu0[:6] = net['mu_true']; mu0[6:] = 3.0 * rs.randn(14, 2)
            m.fit(net['Y'], init=dict(X=net['X_init'], intercept=[net['intercept']], mu=mu0,
                                      sigma=np.full(20, float(net['sigma_true'].mean())), z=net['z_true']))
        secs = time.perf_counter() - t0
        fits.append(m)
        rms, scale = aligned_rms(m.X_, net['X_true'])
        nk = np.bincount(m.z_.ravel(), minlength=20)
        print(json.dumps(dict(config='C3 fit', start=kind, seconds=round(secs, 2), loop=round(m.loop_seconds_, 2),
                              ari=round(adjusted_rand_score(net['z_true'].ravel(), m.z_.ravel()), 4),
                              clusters_ge_1pct=int((nk >= 0.01 * nk.sum()).sum()),
                              lambda_mean=float(m.lambda_mean_[0]), X_rms=rms, X_scale=scale,
                              intercept_mean=float(np.ravel(m.intercept_)[0]), generating=net['intercept'])))
    nb = fits[0].n_burn_
    out = {'config': 'C3 cold vs truth start, split R-hat over the first k kept iterations'}
    for name in ('lambdas_', 'intercepts_', 'logps_'):
        tr = np.stack([np.asarray(getattr(m, name)).reshape(m.logps_.shape[0], -1)[:, 0] for m in fits])
        out.update(windows(tr, name, nb))
    print(json.dumps(out))
    for m in fits:
        m.chain_.close()

if 'c2' in what:
    from dynetlsm_amd.synthetic import synthetic_lsm_network
    net = synthetic_lsm_network(T=10, N=2000, D=2, density=0.03, seed=0)
    n_iter, tune, burn = 12000, 2000, 2000
    traces = []
    for kind in ('cold', 'truth'):
        m = da.DynamicNetworkLSM(n_iter=n_iter, tune=tune, burn=burn, random_state=3 if kind == 'cold' else 4,
                                 chain_id=0 if kind == 'cold' else 1)
        t0 = time.perf_counter()
        if kind == 'cold':
            m.fit(net['Y'])
        else:
            m.fit(net['Y'], init=dict(X=net['X_init'], intercept=[net['intercept']]))
        secs = time.perf_counter() - t0
        nb = m.n_burn_
        rms, scale = aligned_rms(m.X_, net['X_true'])
        rms_mean, _ = aligned_rms(m.Xs_[nb:].mean(axis=0), net['X_true'])
        rms0, _ = aligned_rms(m.Xs_[0], net['X_true'])
        print(json.dumps(dict(config='C2 fit', start=kind, seconds=round(secs, 2), loop=round(m.loop_seconds_, 2),
                              X_rms_map=rms, X_rms_mean=rms_mean, X_rms_start=rms0, X_scale=scale,
                              intercept_start=float(m.intercepts_[0, 0]),
                              intercept_map=float(np.ravel(m.intercept_)[0]), generating=net['intercept'])))
        traces.append((m.intercepts_[:, 0].copy(), m.logps_.copy()))
        m.chain_.close()
        del m
    out = {'config': 'C2 cold vs truth start'}
    out.update(windows(np.stack([t[0] for t in traces]), 'intercepts_', nb))
    out.update(windows(np.stack([t[1] for t in traces]), 'logps_', nb))
    print(json.dumps(out))

if 'c4' in what:
    from dynetlsm_amd.synthetic import synthetic_sparse_directed
    T, N, Cn = 5, 10000, 100
    X, radii, degree, in_edges, out_edges = synthetic_sparse_directed(T, N, 20, 0)
    n_burn, n_keep, n_res = 4000, 16000, 100
    chains = []
    t0 = time.perf_counter()
    for cid in (0, 1):
        ch = da.Chain(T, N, 2, 'case_control', seed=20240229, chain_id=cid)
        ch.upload_edges(in_edges, out_edges, degree)
        ch.resample_controls(0, Cn)
        ch.set_positions(X); ch.set_radii(radii); ch.set_intercepts([1.0, 0.5])
        ch.set_prior_random_walk(1e-4, 1e-5)
        ch.set_samplers(da.SamplerGrid(T, N, step_size=0.002, tune=n_burn, tune_interval=100))
        ch.lsm_configure([1.0, 0.5], 2.0, step_size_intercept=0.1, tune=n_burn, tune_interval=100,
                         n_iter_procrustes=0, sweep_algo=0, step_size_radii=175000., radii_tune=n_burn,
                         radii_tune_interval=100)
        ch.trace_alloc(1 + n_burn + n_keep, logp0=0.0)
        chains.append(ch)
    it = 1
    while it <= n_burn + n_keep:
        nxt = min(n_burn + n_keep + 1, (it // n_res + 1) * n_res)
        for ch in chains:
            if it % n_res == 0:
                ch.resample_controls(it, Cn)
            ch.lsm_run(it, nxt - it, procrustes_ref=0)
        it = nxt
    for ch in chains:
        ch.synchronize()
    secs = time.perf_counter() - t0
    tr = [ch.trace_read(0, 1 + n_burn + n_keep, positions=False) for ch in chains]
    out = {'config': 'C4 two chains', 'seconds': round(secs, 2)}
    out.update(windows(np.stack([t[1][:, 0] for t in tr]), 'b_in', 1 + n_burn))
    out.update(windows(np.stack([t[1][:, 1] for t in tr]), 'b_out', 1 + n_burn))
    out.update(windows(np.stack([t[2] for t in tr]), 'logp', 1 + n_burn))
    for ch in chains:
        g = ch.get_samplers(da.SamplerGrid(T, N, 0.002, tune=None))
        cfg = ch.lsm_get_config()
        out.setdefault('acc', []).append(round(float(g.n_accepted.sum()) / max(1.0, float(g.n_steps.sum())), 3))
        out.setdefault('steps', []).append([round(float(np.median(g.step_size)), 6), float(cfg.i_step_size[0]),
                                            float(cfg.i_step_size[1]), float(cfg.r_step_size)])
    # by 2000-iteration blocks: where the chains are
    b_in = np.stack([t[1][:, 0] for t in tr]); lp = np.stack([t[2] for t in tr])
    out['b_in_block_means'] = [[round(float(x[i:i + 2000].mean()), 4) for i in range(1, x.shape[0] - 1, 2000)] for x in b_in]
    out['logp_block_means'] = [[round(float(x[i:i + 2000].mean()), 1) for i in range(1, x.shape[0] - 1, 2000)] for x in lp]
    print(json.dumps(out))
    for ch in chains:
        ch.close()

if 'c4m' in what:
    # config 4's size with a network drawn FROM the model (synthetic_directed_from_model), chains
    # started at the generating values + noise: do two chains agree, do they stay at the truth?
    from dynetlsm_amd.synthetic import synthetic_directed_from_model
    T, N, Cn = 5, 10000, 100
    t0 = time.perf_counter()
    net = synthetic_directed_from_model(T, N, 20.0, seed=0)
    print(json.dumps(dict(config='C4 model network', seconds=round(time.perf_counter() - t0, 2), width=net['width'],
                          mean_degree=float(net['degree'][:, :, 1].mean()), Din=int(net['in_edges'].shape[2]),
                          Dout=int(net['out_edges'].shape[2]))))
    w = net['width']
    n_burn, n_keep, n_res = 4000, 16000, 100
    chains = []
    rs = np.random.RandomState(1)
    X0 = net['X'] + 0.05 * w * rs.randn(*net['X'].shape)
    t0 = time.perf_counter()
    for cid in (0, 1):
        ch = da.Chain(T, N, 2, 'case_control', seed=20240229, chain_id=cid)
        ch.upload_edges(net['in_edges'], net['out_edges'], net['degree'])
        ch.resample_controls(0, Cn)
        ch.set_positions(X0); ch.set_radii(net['radii']); ch.set_intercepts(net['intercepts'])
        ch.set_prior_random_walk(w * w, (0.1 * w) ** 2)
        ch.set_samplers(da.SamplerGrid(T, N, step_size=0.02 * w, tune=n_burn, tune_interval=100))
        ch.lsm_configure(net['intercepts'], 2.0, step_size_intercept=0.01, tune=n_burn, tune_interval=100,
                         n_iter_procrustes=0, sweep_algo=0, step_size_radii=175000., radii_tune=n_burn,
                         radii_tune_interval=100)
        ch.trace_alloc(1 + n_burn + n_keep, logp0=0.0)
        chains.append(ch)
    it = 1
    while it <= n_burn + n_keep:
        nxt = min(n_burn + n_keep + 1, (it // n_res + 1) * n_res)
        for ch in chains:
            if it % n_res == 0:
                ch.resample_controls(it, Cn)
            ch.lsm_run(it, nxt - it, procrustes_ref=0)
        it = nxt
    for ch in chains:
        ch.synchronize()
    secs = time.perf_counter() - t0
    tr = [ch.trace_read(0, 1 + n_burn + n_keep, positions=False) for ch in chains]
    out = {'config': 'C4 model network, two chains', 'seconds': round(secs, 2), 'generating': [0.3, 0.7]}
    out.update(windows(np.stack([t[1][:, 0] for t in tr]), 'b_in', 1 + n_burn))
    out.update(windows(np.stack([t[1][:, 1] for t in tr]), 'b_out', 1 + n_burn))
    out.update(windows(np.stack([t[2] for t in tr]), 'logp', 1 + n_burn))
    for ch in chains:
        g = ch.get_samplers(da.SamplerGrid(T, N, 0.002, tune=None))
        cfg = ch.lsm_get_config()
        out.setdefault('acc', []).append(round(float(g.n_accepted.sum()) / max(1.0, float(g.n_steps.sum())), 3))
        out.setdefault('steps', []).append([float(np.median(g.step_size)), float(cfg.i_step_size[0]),
                                            float(cfg.i_step_size[1]), float(cfg.r_step_size),
                                            int(cfg.i_n_accepted[0]), int(cfg.i_n_accepted[1]), int(cfg.r_n_accepted)])
        rad = ch.trace_read_radii(n_burn + n_keep, 1)[0]
        out.setdefault('radii_rel_rms', []).append(float(np.sqrt(np.mean((rad / net['radii'] - 1) ** 2))))
        Xl = ch.get_positions()
        out.setdefault('X_rms_over_width', []).append(aligned_rms(Xl, net['X'])[0] / w)
    b_in = np.stack([t[1][:, 0] for t in tr]); b_out = np.stack([t[1][:, 1] for t in tr]); lp = np.stack([t[2] for t in tr])
    out['b_in_block_means'] = [[round(float(x[i:i + 2000].mean()), 4) for i in range(1, x.shape[0] - 1, 2000)] for x in b_in]
    out['b_out_block_means'] = [[round(float(x[i:i + 2000].mean()), 4) for i in range(1, x.shape[0] - 1, 2000)] for x in b_out]
    out['logp_block_means'] = [[round(float(x[i:i + 2000].mean()), 1) for i in range(1, x.shape[0] - 1, 2000)] for x in lp]
    print(json.dumps(out))
    for ch in chains:
        ch.close()

if 'c4long' in what:
    # the same two chains over 100 000 iterations, in segments that carry the state over into a fresh
    # handle (a 5000-row positions trace is 4 GB at this size; the segment's Philox chain id keys its draws)
    from dynetlsm_amd.synthetic import synthetic_directed_from_model
    T, N, Cn = 5, 10000, 100
    net = synthetic_directed_from_model(T, N, 20.0, seed=0)
    w = net['width']
    seg, n_seg, n_res, n_tune_seg = 5000, 20, 100, 2
    rs = np.random.RandomState(1)
    X0 = net['X'] + 0.05 * w * rs.randn(*net['X'].shape)
    state = [dict(X=X0, radii=net['radii'], b=net['intercepts'], grid=da.SamplerGrid(T, N, 0.02 * w, tune=seg, tune_interval=100),
                  ist=None, rstep=175000.) for _ in range(2)]
    traces = [[], []]
    t0 = time.perf_counter()
    for s in range(n_seg):
        tuning = s < n_tune_seg
        chains = []
        for cid in (0, 1):
            st = state[cid]
            ch = da.Chain(T, N, 2, 'case_control', seed=20240229, chain_id=cid + 2 * s)
            ch.upload_edges(net['in_edges'], net['out_edges'], net['degree'])
            ch.resample_controls(0, Cn)
            ch.set_positions(st['X']); ch.set_radii(st['radii']); ch.set_intercepts(st['b'])
            ch.set_prior_random_walk(w * w, (0.1 * w) ** 2)
            g = st['grid']
            g.tune = seg if tuning else None
            g.n_accepted[:] = 0; g.n_steps[:] = 0; g.steps_until_tune[:] = 100
            ch.set_samplers(g)
            ch.lsm_configure(net['intercepts'], 2.0, step_size_intercept=0.01, tune=seg if tuning else None,
                             tune_interval=100, n_iter_procrustes=0, sweep_algo=0, step_size_radii=st['rstep'],
                             radii_tune=seg if tuning else None, radii_tune_interval=100, state=st['ist'])
            ch.trace_alloc(1 + seg, logp0=0.0)
            chains.append(ch)
        it = 1
        while it <= seg:
            nxt = min(seg + 1, (it // n_res + 1) * n_res)
            for ch in chains:
                if it % n_res == 0:
                    ch.resample_controls(it, Cn)
                ch.lsm_run(it, nxt - it, procrustes_ref=0)
            it = nxt
        for cid, ch in enumerate(chains):
            ch.synchronize()
            _, ics, lps = ch.trace_read(1, seg, positions=False)
            traces[cid].append((ics.copy(), lps.copy()))
            cfg = ch.lsm_get_config()
            st = state[cid]
            st['X'] = ch.get_positions(); st['radii'] = ch.get_radii(); st['b'] = ics[-1].copy()
            st['grid'] = ch.get_samplers(st['grid'])
            st['ist'] = [(cfg.i_step_size[k], 0, 0, 100) for k in range(2)]
            st['rstep'] = float(cfg.r_step_size)
            ch.close()
    secs = time.perf_counter() - t0
    b_in = np.stack([np.concatenate([x[0][:, 0] for x in tr]) for tr in traces])
    b_out = np.stack([np.concatenate([x[0][:, 1] for x in tr]) for tr in traces])
    lp = np.stack([np.concatenate([x[1] for x in tr]) for tr in traces])
    out = {'config': 'C4 model network, 2 chains x %d iterations in segments of %d' % (seg * n_seg, seg),
           'seconds': round(secs, 1)}
    for nb in (10000, 20000, 40000):
        for k in (20000, 40000, 60000, 80000):
            if nb + k <= seg * n_seg:
                out['rhat_burn%d_keep%d' % (nb, k)] = [round(split_rhat(x[:, nb:nb + k]), 4) for x in (b_in, b_out, lp)]
    out['ess_after_20000'] = [[round(effective_n(x[20000:]), 0) for x in v] for v in (b_in, b_out, lp)]
    out['b_in_block_means'] = [[round(float(x[i:i + 10000].mean()), 4) for i in range(0, x.shape[0], 10000)] for x in b_in]
    out['b_out_block_means'] = [[round(float(x[i:i + 10000].mean()), 4) for i in range(0, x.shape[0], 10000)] for x in b_out]
    out['b_sd_after_20000'] = [round(float(b_in[:, 20000:].std()), 5), round(float(b_out[:, 20000:].std()), 5)]
    print(json.dumps(out))
