#!/usr/bin/env python
"""Benchmark of the Gibbs hot path on MI355X: Gibbs iterations/s of
DynamicNetworkLSM on a synthetic undirected network, T=10 N=2000 d=2
(BASELINE.json configs[1]), one independent chain per GPU.

One "step" = one full Gibbs iteration of lsm.py:474-572 for the fully observed
undirected model: latent-position sweep (20 000 MH steps), Procrustes rotation
to the reference sample, centring, intercept MH step and log-posterior trace
(the three full log-likelihood evaluations of the reference fused into one
pass), sample stored in the device-resident trace.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by torch.distributed.run (one rank per GPU).  Rank 0 builds
the network and broadcasts it over RCCL; chains are independent (no
intra-iteration collective); per-chain summaries are all-gathered at the end.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--T', type=int, default=10)
    ap.add_argument('--N', type=int, default=2000)
    ap.add_argument('--D', type=int, default=2)
    ap.add_argument('--density', type=float, default=0.03)
    ap.add_argument('--algo', type=int, default=0, help='sweep algorithm (0 auto)')
    ap.add_argument('--profile-steps', type=int, default=20)
    ap.add_argument('--cpu-iters', type=int, default=8,
                    help='oracle iterations timed for cpu_baseline (0 = skip)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--backend', default=None,
                    help='collective backend (default nccl = RCCL; gloo for dry runs)')
    ap.add_argument('--share-device0', action='store_true',
                    help='dry run: every rank drives cuda:0 (with --backend gloo)')
    return ap.parse_args()


def main():
    args = parse()
    import torch

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if rank == 0:
            print('bench.py: --gpus %d but WORLD_SIZE=%d; launch with '
                  'torch.distributed.run --nproc-per-node %d' % (args.gpus, world, args.gpus),
                  file=sys.stderr)
        sys.exit(2)

    from dynetlsm_amd import Chain, SamplerGrid
    from dynetlsm_amd import _lib
    from dynetlsm_amd.multichain import init_chain_group
    from dynetlsm_amd.synthetic import synthetic_lsm_network

    # one process per GPU; collectives over RCCL (backend "nccl") unless told otherwise
    group = init_chain_group(backend=args.backend or 'nccl') if world > 1 else \
        init_chain_group(backend='gloo')
    if args.share_device0:
        local_rank = 0
    torch.cuda.set_device(local_rank)

    T, N, D = args.T, args.N, args.D
    K, W, P = args.steps, args.warmup, args.profile_steps

    # ---- the network: built on rank 0, broadcast (uint8 on the wire) -------------
    net = synthetic_lsm_network(T, N, D, density=args.density, seed=0) if rank == 0 else None
    Y = group.broadcast_network(net['Y'] if rank == 0 else None)
    X_init = group.broadcast_array(net['X_init'] if rank == 0 else np.zeros((T, N, D)))
    b_init = float(group.broadcast_array(np.array([net['intercept']]) if rank == 0
                                         else np.zeros(1))[0])
    density = float(Y.mean())

    # ---- one chain per rank --------------------------------------------------
    chain = Chain(T, N, D, 'undirected', seed=20240229, chain_id=rank, device=local_rank)
    chain.upload_network(Y)
    chain.set_positions(X_init)
    chain.set_intercepts([b_init])
    chain.set_prior_random_walk(2.0, 0.1)
    chain.set_samplers(SamplerGrid(T, N, step_size=0.1, tune=None))
    chain.lsm_configure([b_init], 2.0, step_size_intercept=0.1, tune=None,
                        n_iter_procrustes=0, sweep_algo=args.algo)
    n_total = 1 + W + K + P
    chain.trace_alloc(n_total, logp0=0.0)

    barrier = group.barrier

    chain.lsm_run(1, W, procrustes_ref=0)
    chain.synchronize()
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    chain.lsm_run(1 + W, K, procrustes_ref=0)
    chain.synchronize()
    torch.cuda.synchronize()
    barrier()
    elapsed = group.max_over_ranks(time.perf_counter() - t0)

    # ---- per-kernel timing by HIP events on the chain's stream --------------
    roofline = None
    extra = {}
    if P > 0:
        chain.profile_enable(True)
        chain.lsm_run(1 + W + K, P, procrustes_ref=0)
        chain.synchronize()
        ms_sw, n_sw = chain.profile_read(_lib.K_SWEEP)
        ms_ll, n_ll = chain.profile_read(_lib.K_LOGLIK)
        ms_ps, n_ps = chain.profile_read(_lib.K_CENTER)
        ms_fi, n_fi = chain.profile_read(_lib.K_FINALIZE)
        chain.profile_enable(False)
        ms_ev, n_ev = chain.profile_read(_lib.K_SWEEP_EVAL)
        ms_rs, n_rs = chain.profile_read(_lib.K_SWEEP_RESOLVE)
        sweep_ms = ms_sw / max(n_sw, 1)
        ll_ms = ms_ll / max(n_ll, 1)
        # algorithmic bytes: a sweep touches every float64 Y entry
        # once (row j of slice t per MH step) + X[t] once per slice; one fused
        # eval reads the upper triangle once + X
        sweep_bytes = 8.0 * T * N * N + 8.0 * T * N * D
        ll_bytes = 8.0 * T * N * (N - 1) / 2 + 8.0 * T * N * D
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if n_ev > 0:
            # dominant kernel of the sweep: k_spec_eval, (n_ev / P) launches per
            # sweep, each covering (slices of one parity) x (batch of <= 128 nodes)
            launches = n_ev / float(P)
            # algo 4: the fused resolve(b) + eval(b + 1) step; algo 2 / 3: the eval kernel
            algo_used = chain.resolve_sweep_algo(args.algo)
            kname = 'k_pipe_step' if algo_used == 4 else 'k_spec_eval'
            k_ms = ms_ev / n_ev
            # the eval launches carry their own start/stop events (hipExtLaunchKernelGGL):
            # the dispatch's begin/end timestamps, which is what the rocprofv3 kernel
            # trace reports
            k_bytes = sweep_bytes / launches
        else:
            launches = 2.0
            kname, k_ms = 'k_sweep_slice', sweep_ms / 2.0
            k_bytes = sweep_bytes / 2.0
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(kname, {}).get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        ach = k_bytes / (k_ms * 1e-3) / 1e9
        roofline = {'bound': 'hbm', 'kernel': kname,
                    'achieved': round(ach, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(ach / HBM_PEAK_GBS, 5), 'traffic': traffic,
                    'us_per_launch': round(1e3 * k_ms, 3),
                    'launches_per_sweep': launches,
                    'algorithmic_bytes_per_launch': round(k_bytes, 1),
                    'sweep_GBps_all_launches': round(sweep_bytes / (sweep_ms * 1e-3) / 1e9, 2),
                    # what actually bounds it (network bit-packed: float64 issue, not HBM):
                    # pairwise terms (distance + exp) evaluated per second over the sweep,
                    # 2 (proposal, current) x T x N x (N - 1) per sweep
                    'dyad_terms_per_s': round(2.0 * T * N * (N - 1) / (sweep_ms * 1e-3), 0)}
        # algo 2 / 3 only: their resolve launches (algo 4 resolves inside the fused launch)
        extra_r = {'us_resolve_per_launch': round(1e3 * ms_rs / n_rs, 3)} if n_rs > 0 else {}
        extra = {'ms_per_loglik_eval': round(ll_ms, 4),
                 'loglik_eval_GBps': round(ll_bytes / (ll_ms * 1e-3) / 1e9, 1),
                 'ms_sweep': round(sweep_ms, 4), 'ms_post_sweep': round(ms_ps / max(n_ps, 1), 4),
                 'ms_finalize': round(ms_fi / max(n_fi, 1), 4)}
        extra.update(extra_r)

    grid_f = chain.get_samplers(SamplerGrid(T, N, 0.1, tune=None))
    acc_rate = float(grid_f.n_accepted.sum()) / max(float(grid_f.n_steps.sum()), 1.0)

    # ---- chain summaries: gather over RCCL ------------------------------------
    _, ics, lps = chain.trace_read(1 + W, K, positions=False)
    summaries = [a.tolist() for a in group.gather_arrays(
        np.array([ics[:, 0].mean(), ics[:, 0].std(), lps.mean(), lps[-1]]))]

    # ---- CPU baseline leg (rank 0): the scalar C oracle timed on this host's cores
    #      on a bounded sample of the same workload; the same leg checks the engine's
    #      log-likelihood at the chain's final state against the oracle ---------------
    cpu = None
    if rank == 0 and not args.no_cpu and args.cpu_iters > 0:
        from oracle import oracle as orc
        Xf = chain.get_positions()
        bf = chain.get_intercepts()[0]
        g = chain.loglik_full([[bf]])[0]
        o = orc.dynamic_network_loglikelihood_undirected(Y, Xf, bf)
        og = orc.SamplerGrid(T, N, 0.1, tune=None)
        st = orc.ChainState(X_init, og, Y=Y, intercept=[b_init], tau_sq=2.0,
                            sigma_sq=0.1, seed=20240229, chain=0)
        isamp = orc.ScalarSampler(0.1, 0, 0, 100, -1, 100)
        tc = time.perf_counter()
        for it in range(1, args.cpu_iters + 1):
            st.c.iter = it
            orc.lsm_iteration_undirected(st, isamp, b_init, 2.0)
        tc = time.perf_counter() - tc
        cpu = {'value': round(args.cpu_iters / tc, 5), 'unit': 'Gibbs iterations/s',
               'cores': 1, 'kind': 'port',
               'sample': '%d iterations of the same T=%d N=%d d=%d workload by the '
                         'scalar C oracle (sweep + 2 full log-lik evals per iteration), '
                         '%.1f s' % (args.cpu_iters, T, N, D, tc),
               'engine_loglik_rel_err_vs_oracle': abs(g - o) / abs(o)}

    if rank == 0:
        value = world * K / elapsed
        line = {
            'metric': 'Gibbs iterations/sec (and ms/log-lik eval), T=10 N=2000 d=2',
            'value': round(value, 3), 'unit': 'Gibbs iterations/s', 'n_gpus': world,
            'steps': K, 'warmup': W, 'ms_per_step': round(1e3 * elapsed / K, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'DynamicNetworkLSM synthetic undirected T=%d N=%d d=%d, '
                                   '1 chain per GPU' % (T, N, D),
                       'density': round(density, 4), 'chains': world,
                       'iteration': 'sweep + procrustes + centring + intercept MH + logp trace',
                       'sweep_algo': chain.resolve_sweep_algo(args.algo),
                       'mh_acceptance_rate': round(acc_rate, 3)},
            'roofline': roofline, 'cpu_baseline': cpu,
            'chain_summaries[intercept_mean,intercept_sd,logp_mean,logp_last]': summaries,
        }
        line.update(extra)
        print(json.dumps(line))
    chain.close()
    group.close()


if __name__ == '__main__':
    main()
