#!/usr/bin/env python
"""Benchmark of the Gibbs hot path on MI355X, one independent chain per GPU.

    python bench.py --gpus N --steps K --warmup W [--model lsm|hdp|all]

Headline (``--model lsm``, BASELINE.json configs[1]): Gibbs iterations/s of
DynamicNetworkLSM on a synthetic undirected network, T=10 N=2000 d=2.  One "step" =
one full Gibbs iteration of lsm.py:474-572: latent-position sweep (20 000 MH steps),
Procrustes rotation, centring, intercept MH step and log-posterior trace (the three
full log-likelihood evaluations of the reference fused into one pass), sample stored
in the device-resident trace.

``--model hdp`` (configs[2] on one GPU, configs[4] = 8 chains on 8 GPUs): one step =
one Gibbs iteration of DynamicNetworkHDPLPCM._fit (hdp_lpcm.py:823-1069), K_max = 20:
sweep with the AR-mixture prior, centring, intercept MH, label block update, the
HDP's auxiliary / conjugate / hyper-parameter draws and the log-posterior trace.

``--model cc`` (configs[3]): one step = one Gibbs iteration of the directed case-control
DynamicNetworkLSM (T=5 N=10 000, 100 controls): sweep over the case-control partial
likelihoods, centring, the two intercept steps and the radii step.

``--model all`` (the default) prints the LSM line as the headline - the metric of
BASELINE.json - and attaches the HDP-LPCM and case-control measurements of the same run as
``extra_configs`` (so that the N-GPU runs exercise configs[4] too).  ``--chains-per-gpu C``
runs C independent chains per GPU, each on its own handle, stream and host thread.

N > 1: when RANK is not set the process spawns the N ranks itself (it makes no GPU
call before that), relays rank 0's JSON line and exits non-zero if a rank fails;
under ``torch.distributed.run`` the ranks are the launcher's.  Rank 0 uploads and
packs the network once and broadcasts the packed words device to device over RCCL;
chains are independent (no intra-iteration collective); posterior-mean positions,
log-posterior and intercept traces are all-gathered at the end.  Prints ONE JSON line
on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# float64 vector peak: v_fma_f64 issues at half the FP32 vector rate (157.3 TFLOP/s spec in
# MI355X_MICROARCH.md) = 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz.  A pure fma stream at
# the sweep's occupancy (4 wavefronts per SIMD) issues one wavefront-instruction per 4.15
# cycles at the 2.05 GHz the chip holds under it = 82 % of nominal
# (profiles/micro/valu_rates.hip; one wavefront per SIMD: 4.57 cycles, the 76 % of round 1).
F64_VALU_PEAK_TFLOPS = 78.6
# the f64 matrix instructions run at the vector rate on this part (v_mfma_f64_16x16x4_f64:
# 2048 flop in 64 cycles per SIMD)
FP64_MATRIX_PEAK_TFLOPS = 78.6
# gathers of records that stay in L2: a CU's vector L1 looks up one cache line per clock, so a
# wavefront load that touches 64 lines occupies it for 64 clocks: 256 CUs x 2.4 GHz
GATHER_PEAK_GRECS = 256 * 2.4
F64_VALU_SUSTAINED_FRAC = 0.82
# ALGORITHMIC float64 operations per dyad term (one distance + one exp(-d) + product
# bookkeeping at ONE position): squared distance 4, root to the last bit 7, exp to 1 ulp 17,
# product update 4, edge sum 2 = 34 (the log-likelihood pass shares the distance and the
# exponential between its two candidates: 17 per candidate).  FIXED since round 1 so that
# `roofline.frac` compares between rounds; what the kernels spend today is reported beside it
# (the table exponential of round 2 brought the neighbour loop to 29, the pass to 15).
OPS_PER_TERM_SWEEP = 34
OPS_PER_TERM_LOGLIK = 17


_INSTR_CACHE = []


def kernel_instr_per_term():
    """vector instructions the built kernels spend per dyad term, counted in the library's own
    code object (profiles/instr_counts.py: disassembly of the hot loops); None where the
    disassembler is not at hand.  Disassembling takes a second or two of host time: run_rank calls
    this once BEFORE any device work, so that the profile phase, the warm-up and the timed steps
    follow each other without an idle device in between (a short timed window behind an idle
    device reads 5-9 % low)."""
    if _INSTR_CACHE:
        return _INSTR_CACHE[0]
    _INSTR_CACHE.append(_kernel_instr_per_term())
    return _INSTR_CACHE[0]


def _kernel_instr_per_term():
    try:
        sys.path.insert(0, os.path.join(ROOT, 'profiles'))
        import instr_counts
        c = instr_counts.counts()
        return (c.get('k_pipe_step<2,0,1>', {}).get('valu_per_term'),
                c.get('k_loglik_undirected<2,2>', {}).get('valu_per_candidate_term'),
                c.get('k_pipe_step<2,0,1>', {}).get('issue_slots_per_term'),
                c.get('k_loglik_undirected<2,2>', {}).get('issue_slots_per_candidate_term'))
    except Exception:       # noqa: BLE001
        return None, None, None, None


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--model', default='all', choices=['lsm', 'hdp', 'cc', 'ccu', 'all'],
                    help="cc: config 4 on a network drawn FROM the model (the quoted figure); ccu: on the "
                         "degree-regular network of rounds 1-5 (every node draws 20 out-neighbours uniformly: "
                         "the labelled easy case); all: lsm, hdp, cc, ccu, monks and lsm at 4 chains per GPU")
    ap.add_argument('--T', type=int, default=10)
    ap.add_argument('--N', type=int, default=2000)
    ap.add_argument('--D', type=int, default=2)
    ap.add_argument('--K', type=int, default=20, help='n_components of the HDP-LPCM')
    ap.add_argument('--density', type=float, default=0.03)
    ap.add_argument('--cc-T', type=int, default=5, help='--model cc: time steps')
    ap.add_argument('--cc-N', type=int, default=10000, help='--model cc: nodes')
    ap.add_argument('--cc-controls', type=int, default=100, help='--model cc: n_control')
    ap.add_argument('--chains-per-gpu', type=int, default=1,
                    help='independent chains per GPU, each on its own handle and stream, '
                         'enqueued by its own host thread (aggregate throughput)')
    ap.add_argument('--algo', type=int, default=0, help='sweep algorithm (0 auto)')
    ap.add_argument('--profile-steps', type=int, default=100,
                    help='iterations of the per-kernel event phase (averages over 100 x 18 sweep launches); it runs '
                         'BEHIND the timed window since round 5: a 20-step window behind it read 2-6 %% low')
    ap.add_argument('--settle-steps', type=int, default=300,
                    help='plain untimed iterations of every chain in front of the W warm-up steps (reported in the '
                         'line as untimed_steps_before_warmup): the chain starts on a device that sat idle through '
                         'the host-side generation of the network, and a window behind 50 ms of idle device reads '
                         '9 %% low, the next one still 5 %% (profiles/window_probe.py).  A call of the loop itself '
                         'costs 20 / 140 / 130 us beyond its iterations (LSM / HDP-LPCM / case-control: '
                         'profiles/per_call_cost.py).  0 switches them off')
    ap.add_argument('--windows', type=int, default=1,
                    help='diagnostic: time this many further windows of K steps behind the contract\'s one '
                         '(reported as later_windows; `value` is always the first window)')
    ap.add_argument('--cpu-iters', type=int, default=8,
                    help='oracle iterations timed for cpu_baseline (0 = skip)')
    ap.add_argument('--cpu-procs', type=int, default=1,
                    help='also time this many independent CPU chains in as many processes')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--backend', default=None,
                    help='collective backend (default nccl = RCCL; gloo for dry runs)')
    ap.add_argument('--force-collectives', action='store_true',
                    help='with --gpus 1: initialise the backend anyway and route the broadcast / '
                         'gather / barrier through it (exercises RCCL on a single-GPU box)')
    ap.add_argument('--share-device0', action='store_true',
                    help='dry run: every rank drives cuda:0 (with --backend gloo)')
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------
# self-launch: N ranks as child processes (the parent never touches the GPU)
# ------------------------------------------------------------------------------------
def launch_ranks(args):
    """N ranks as child processes through the package's own launcher (the one
    ``dynetlsm_amd.multichain.fit_chains`` uses): rank 0's JSON line is the output"""
    from dynetlsm_amd.multichain import launch_ranks as launch
    return launch([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus)


# ------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------
def in_threads(calls):
    """run the zero-argument callables side by side, one host thread each (the engine's
    enqueue calls are C calls that release the GIL); re-raises the first failure"""
    if len(calls) == 1:
        calls[0]()
        return
    import threading
    errs = []

    def wrap(f):
        try:
            f()
        except BaseException as e:       # noqa: B902
            errs.append(e)
    ths = [threading.Thread(target=wrap, args=(f,)) for f in calls]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errs:
        raise errs[0]


def share_network(src, others):
    """the packed network of chain `src` into the other chains of this GPU (device to device)"""
    if not others:
        return
    import torch
    n = src.network_packed_words()
    buf = torch.empty(n, dtype=torch.int32, device='cuda')
    src.get_network_packed(buf.data_ptr(), n)
    for c in others:
        c.set_network_packed(buf.data_ptr(), n)

class LsmWorkload(object):
    """configs[1]: DynamicNetworkLSM, undirected, device-resident loop (dlsm_lsm_run)"""
    name = 'lsm'

    def __init__(self, args, group, local_rank):
        from dynetlsm_amd import Chain, SamplerGrid
        from dynetlsm_amd.synthetic import synthetic_lsm_network
        self.args, self.group = args, group
        T, N, D = args.T, args.N, args.D
        rank = group.rank
        K, W, P = args.steps, args.warmup, args.profile_steps
        C = args.chains_per_gpu
        self.chains = [Chain(T, N, D, 'undirected', seed=20240229, chain_id=rank * C + c,
                             device=local_rank) for c in range(C)]
        self.chain = chain = self.chains[0]
        # the network: built, uploaded and packed on rank 0; the packed words are broadcast
        net = synthetic_lsm_network(T, N, D, density=args.density, seed=0) if rank == 0 else None
        self.net = net
        if rank == 0:
            chain.upload_network(net['Y'])
        group.broadcast_chain_network(chain)
        share_network(chain, self.chains[1:])
        self.X_init = group.broadcast_array(net['X_init'] if rank == 0 else np.zeros((T, N, D)))
        self.b_init = float(group.broadcast_array(
            np.array([net['intercept'], net['Y'].mean()]) if rank == 0 else np.zeros(2))[0])
        self.density = float(net['Y'].mean()) if rank == 0 else None
        for ch in self.chains:
            ch.set_positions(self.X_init)
            ch.set_intercepts([self.b_init])
            ch.set_prior_random_walk(2.0, 0.1)
            ch.set_samplers(SamplerGrid(T, N, step_size=0.1, tune=None))
            ch.lsm_configure([self.b_init], 2.0, step_size_intercept=0.1, tune=None,
                             n_iter_procrustes=0, sweep_algo=args.algo)
            ch.trace_alloc(1 + W + K * max(args.windows, 1) + P + args.settle_steps, logp0=0.0)
        self.next_it = 1

    def run(self, count):
        it = self.next_it
        in_threads([(lambda ch=ch: ch.lsm_run(it, count, procrustes_ref=0)) for ch in self.chains])
        self.next_it += count

    def synchronize(self):
        for ch in self.chains:
            ch.synchronize()

    def workload(self):
        a = self.args
        return ('DynamicNetworkLSM synthetic undirected T=%d N=%d d=%d, %d chain%s per GPU'
                % (a.T, a.N, a.D, a.chains_per_gpu, '' if a.chains_per_gpu == 1 else 's'))

    def metric(self):
        a = self.args
        return 'Gibbs iterations/sec (and ms/log-lik eval), T=%d N=%d d=%d' % (a.T, a.N, a.D)

    def profile(self):
        """per-kernel timing by HIP events on the chain's stream -> (roofline, extras)"""
        from dynetlsm_amd import _lib
        a, chain = self.args, self.chain
        T, N, D, P = a.T, a.N, a.D, a.profile_steps
        chain.profile_enable(True)
        chain.lsm_run(self.next_it, P, procrustes_ref=0)      # the first chain alone
        self.next_it += P
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
        return sweep_rooflines(chain, a, P, sweep_ms, ll_ms, ms_ev, n_ev, ms_rs, n_rs,
                               ms_ps / max(n_ps, 1), ms_fi / max(n_fi, 1))

    def results(self, first, count):
        """per-chain results for the final gather: posterior-mean positions, traces"""
        out = []
        for ch in self.chains:
            Xs, ics, lps = ch.trace_read(first, count, positions=True)
            out.append(dict(X_mean=Xs.mean(axis=0), logps=lps, intercepts=ics))
        return {k: np.stack([o[k] for o in out]) for k in out[0]}

    def acceptance(self):
        from dynetlsm_amd import SamplerGrid
        g = self.chain.get_samplers(SamplerGrid(self.args.T, self.args.N, 0.1, tune=None))
        return float(g.n_accepted.sum()) / max(float(g.n_steps.sum()), 1.0)

    def cpu_baseline(self):
        """rank 0: the scalar C oracle timed on this host's cores on a bounded sample of the
        same workload; the same leg checks the engine's log-likelihood at the chain's final
        state against the oracle"""
        from oracle import oracle as orc
        a, chain, net = self.args, self.chain, self.net
        T, N, D = a.T, a.N, a.D
        Y = net['Y']
        Xf = chain.get_positions()
        bf = chain.get_intercepts()[0]
        g = chain.loglik_full([[bf]])[0]
        o = orc.dynamic_network_loglikelihood_undirected(Y, Xf, bf)
        og = orc.SamplerGrid(T, N, 0.1, tune=None)
        st = orc.ChainState(self.X_init, og, Y=Y, intercept=[self.b_init], tau_sq=2.0,
                            sigma_sq=0.1, seed=20240229, chain=0)
        isamp = orc.ScalarSampler(0.1, 0, 0, 100, -1, 100)
        tc = time.perf_counter()
        for it in range(1, a.cpu_iters + 1):
            st.c.iter = it
            orc.lsm_iteration_undirected(st, isamp, self.b_init, 2.0)
        tc = time.perf_counter() - tc
        cpu = {'value': round(a.cpu_iters / tc, 5), 'unit': 'Gibbs iterations/s',
               'cores': 1, 'kind': 'port',
               'sample': '%d iterations of the same T=%d N=%d d=%d workload by the '
                         'scalar C oracle (sweep + 2 full log-lik evals per iteration), '
                         '%.1f s' % (a.cpu_iters, T, N, D, tc),
               'engine_loglik_rel_err_vs_oracle': abs(g - o) / abs(o)}
        if a.cpu_procs > 1:
            cpu['multi_process'] = cpu_chains_in_processes(a, a.cpu_procs)
        return cpu

    def close(self):
        for ch in self.chains:
            ch.close()


def cpu_chains_in_processes(a, procs):
    """`procs` independent CPU chains (the C oracle), one process per host core:
    the comparator of the N-GPU rows.  Each child builds the same network and times the same
    iterations; the aggregate is chains x iterations / the slowest child's time."""
    code = ('import sys, time, json; sys.path.insert(0, %r)\n'
            'import numpy as np\n'
            'from oracle import oracle as orc\n'
            'from dynetlsm_amd.synthetic import synthetic_lsm_network\n'
            'T, N, D, n, c = %d, %d, %d, %d, int(sys.argv[1])\n'
            'net = synthetic_lsm_network(T, N, D, density=%r, seed=0)\n'
            'st = orc.ChainState(net["X_init"], orc.SamplerGrid(T, N, 0.1, tune=None), Y=net["Y"],\n'
            '                    intercept=[net["intercept"]], tau_sq=2.0, sigma_sq=0.1,\n'
            '                    seed=20240229, chain=c)\n'
            'isamp = orc.ScalarSampler(0.1, 0, 0, 100, -1, 100)\n'
            't0 = time.perf_counter()\n'
            'for it in range(1, n + 1):\n'
            '    st.c.iter = it\n'
            '    orc.lsm_iteration_undirected(st, isamp, net["intercept"], 2.0)\n'
            'print(json.dumps(time.perf_counter() - t0))\n'
            % (ROOT, a.T, a.N, a.D, a.cpu_iters, a.density))
    ps = [subprocess.Popen([sys.executable, '-c', code, str(c)], stdout=subprocess.PIPE)
          for c in range(procs)]
    secs = [float(json.loads(p.communicate()[0].decode().strip().splitlines()[-1])) for p in ps]
    return {'value': round(procs * a.cpu_iters / max(secs), 5), 'unit': 'Gibbs iterations/s',
            'cores': procs, 'kind': 'port',
            'sample': '%d chains x %d iterations, one process each, slowest %.1f s; the host '
                      'has %d cores' % (procs, a.cpu_iters, max(secs), os.cpu_count())}


_DUR_CACHE = []


def stored_kernel_us(kname, model):
    """the rocprofv3 average of a kernel from profiles/kernel_durations.json (None if absent):
    the longest-running instantiation whose name starts with `kname` in that model's profile"""
    if not _DUR_CACHE:
        try:
            _DUR_CACHE.append(json.load(open(os.path.join(ROOT, 'profiles', 'kernel_durations.json'))))
        except Exception:       # noqa: BLE001
            _DUR_CACHE.append({})
    rows = _DUR_CACHE[0].get(model, {})
    best = None
    for name, r in rows.items():
        if name.split('<')[0] == kname.split('<')[0] and (best is None or r['calls'] > best[1]['calls']):
            best = (name, r)
    if best is None:
        return None
    return {'avg_us': best[1]['avg_us'], 'calls': best[1]['calls'],
            'source': 'rocprofv3 --kernel-trace --stats average of %s over %d calls (profiles/'
                      'kernel_durations.json <- %s)' % (best[0], best[1]['calls'],
                                                        _DUR_CACHE[0].get('_source', '').split(' (')[0])}


def sweep_rooflines(chain, a, P, sweep_ms, ll_ms, ms_ev, n_ev, ms_rs, n_rs, post_ms, fin_ms,
                    iteration_ms=None):
    """roofline blocks of the sweep's dominant kernel and of the log-likelihood pass.

    With the network bit-packed neither kernel is HBM bound (measured traffic is 0.5x / 0.16x
    the algorithmic float64 bytes): both are bound by float64 vector issue (distance + root,
    exp).  ``roofline`` therefore prices the dominant kernel against the float64 VALU peak;
    ``roofline_hbm`` keeps the algorithmic-bytes figure the metric asks for, next to the
    fraction of the HBM peak that the MEASURED traffic (rocprofv3 PMC, profiles/traffic.json -
    a stored number of the same kernel, not collected in this run) amounts to."""
    T, N, D = a.T, a.N, a.D
    sweep_bytes = 8.0 * T * N * N + 8.0 * T * N * D      # row j of Y[t] per MH step
    ll_bytes = 8.0 * T * N * (N - 1) / 2 + 8.0 * T * N * D
    sweep_terms = 2.0 * T * N * (N - 1)                  # (proposal, current) x ordered pairs
    ll_terms = 2.0 * T * N * (N - 1) / 2                 # 2 candidates x unordered pairs
    algo_used = chain.resolve_sweep_algo(a.algo)
    if n_ev > 0:
        launches = n_ev / float(P)
        kname = 'k_pipe_step' if algo_used == 4 else 'k_spec_eval'
        k_ms = ms_ev / n_ev        # start/stop events attached to the launch itself
    else:
        launches, kname, k_ms = 2.0, 'k_sweep_slice', sweep_ms / 2.0
    k_bytes, k_terms = sweep_bytes / launches, sweep_terms / launches
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
    ll_traffic = None
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get(kname, {}).get('hbm_bytes_per_launch')
            ll_traffic = tj.get('k_loglik_undirected', {}).get('hbm_bytes_per_launch')
        except Exception:
            traffic = None

    def valu(terms, instr, ms):
        tf = terms * instr * 2.0 / (ms * 1e-3) / 1e12
        return round(tf, 3), round(tf / F64_VALU_PEAK_TFLOPS, 4)

    # The duration `frac` divides by is the rocprofv3 average of this kernel stored under
    # profiles/ (kernel_durations.json <- rNN_kernel_stats_*.csv, the same command under the
    # profiler), so that the fraction follows from the committed files; the duration THIS run
    # measured - begin / end timestamps of the dispatches, events attached to the launches -
    # rides beside it (`us_per_launch_events`, `frac_events`; it reads ~7 % shorter: the
    # profiler serialises the dispatches and keeps their ramps from overlapping).
    stored = stored_kernel_us(kname, getattr(a, 'model_tag', 'lsm'))
    prof_ms = stored['avg_us'] * 1e-3 if stored else k_ms
    ach, frac = valu(k_terms, OPS_PER_TERM_SWEEP, prof_ms)
    ach_ev, frac_ev = valu(k_terms, OPS_PER_TERM_SWEEP, k_ms)
    ipt_sweep, ipt_ll, slots_sweep, slots_ll = kernel_instr_per_term()
    frac_x = valu(k_terms, ipt_sweep, prof_ms)[1] if ipt_sweep else None
    frac_slots = valu(k_terms, slots_sweep, prof_ms)[1] if slots_sweep else None
    roofline = {
        'bound': 'fp64_valu', 'kernel': kname, 'achieved': ach, 'peak': F64_VALU_PEAK_TFLOPS,
        'unit': 'TFLOP/s', 'frac': frac, 'traffic': traffic,
        'us_per_launch': round(1e3 * prof_ms, 3),
        'us_per_launch_source': (stored['source'] if stored else
                                 'events attached to the launches of this run (no stored profile)'),
        'us_per_launch_events': round(1e3 * k_ms, 3), 'achieved_events': ach_ev, 'frac_events': frac_ev,
        'launches_per_sweep': launches,
        'dyad_terms_per_launch': round(k_terms, 0),
        'f64_ops_per_term': OPS_PER_TERM_SWEEP,
        # the same duration priced by the vector instructions the kernel EXECUTES per term
        # (its code object, profiles/instr_counts.py): issue utilisation of the neighbour loop
        'valu_instr_per_term_in_kernel': ipt_sweep,
        'frac_executed': frac_x,
        # ... and with every instruction weighted by its measured issue cost (a v_rsq_f64 occupies the
        # pipe for 16.1 cycles against 4.15 for an fma: profiles/r04_valu_rates.txt): the fraction of the
        # SIMDs' issue slots the neighbour loop's instructions fill over the whole launch
        'issue_slots_per_term_in_kernel': slots_sweep,
        'frac_issue_slots': frac_slots,
        'instr_source': 'profiles/instr_counts.py on the built library (llvm-objdump of the hot loop)',
        'peak_note': 'nominal float64 vector peak (fma = 2 flop).  `frac` prices the ALGORITHMIC '
                     'operations of a dyad term as fma slots - 34, fixed since round 1 so that '
                     'rounds compare; the kernel itself executes %s vector instructions per term '
                     'since the table exponential, i.e. 34 > executed: `frac` overstates issue '
                     'utilisation by that ratio and `frac_executed` is the utilisation figure; a '
                     'pure fma stream sustains %.2f of the peak at 4 wavefronts per SIMD'
                     % (ipt_sweep, F64_VALU_SUSTAINED_FRAC),
        'frac_of_sustained': round(frac / F64_VALU_SUSTAINED_FRAC, 4),
        'dyad_terms_per_s_sweep': round(sweep_terms / (sweep_ms * 1e-3), 0)}
    ach_b = k_bytes / (prof_ms * 1e-3) / 1e9
    roofline_hbm = {
        'bound': 'hbm', 'kernel': kname, 'achieved': round(ach_b, 2), 'peak': HBM_PEAK_GBS,
        'unit': 'GB/s', 'frac': round(ach_b / HBM_PEAK_GBS, 5), 'traffic': traffic,
        'algorithmic_bytes_per_launch': round(k_bytes, 1),
        'measured_traffic_GBps': (round(traffic / (prof_ms * 1e-3) / 1e9, 2) if traffic else None),
        'measured_traffic_frac': (round(traffic / (prof_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                                  if traffic else None),
        'traffic_source': 'profiles/traffic.json (rocprofv3 PMC of this kernel, stored)',
        'sweep_GBps_all_launches': round(sweep_bytes / (sweep_ms * 1e-3) / 1e9, 2)}
    stored_ll = stored_kernel_us('k_loglik_undirected', getattr(a, 'model_tag', 'lsm'))
    ll_prof_ms = stored_ll['avg_us'] * 1e-3 if stored_ll else ll_ms
    ach_l, frac_l = valu(ll_terms, OPS_PER_TERM_LOGLIK, ll_prof_ms)
    roofline_ll = {
        'bound': 'fp64_valu', 'kernel': 'k_loglik_undirected<2,2>', 'achieved': ach_l,
        'peak': F64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': frac_l, 'traffic': ll_traffic,
        'us_per_launch': round(1e3 * ll_prof_ms, 3),
        'us_per_launch_source': stored_ll['source'] if stored_ll else 'events of this run',
        'us_per_launch_events': round(1e3 * ll_ms, 3),
        'frac_events': valu(ll_terms, OPS_PER_TERM_LOGLIK, ll_ms)[1],
        'f64_ops_per_term': OPS_PER_TERM_LOGLIK,
        'valu_instr_per_candidate_term_in_kernel': ipt_ll,
        'frac_executed': (valu(ll_terms, ipt_ll, ll_prof_ms)[1] if ipt_ll else None),
        'frac_issue_slots': (valu(ll_terms, slots_ll, ll_prof_ms)[1] if slots_ll else None),
        'algorithmic_GBps': round(ll_bytes / (ll_ms * 1e-3) / 1e9, 1),
        'algorithmic_frac_of_hbm': round(ll_bytes / (ll_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        'frac_of_sustained': round(frac_l / F64_VALU_SUSTAINED_FRAC, 4)}
    extra = {'roofline_hbm': roofline_hbm, 'roofline_loglik': roofline_ll,
             'ms_per_loglik_eval': round(ll_ms, 4),
             'loglik_eval_GBps': round(ll_bytes / (ll_ms * 1e-3) / 1e9, 1),
             'ms_sweep': round(sweep_ms, 4), 'ms_post_sweep': round(post_ms, 4),
             'ms_finalize': round(fin_ms, 4)}
    if n_rs > 0:
        extra['us_resolve_per_launch'] = round(1e3 * ms_rs / n_rs, 3)
    return roofline, extra


def iteration_valu_fraction(ms_per_step, a):
    """the whole iteration against the float64 vector peak: its algorithmic dyad terms
    (sweep + fused evaluation) x their instruction counts / wall time per iteration"""
    T, N = a.T, a.N
    ops = (2.0 * T * N * (N - 1) * OPS_PER_TERM_SWEEP +
           2.0 * T * N * (N - 1) / 2 * OPS_PER_TERM_LOGLIK) * 2.0
    tf = ops / (ms_per_step * 1e-3) / 1e12
    return {'achieved_TFLOPs': round(tf, 3), 'frac': round(tf / F64_VALU_PEAK_TFLOPS, 4),
            'frac_of_sustained': round(tf / F64_VALU_PEAK_TFLOPS / F64_VALU_SUSTAINED_FRAC, 4)}


class HdpWorkload(object):
    """configs[2] / configs[4]: DynamicNetworkHDPLPCM through the estimator facade's
    prepare / run seam (starting values supplied: the initialisation pipeline is measured
    separately, profiles/init_timing.py)"""
    name = 'hdp'

    def __init__(self, args, group, local_rank):
        from dynetlsm_amd import DynamicNetworkHDPLPCM
        from dynetlsm_amd.synthetic import synthetic_hdp_network
        self.args, self.group = args, group
        T, N, D, Kc = args.T, args.N, args.D, args.K
        rank = group.rank
        K, W, P = args.steps, args.warmup, args.profile_steps
        C = args.chains_per_gpu
        net = synthetic_hdp_network(T, N, D, density=args.density, seed=0) if rank == 0 else None
        self.net = net
        # starting values: the truth plus noise (positions), true labels and centres
        nc = 6
        if rank == 0:
            init_small = np.concatenate([net['X_init'].ravel(), [net['intercept']],
                                         net['mu_true'].ravel(), net['sigma_true'].ravel(),
                                         net['z_true'].ravel().astype(np.float64),
                                         [net['Y'].mean()]])
        else:
            init_small = np.zeros(T * N * D + 1 + nc * D + nc + T * N + 1)
        init_small = group.broadcast_array(init_small)
        o = 0
        X0 = init_small[o:o + T * N * D].reshape(T, N, D); o += T * N * D
        b0 = float(init_small[o]); o += 1
        mu_t = init_small[o:o + nc * D].reshape(nc, D); o += nc * D
        sg_t = init_small[o:o + nc]; o += nc
        z0 = init_small[o:o + T * N].reshape(T, N).astype(np.int64); o += T * N
        self.density = float(init_small[o])
        mu0 = np.zeros((Kc, D)); mu0[:nc] = mu_t
        rs = np.random.RandomState(5)
        mu0[nc:] = 3.0 * rs.randn(Kc - nc, D)
        sg0 = np.full(Kc, float(sg_t.mean())); sg0[:nc] = sg_t
        # the facade wants a network argument: rank 0 has the real one; the other ranks (and
        # the other chains of this GPU) get the packed words and only need the shape
        # (a zero-stride view: the estimator checks shape and NaNs, the 320 MB are never allocated)
        Yarg = net['Y'] if rank == 0 else np.broadcast_to(np.zeros(()), (T, N, N))
        self.start = dict(X=X0, b=b0, mu=mu0, sigma=sg0, z=z0)
        self.models = []
        for c in range(C):
            m = DynamicNetworkHDPLPCM(
                n_iter=1 + W + K * max(args.windows, 1) + P + args.settle_steps, tune=None, burn=None, n_components=Kc, n_features=D,
                random_state=1 + rank * C + c, device=local_rank, chain_id=rank * C + c,
                sweep_algo=args.algo, selection_type='map')
            first = self.models[0].chain_ if self.models else None

            def network_from(chain, first=first):
                if first is not None:
                    share_network(first, [chain])
                    return
                if rank == 0:
                    chain.upload_network(net['Y'])
                group.broadcast_chain_network(chain)
            m.copy = False
            m._prepare(Yarg, init=dict(X=X0, intercept=[b0], mu=mu0, sigma=sg0, z=z0),
                       network_from=network_from)
            if c == 0:              # the state the chain starts from (for the CPU leg)
                self.start.update(beta=m._st['beta'].copy(), weights=m._st['weights'].copy(),
                                  hyper=dict(m.hyper_.__dict__))
            self.models.append(m)
        self.model = self.models[0]
        self.next_it = 1

    def run(self, count):
        it = self.next_it
        in_threads([(lambda m=m: m._run(it, count)) for m in self.models])
        self.next_it += count

    def synchronize(self):
        for m in self.models:
            m.chain_.synchronize()

    def workload(self):
        a = self.args
        return ('DynamicNetworkHDPLPCM synthetic undirected T=%d N=%d d=%d K_max=%d '
                '(6 true clusters), %d chain%s per GPU'
                % (a.T, a.N, a.D, a.K, a.chains_per_gpu, '' if a.chains_per_gpu == 1 else 's'))

    def metric(self):
        a = self.args
        return 'Gibbs iterations/sec, HDP-LPCM T=%d N=%d d=%d K=%d' % (a.T, a.N, a.D, a.K)

    def profile(self):
        from dynetlsm_amd import _lib
        a, chain = self.args, self.model.chain_
        P = a.profile_steps
        chain.profile_enable(True)
        self.model._run(self.next_it, P)                      # the first chain alone
        self.next_it += P
        chain.synchronize()
        ms_sw, n_sw = chain.profile_read(_lib.K_SWEEP)
        ms_ll, n_ll = chain.profile_read(_lib.K_LOGLIK)
        ms_ps, n_ps = chain.profile_read(_lib.K_CENTER)
        ms_lb, n_lb = chain.profile_read(_lib.K_LABELS)
        ms_tl, n_tl = chain.profile_read(_lib.K_HDP_TAIL)
        ms_fi, n_fi = chain.profile_read(_lib.K_FINALIZE)
        chain.profile_enable(False)
        ms_ev, n_ev = chain.profile_read(_lib.K_SWEEP_EVAL)
        import copy
        a_h = copy.copy(a); a_h.model_tag = 'hdp'          # (the stored profile of THIS model)
        roofline, extra = sweep_rooflines(chain, a_h, P, ms_sw / max(n_sw, 1), ms_ll / max(n_ll, 1),
                                          ms_ev, n_ev, 0.0, 0, ms_ps / max(n_ps, 1), 0.0)
        extra['ms_label_kernels_per_iteration'] = round(ms_lb / max(P, 1), 4)
        extra['ms_hdp_draws_and_logp_per_iteration'] = round(ms_tl / max(P, 1), 4)
        extra['ms_finalize'] = round(ms_fi / max(n_fi, 1), 4)
        # what follows the centring pass: the label block update and the HDP's conjugate draws (four
        # launches; the label counts are a role of the first).  Neither bandwidth nor arithmetic bounds
        # it: 5 dependent launches of 1 .. 760 workgroups, each a chain of dependent draws (DESIGN.md,
        # phase stamps in profiles/r03_labels_notes.md and profiles/hdp_tail_timing.py); in the timed
        # run the intercept's likelihood pass runs BESIDE them on the chain's second queue
        # (`hdp_queues`; profiles/r04_hdp_two_queues.md) - the per-launch figures here are from the
        # profiled steps, which keep one queue (their events serialise the launches).  `achieved` is
        # the label update's algorithmic rate (backward messages 2 (T-1) N K^2 flop + forward
        # draws 2 T N K) against the f64 matrix peak, for the record: 1-2 % of it.
        T, N, K = a.T, a.N, a.K
        lab_flop = 2.0 * (T - 1) * N * K * K + 2.0 * T * N * K
        lab_s = max(ms_lb / max(P, 1), 1e-9) * 1e-3
        tail_us = 1e3 * (ms_lb + ms_tl) / max(P, 1)
        extra['roofline_tail'] = {
            'bound': 'latency', 'kernels': ['k_sample_labels_mfma', 'k_hdp_stage1(+label counts)',
                                            'k_hdp_stage2', 'k_hdp_stage3', 'k_hdp_hypers(+proposal pass)'],
            'launches': 5, 'us_per_iteration': round(tail_us, 2),
            'us_label_update': round(1e3 * ms_lb / max(P, 1), 2),
            'us_conjugate_draws': round(1e3 * ms_tl / max(P, 1), 2),
            'us_floor_launch_boundaries': round(5 * 1.7, 1),
            'achieved': round(lab_flop / lab_s / 1e12, 4), 'peak': FP64_MATRIX_PEAK_TFLOPS,
            'unit': 'TFLOP/s', 'frac': round(lab_flop / lab_s / 1e12 / FP64_MATRIX_PEAK_TFLOPS, 5),
            'note': 'label update only in achieved / peak; the draws are scalar chains (gamma, beta, '
                    'truncated normal variates) whose launches are bounded by their longest chain'}
        return roofline, extra

    def results(self, first, count):
        out = []
        for m in self.models:       # summaries from the trace where it lies (no 100 MB pull)
            ch = m.chain_
            tr = ch.hdp_trace_read(first, count, positions=False, labels=False, weights=False)
            nk = ch.post_trace_label_counts(first, count)
            out.append(dict(X_mean=ch.post_trace_mean(first, count), logps=tr['logps'],
                            intercepts=tr['intercepts'], lambdas=tr['lambdas'],
                            n_clusters_used=(nk > 0).any(axis=1).sum(axis=1).astype(np.float64)))
        return {k: np.stack([o[k] for o in out]) for k in out[0]}

    def acceptance(self):
        m = self.model
        g = m.chain_.get_samplers(m.latent_samplers)
        return float(g.n_accepted.sum()) / max(float(g.n_steps.sum()), 1.0)

    def cpu_baseline(self):
        """rank 0: the oracle's restatement of the same iteration (C sweep, log-likelihoods and
        label update; the O(T K^2) draws in numpy) from the same starting state, 1 core"""
        from oracle import oracle as orc
        from oracle import hdp_loop_oracle as hlo
        a, st, m = self.args, self.start, self.model
        T, N = a.T, a.N
        hy = {k: v for k, v in st['hyper'].items() if k != 'n_components'}
        n_it = max(2, a.cpu_iters)
        oc = hlo.HdpChain(self.net['Y'], st['X'].copy(), [st['b']], st['mu'].copy(), st['sigma'].copy(),
                          st['z'].copy(), st['beta'].copy(), st['weights'].copy(), m.lambda_prior,
                          hlo.Hyper(**hy), orc.SamplerGrid(T, N, 0.1, tune=None), st['b'],
                          m.intercept_variance_prior, orc.ScalarMetropolis(0.1, None, 100),
                          seed=m.chain_.seed, chain=m.chain_.chain_id)
        tc = time.perf_counter()
        lps, snaps = [], []
        for it in range(1, n_it + 1):
            lps.append(oc.iteration(it))
            snaps.append((oc.X.copy(), oc.z.copy(), float(oc.intercept[0]), float(oc.lmbda[0])))
        tc = time.perf_counter() - tc
        # the engine ran these very iterations (same start, same Philox key) at the head of its
        # trace: parity evidence at the full size, inside the driver's run
        tr = m.chain_.hdp_trace_read(1, n_it, weights=False)
        err = {'iterations': n_it,
               'labels_equal': bool(all(np.array_equal(tr['zs'][i], snaps[i][1]) for i in range(n_it))),
               'positions_max_abs': float(max(np.abs(tr['Xs'][i] - snaps[i][0]).max() for i in range(n_it))),
               'intercepts_max_abs': float(max(abs(tr['intercepts'][i, 0] - snaps[i][2]) for i in range(n_it))),
               'lambda_max_abs': float(max(abs(tr['lambdas'][i, 0] - snaps[i][3]) for i in range(n_it))),
               'logps_max_rel': float(max(abs(tr['logps'][i] - lps[i]) / abs(lps[i]) for i in range(n_it)))}
        return {'value': round(n_it / tc, 5), 'unit': 'Gibbs iterations/s', 'cores': 1, 'kind': 'port',
                'engine_trace_max_err_vs_oracle': err,
                'sample': '%d iterations of the same T=%d N=%d K=%d workload by the oracle (scalar C '
                          'sweep, two log-likelihood evaluations and label update; numpy draws), '
                          '%.1f s' % (n_it, T, N, a.K, tc)}

    def close(self):
        for m in self.models:
            m.chain_.close()


class CcWorkload(object):
    """configs[3]: directed case-control DynamicNetworkLSM, T=5 N=10 000 d=2, n_control=100,
    device-resident loop (dlsm_lsm_run: sweep, centring, two intercept steps and the radii
    step around case-control log-likelihood passes); controls redrawn on the device every
    n_resample_control = 100 iterations (case_control_likelihood.py:27-33).  The network is
    given as edge tables (the dense tensor would be 4 GB)."""
    name = 'cc'
    network = 'model'

    def __init__(self, args, group, local_rank):
        from dynetlsm_amd import Chain, SamplerGrid
        from dynetlsm_amd.synthetic import synthetic_sparse_directed, synthetic_directed_from_model
        self.args, self.group = args, group
        T, N, C = args.cc_T, args.cc_N, args.cc_controls
        self.T, self.N, self.C = T, N, C
        rank = group.rank
        K, W, P = args.steps, args.warmup, args.profile_steps
        # starting state and samplers: the degree-regular network as in rounds 1-5; the model-drawn one as in
        # tests/test_gpu_c4_full_size.py (generating intercepts, priors and step sizes at the scale of the cloud)
        self.b0, self.tau_sq, self.sigma_sq, self.step_x, self.step_b = [1.0, 0.5], 1e-4, 1e-5, 0.002, 0.1
        if rank == 0:
            if self.network == 'model':
                net = synthetic_directed_from_model(T, N, 20.0, seed=0)
                X, radii, degree, in_edges, out_edges = (net['X'], net['radii'], net['degree'], net['in_edges'],
                                                          net['out_edges'])
                w = float(net['width'])
                par = np.array([net['intercepts'][0], net['intercepts'][1], w * w, (0.1 * w) ** 2, 0.02 * w, 0.01])
            else:
                X, radii, degree, in_edges, out_edges = synthetic_sparse_directed(T, N, 20, 0)
                par = np.array(self.b0 + [self.tau_sq, self.sigma_sq, self.step_x, self.step_b])
            shp = np.array([in_edges.shape[2], out_edges.shape[2]] + list(par), dtype=np.float64)
        else:
            shp = np.zeros(8)
        shp = group.broadcast_array(shp)
        Din, Dout = int(shp[0]), int(shp[1])
        self.b0 = [float(shp[2]), float(shp[3])]
        self.tau_sq, self.sigma_sq, self.step_x, self.step_b = (float(v) for v in shp[4:8])
        if os.environ.get('DLSM_BENCH_CC_STEP_SCALE'):      # (experiments: the positions' step size, hence the acceptance rate)
            self.step_x *= float(os.environ['DLSM_BENCH_CC_STEP_SCALE'])
        if rank != 0:
            X, radii = np.zeros((T, N, 2)), np.zeros(N)
            degree = np.zeros((T, N, 2)); in_edges = np.zeros((T, N, Din))
            out_edges = np.zeros((T, N, Dout))
        X = group.broadcast_array(X); radii = group.broadcast_array(radii)
        degree = group.broadcast_array(degree).astype(np.int64)
        in_edges = group.broadcast_array(in_edges).astype(np.int64)
        out_edges = group.broadcast_array(out_edges).astype(np.int64)
        self.density = float(degree[:, :, 1].mean() / (N - 1))
        self.mean_terms = float(degree.sum(axis=2).mean() + 2 * C)
        self.mean_out = float(degree[:, :, 1].mean())
        # how uneven the nodes' term rows are (the uniform network: out-degree 20 +- 0, in-degree Poisson(20))
        tot = degree.sum(axis=2)
        self.degree_stats = {'out_max': int(degree[:, :, 1].max()), 'out_p99': float(np.percentile(degree[:, :, 1], 99)),
                             'in_max': int(degree[:, :, 0].max()), 'in_p99': float(np.percentile(degree[:, :, 0], 99)),
                             'terms_per_node_mean': float(tot.mean() + 2 * C),
                             'terms_per_node_max': int(tot.max() + 2 * C)}
        self.tables = (X, radii, degree, in_edges, out_edges) if rank == 0 else None
        self.chains = []
        for c in range(args.chains_per_gpu):
            ch = Chain(T, N, 2, 'case_control', seed=20240229, chain_id=rank * args.chains_per_gpu + c,
                       device=local_rank)
            ch.upload_edges(in_edges, out_edges, degree)
            ch.resample_controls(0, C)
            if rank == 0 and c == 0:        # the controls the first 99 iterations use (CPU leg)
                self.controls0 = ch.get_controls()
            ch.set_positions(X); ch.set_radii(radii); ch.set_intercepts(self.b0)
            ch.set_prior_random_walk(self.tau_sq, self.sigma_sq)
            ch.set_samplers(SamplerGrid(T, N, step_size=self.step_x, tune=None))
            ch.lsm_configure(self.b0, 2.0, step_size_intercept=self.step_b, tune=None,
                             n_iter_procrustes=0, sweep_algo=args.algo, step_size_radii=175000.,
                             radii_tune=None)
            ch.trace_alloc(1 + W + K * max(args.windows, 1) + P + args.settle_steps, logp0=0.0)
            self.chains.append(ch)
        self.chain = self.chains[0]
        self.next_it = 1
        self.n_resample = 100

    def _run_chain(self, ch, first, count):
        it, last = first, first + count - 1
        while it <= last:
            if it % self.n_resample == 0:
                ch.resample_controls(it, self.C)
            nxt = min(last + 1, (it // self.n_resample + 1) * self.n_resample)
            ch.lsm_run(it, nxt - it, procrustes_ref=0)
            it = nxt

    def run(self, count):
        it = self.next_it
        in_threads([(lambda ch=ch: self._run_chain(ch, it, count)) for ch in self.chains])
        self.next_it += count

    def synchronize(self):
        for ch in self.chains:
            ch.synchronize()

    def workload(self):
        return ('DynamicNetworkLSM directed case-control T=%d N=%d d=2, n_control=%d, network %s, mean degree '
                '%.1f, %d chain%s per GPU' % (self.T, self.N, self.C,
                                              'drawn from the model (synthetic_directed_from_model)'
                                              if self.network == 'model' else
                                              'degree-regular (20 uniform out-neighbours per node: the easy case)',
                                              self.density * (self.N - 1),
                                              len(self.chains), '' if len(self.chains) == 1 else 's'))

    def metric(self):
        return ('Gibbs iterations/sec, directed case-control T=%d N=%d d=2 n_control=%d, %s network'
                % (self.T, self.N, self.C, 'model-drawn' if self.network == 'model' else 'degree-regular'))

    def profile(self):
        from dynetlsm_amd import _lib
        a, chain = self.args, self.chain
        P = a.profile_steps
        chain.profile_enable(True)
        self._run_chain(chain, self.next_it, P)
        self.next_it += P
        chain.synchronize()
        ms_sw, n_sw = chain.profile_read(_lib.K_SWEEP)
        ms_ll, n_ll = chain.profile_read(_lib.K_LOGLIK)
        ms_ps, n_ps = chain.profile_read(_lib.K_CENTER)
        chain.profile_enable(False)
        ms_ev, n_ev = chain.profile_read(_lib.K_SWEEP_EVAL)
        T, N = self.T, self.N
        sweep_ms = ms_sw / max(n_sw, 1)
        launches = n_ev / float(max(P, 1))
        k_ms = ms_ev / max(n_ev, 1)
        algo = chain.resolve_sweep_algo(a.algo)
        kname = {5: 'k_ccpipe_step', 4: 'k_pipe_step<case-control>'}.get(algo, 'k_spec_eval_cc')
        # algorithmic bytes: every MH step evaluates the partial log-likelihood
        # at the proposal and at the current position; each evaluation walks the node's edge
        # and control lists (int64 index 8 B) and gathers X[e] (16 B at d = 2) and radii[e] (8 B)
        sweep_bytes = 2.0 * T * N * self.mean_terms * (8 + 16 + 8)
        k_bytes = sweep_bytes / max(launches, 1)
        k_ms_events = k_ms
        stored = stored_kernel_us(kname, self.name)     # the rocprofv3 average under profiles/, as in the headline
        if stored:
            k_ms = stored['avg_us'] * 1e-3
        ach = k_bytes / (k_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(kname, {}).get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        roofline = {'bound': 'hbm', 'kernel': kname, 'achieved': round(ach, 2), 'peak': HBM_PEAK_GBS,
                    'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 5), 'traffic': traffic,
                    'traffic_source': 'profiles/traffic.json (rocprofv3 PMC of this kernel, stored)',
                    'us_per_launch': round(1e3 * k_ms, 3),
                    'us_per_launch_source': stored['source'] if stored else 'events of this run',
                    'us_per_launch_events': round(1e3 * k_ms_events, 3),
                    'launches_per_sweep': launches,
                    'algorithmic_bytes_per_launch': round(k_bytes, 1),
                    'gathered_terms_per_s_sweep': round(2.0 * T * N * self.mean_terms /
                                                        (sweep_ms * 1e-3), 0),
                    'note': 'not an HBM kernel: the term rows and records (2.4 MB a slice) are L2 resident, HBM '
                            'traffic is far below the algorithmic bytes; its evaluator is bound by float64 vector '
                            'issue (roofline_valu), its resolver by the dependency chain of accepted moves'}
        # The bound that fits both kernels: a gathered term is one 32-byte record from a random
        # node of the slice, and a CU's vector L1 looks up one cache line per clock, whatever
        # the hit rate (the records are L2 resident): 256 CUs x 2.4 GHz records per second.
        rec_sweep = float(T) * N * self.mean_terms          # one record per term, both candidates share it
        roofline_gather = {
            'bound': 'gather rate (one cache-line look-up per CU per clock)', 'kernel': kname,
            'achieved': round(rec_sweep / (sweep_ms * 1e-3) / 1e9, 2), 'peak': GATHER_PEAK_GRECS,
            'unit': 'G records/s', 'frac': round(rec_sweep / (sweep_ms * 1e-3) / 1e9 / GATHER_PEAK_GRECS, 4),
            'records_per_sweep': rec_sweep}
        # What the evaluator of k_ccpipe_step is actually bound by (profiles/r05_ccpipe_notes.md): float64 vector
        # issue.  One item (node) is one wavefront's pass over its gathered terms; its vector instructions are
        # counted in the library's code object (profiles/instr_counts.py, quarter-rate float64 instructions at
        # four issue slots); a SIMD issues one slot per clock.
        roofline_valu = None
        if algo == 5:
            try:
                sys.path.insert(0, os.path.join(ROOT, 'profiles'))
                import instr_counts
                cc = instr_counts.counts().get('k_ccpipe_step<2>')
            except Exception:       # noqa: BLE001
                cc = None
            if cc:
                items = float(T) * N / max(launches, 1)
                slots = items * cc['issue_slots_per_item']
                peak_slots = 4 * 256 * 2.4e9                            # SIMDs x clocks a second
                roofline_valu = {
                    'bound': 'vector issue (one slot per SIMD per clock; float64 mul/fma/add take four)',
                    'kernel': kname, 'items_per_launch': round(items, 1),
                    'valu_per_item': cc['valu_per_item'], 'issue_slots_per_item': cc['issue_slots_per_item'],
                    'achieved': round(slots / (k_ms * 1e-3) / 1e12, 3), 'peak': round(peak_slots / 1e12, 3),
                    'unit': 'T issue slots/s', 'frac': round(slots / (k_ms * 1e-3) / peak_slots, 4),
                    'note': 'over the whole launch, the serial resolver included.  Far from issue-bound chip-wide: '
                            'with its items spread over every CU (two or three wavefronts per SIMD) the evaluator is a '
                            'latency chain per wavefront - row -> records -> dependent float64 arithmetic - and the '
                            'launch ends with the resolver\'s chain of accepted moves (profiles/r05_ccpipe_notes.md)'}
        ll_ms = ms_ll / max(n_ll, 1)
        rec_ll = float(T) * N * (float(self.mean_out) + self.C)     # out-edges + out-controls
        # (the pass since round 6: k_loglik_casecontrol_stream - streaming wavefronts, reciprocal radii in LDS, one
        # 16-byte position gathered per term.  Its two bounds, both from counters of this kernel under rocprofv3
        # (profiles/r06_cc_pass_notes.md): vector issue - SQ_INSTS_VALU per pass, four clocks each on 4 x n_cu
        # SIMDs - and the vector L1's line fills - one 128-byte line per gathered term at 64 bytes per CU and clock)
        pass_form = os.environ.get('DLSM_CC_PASS', 'stream')
        valu_per_pass = {'stream': 0.5 * (20.32e6 + 8.98e6), 'records': 0.5 * (18.81e6 + 8.40e6),
                         'rows': 0.5 * (39.68e6 + 16.59e6)}.get(pass_form)
        valu_source = 'profiles/r06_cc_pass_notes.md'
        try:        # this round's stored counters of the default form (collect_round.sh -> profiles/r06_cc_pass_counters.jsonl)
            rows = [json.loads(ln) for ln in open(os.path.join(ROOT, 'profiles', 'r06_cc_pass_counters.jsonl'))
                    if ln.startswith('{')]
            v = [r['SQ_INSTS_VALU'] for r in rows if 'loglik_casecontrol' in r['kernel'] and 'SQ_INSTS_VALU' in r]
            if pass_form == 'stream' and len(v) == 2:
                valu_per_pass, valu_source = 0.5 * sum(v), 'profiles/r06_cc_pass_counters.jsonl'
        except Exception:       # noqa: BLE001
            pass
        n_cu = 256
        roofline_ll = {
            'bound': 'vector issue and L1 line fills (neither alone: see note)',
            'kernel': 'k_loglik_casecontrol_%s' % ('rows' if pass_form == 'rows' else 'stream'),
            'achieved': round(rec_ll / (ll_ms * 1e-3) / 1e9, 2), 'peak': GATHER_PEAK_GRECS,
            'unit': 'G records/s', 'frac': round(rec_ll / (ll_ms * 1e-3) / 1e9 / GATHER_PEAK_GRECS, 4),
            'records_per_pass': rec_ll, 'us_per_pass': round(1e3 * ll_ms, 2),
            'algorithmic_bytes_per_pass': rec_ll * (4 + 32),
            'hbm_equivalent_GBs': round(rec_ll * 36 / (ll_ms * 1e-3) / 1e9, 1),
            'valu_instructions_per_pass_measured': valu_per_pass, 'valu_instructions_source': valu_source,
            'frac_vector_issue': (round(valu_per_pass * 4 / (4 * n_cu * 2.4e9) / (ll_ms * 1e-3), 4)
                                  if valu_per_pass else None),
            'frac_l1_line_fill': round(rec_ll * 128 / (n_cu * 64 * 2.4e9) / (ll_ms * 1e-3), 4),
            'note': 'average over the iteration\'s two passes (the four candidates of both intercept steps in '
                    'one, the radii\'s in the other).  Round 5\'s two-rows-per-wavefront kernel executed 39.7 M / 16.6 M '
                    'vector instructions per pass (81.5 / 42 us: it was bound by vector issue, not by gathers); the '
                    'streaming kernel executes 20.3 M / 9.0 M (running control product per run of equal out-degrees, '
                    'brackets shared between candidates, header by lane) and gathers one 16-byte position per term '
                    'instead of a 32-byte record in two requests: 54 / 37.5 us'}
        extra = {'roofline_gather': roofline_gather, 'roofline_valu': roofline_valu, 'roofline_loglik': roofline_ll,
                 'ms_sweep': round(sweep_ms, 4),
                 'ms_per_loglik_pass': round(ms_ll / max(n_ll, 1), 4),
                 'loglik_passes_per_iteration': round(n_ll / float(max(P, 1)), 2),
                 'ms_post_sweep': round(ms_ps / max(n_ps, 1), 4)}
        return roofline, extra

    def results(self, first, count):
        out = []
        for ch in self.chains:
            Xs, ics, lps = ch.trace_read(first, count, positions=True)
            out.append(dict(X_mean=Xs.mean(axis=0), logps=lps, intercepts=ics))
        return {k: np.stack([o[k] for o in out]) for k in out[0]}

    def acceptance(self):
        from dynetlsm_amd import SamplerGrid
        g = self.chain.get_samplers(SamplerGrid(self.T, self.N, self.step_x, tune=None))
        return float(g.n_accepted.sum()) / max(float(g.n_steps.sum()), 1.0)

    def cpu_baseline(self):
        """rank 0: the oracle's restatement of the same iteration (scalar C sweep and case-control
        log-likelihoods; the radii's Dirichlet proposal in numpy) with the chain's controls"""
        from oracle import oracle as orc
        X, radii, degree, in_edges, out_edges = self.tables
        T, N = self.T, self.N
        ci, co = self.controls0
        cc = dict(in_edges=in_edges, out_edges=out_edges, degree=degree, control_nodes_in=ci,
                  control_nodes_out=co)
        st = orc.ChainState(X, orc.SamplerGrid(T, N, self.step_x, tune=None), model=2, intercept=list(self.b0),
                            radii=radii.copy(), case_control=cc, tau_sq=self.tau_sq, sigma_sq=self.sigma_sq,
                            seed=20240229, chain=0)

        def loglik(Xc, b, r):
            return orc.approx_directed_network_loglikelihood(Xc, r, in_edges, out_edges, degree, co,
                                                             b[0], b[1])
        isamp = [orc.ScalarMetropolis(self.step_b, None, 100) for _ in range(2)]
        rsamp = orc.ScalarMetropolis(175000., None, 100)
        n_it = min(max(2, self.args.cpu_iters), self.n_resample - 1)
        X_ref = X.copy()            # the engine rotates every iteration onto its trace row 0 (lsm.py:495-498)
        tc = time.perf_counter()
        lps, snaps = [], []
        for it in range(1, n_it + 1):
            lps.append(orc.lsm_iteration_directed(st, it, loglik, isamp, rsamp, np.array(self.b0), 2.0,
                                                  X_ref=X_ref))
            snaps.append((st.X.copy(), st.intercept.copy(), st.radii.copy()))
        tc = time.perf_counter() - tc
        # the engine's trace rows 1 .. n_it: the same iterations from the same start, same key
        Xs, ics, elps = self.chain.trace_read(1, n_it, positions=True)
        rad = self.chain.trace_read_radii(1, n_it)
        err = {'iterations': n_it,
               'positions_max_abs': float(max(np.abs(Xs[i] - snaps[i][0]).max() for i in range(n_it))),
               'intercepts_max_abs': float(max(np.abs(ics[i] - snaps[i][1]).max() for i in range(n_it))),
               'radii_max_rel': float(max(np.abs(rad[i] / snaps[i][2] - 1).max() for i in range(n_it))),
               'logps_max_rel': float(max(abs(elps[i] - lps[i]) / abs(lps[i]) for i in range(n_it)))}
        return {'value': round(n_it / tc, 5), 'unit': 'Gibbs iterations/s', 'cores': 1, 'kind': 'port',
                'engine_trace_max_err_vs_oracle': err,
                'sample': '%d iterations of the same T=%d N=%d n_control=%d workload by the oracle '
                          '(scalar C sweep, six case-control log-likelihood evaluations; the radii '
                          'proposal in numpy), %.1f s' % (n_it, T, N, self.C, tc)}

    def close(self):
        for ch in self.chains:
            ch.close()


class CcUniformWorkload(CcWorkload):
    """config 4 on the degree-regular network rounds 1-5 timed (every node draws 20 out-neighbours uniformly: no
    parameter of the model generates it - equal-length term rows and lists, the easy case)"""
    name = 'ccu'
    network = 'uniform'


def measure(wl, args, group):
    """untimed settle steps, W untimed warm-up steps, then exactly K steps between barriers +
    synchronisation, max over ranks; the final gather and the roofline profile (P steps with events
    around every launch) follow outside the timed region"""
    import torch
    K, W = args.steps, args.warmup
    # (torch initialises its device state lazily: the first torch.cuda.synchronize() of the process here, not
    # between the warm-up and the timed window)
    torch.cuda.synchronize()
    group.barrier()
    # Untimed settle steps, the W warm-up steps, the timed window - and the per-kernel profile (P steps with HIP
    # events around every launch) BEHIND it.  Round 5 measured what the profile phase does to a window that follows
    # it, 300 settle steps and the warm-up notwithstanding (same box, alternating, device events of the window):
    # 4430-4596 it/s behind the profile phase, 4680-4709 without one - its ~2000 events keep the runtime busy on
    # the device's side for a while; every later window ran at 4650-4710 either way.
    roofline, extra = (None, {})
    if args.settle_steps > 0:
        wl.run(args.settle_steps)
    wl.run(W)
    wl.synchronize()
    torch.cuda.synchronize()
    group.barrier()
    probe = getattr(wl, 'chain', None) if args.windows > 1 and args.chains_per_gpu == 1 else None
    t0 = time.perf_counter()
    if probe is not None:
        probe.timer_start()         # (diagnostic runs only: events on the chain's stream around the window)
    wl.run(K)
    wl.host_enqueue_seconds = time.perf_counter() - t0     # the host's share: the calls return when all is enqueued
    wl.first_window_device_ms = probe.timer_stop() if probe is not None else None
    wl.synchronize()
    torch.cuda.synchronize()
    group.barrier()
    mine = time.perf_counter() - t0
    elapsed = group.max_over_ranks(mine)
    tb = time.perf_counter()
    group.barrier()                 # what the closing barrier of the timed region costs by itself
    wl.barrier_ms = 1e3 * (time.perf_counter() - tb)
    wl.per_rank_seconds = [float(v[0]) for v in group.gather_arrays(np.array([mine]))]
    wl.later_windows = []
    for _ in range(max(args.windows, 1) - 1):       # (diagnostic: the same window again, behind the first)
        group.barrier()
        tw = time.perf_counter()
        wl.run(K)
        wl.synchronize()
        torch.cuda.synchronize()
        group.barrier()
        wl.later_windows.append(group.max_over_ranks(time.perf_counter() - tw))
    acc = wl.acceptance()
    gathered = group.gather_results(wl.results(1 + args.settle_steps + W, K))
    if args.profile_steps > 0:
        roofline, extra = wl.profile()
    # (ranks, chains per GPU, ...) -> (chains, ...)
    gathered = {k: v.reshape((-1,) + v.shape[2:]) for k, v in gathered.items()}
    return elapsed, roofline, extra, acc, gathered


def chain_summaries(g):
    """[intercept mean, intercept sd, logp mean, logp last] per chain + the spread of the
    gathered posterior-mean positions between chains"""
    out = []
    for c in range(g['logps'].shape[0]):
        out.append([float(g['intercepts'][c, :, 0].mean()), float(g['intercepts'][c, :, 0].std()),
                    float(g['logps'][c].mean()), float(g['logps'][c, -1])])
    return out


def run_rank(args):
    import torch
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if rank == 0:
            print('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    from dynetlsm_amd.multichain import init_chain_group
    if args.share_device0:
        local_rank = 0
        os.environ['LOCAL_RANK'] = '0'
        # (processes sharing a GPU: the HDP-LPCM loop stays on one queue - multichain.launch_ranks)
        os.environ.setdefault('DLSM_HDP_QUEUES', '1')
        os.environ.setdefault('DLSM_CC_HELPERS', '0')       # (no in-launch waits across processes either)
        os.environ.setdefault('DLSM_PIPE_XSERVE', '0')
    # one process per GPU; collectives over RCCL (backend "nccl") unless told otherwise
    group = init_chain_group(backend=(args.backend or 'nccl') if (world > 1 or args.force_collectives)
                             else 'gloo', force=args.force_collectives)
    torch.cuda.set_device(local_rank)
    K, W = args.steps, args.warmup
    if args.profile_steps > 0:
        kernel_instr_per_term()         # host-only, cached: not between the profile phase and the timed steps
    models = ['lsm', 'hdp', 'cc', 'ccu'] if args.model == 'all' else [args.model]
    lines = []
    def run_model(name, args=args):
            K, W = args.steps, args.warmup          # (the extra configurations pass their own window)
            wl = {'lsm': LsmWorkload, 'hdp': HdpWorkload, 'cc': CcWorkload,
                  'ccu': CcUniformWorkload}[name](args, group, local_rank)
            elapsed, roofline, extra, acc, gathered = measure(wl, args, group)
            cpu = None
            if rank == 0 and world == 1 and not args.no_cpu and args.cpu_iters > 0:      # at N = 1 only
                cpu = wl.cpu_baseline()
            if rank == 0:
                C = args.chains_per_gpu
                value = world * C * K / elapsed
                xm = gathered['X_mean']
                line = {
                    'metric': wl.metric(), 'value': round(value, 3), 'unit': 'Gibbs iterations/s',
                    'n_gpus': world, 'steps': K, 'warmup': W,
                    'untimed_steps_before_warmup': {'settle_steps': args.settle_steps},
                    'profile_steps_behind_the_timed_window': max(args.profile_steps, 0),
                    'ms_per_step': round(1e3 * elapsed / K, 4), 'higher_is_better': True,
                    'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
                    'config': {'workload': wl.workload(), 'density': round(wl.density, 4),
                               'chains': world * C, 'chains_per_gpu': C,
                               'sweep_algo': (wl.model.chain_ if name == 'hdp' else wl.chain)
                               .resolve_sweep_algo(args.algo),
                               'mh_acceptance_rate': round(acc, 3),
                               'network_broadcast': ('packed words, device to device (%s)'
                                                     % group.backend)
                               if (world > 1 or args.force_collectives) else 'none'},
                    # every rank's own rate between the barriers (a straggler GPU shows here; `value`
                    # uses the slowest)
                    'per_rank_value': [round(C * K / sec, 3) for sec in wl.per_rank_seconds],
                    'barrier_ms': round(wl.barrier_ms, 4),
                    # how long the host took to enqueue the timed steps (asynchronous launches): well below
                    # ms_per_step = the device is the limit, not the launching thread
                    'host_enqueue_ms_per_step': round(1e3 * wl.host_enqueue_seconds / K, 4),
                    'later_windows_it_per_s': [round(world * C * K / sec, 1) for sec in wl.later_windows],
                    'first_window_device_it_per_s': (round(K / (wl.first_window_device_ms * 1e-3), 1)
                                                     if wl.first_window_device_ms else None),
                    'roofline': roofline, 'cpu_baseline': cpu,
                    'chain_summaries[intercept_mean,intercept_sd,logp_mean,logp_last]':
                        chain_summaries(gathered),
                    'gathered': {k: list(v.shape) for k, v in gathered.items()},
                    'X_mean_rms_between_chains': (round(float(np.sqrt(((xm - xm.mean(0)) ** 2).mean())), 5)
                                                  if world * C > 1 else 0.0)}
                if world > 1 or args.force_collectives:
                    # what the collective backend itself reported for rank 0's group and what its data
                    # collectives moved (network broadcast, starting values, final gather): a first run on
                    # an 8-GPU node is checkable from this line alone
                    line['collectives'] = group.describe()
                if name == 'hdp':       # queues of the timed run's calls (2: the intercept's pass beside the tail)
                    line['config']['hdp_queues'] = wl.model.chain_.hdp_queues()
                if name == 'lsm':
                    line['config']['iteration'] = ('sweep + procrustes + centring + intercept MH + '
                                                   'logp trace')
                    line['iteration_fp64_valu'] = iteration_valu_fraction(1e3 * elapsed / K, args)
                elif name in ('cc', 'ccu'):
                    line['config']['network'] = wl.network
                    line['config']['degrees'] = wl.degree_stats
                    line['config']['iteration'] = ('sweep (case-control partial likelihoods) + centring + '
                                                   'intercept_in / intercept_out / radii MH around '
                                                   'case-control log-likelihood passes + logp trace; '
                                                   'controls redrawn every 100 iterations')
                else:
                    line['config']['iteration'] = ('sweep (mixture prior) + centring + intercept MH + '
                                                   'label block update + HDP auxiliary / conjugate / '
                                                   'hyper-parameter draws + logp trace')
                    line['config']['loop'] = getattr(wl.model, 'loop_kind_', 'host-driven')
                    line['n_clusters_used_last'] = [float(v[-1]) for v in gathered['n_clusters_used']]
                line.update(extra)
                lines.append(line)
            wl.close()

    def run_monks():
        """BASELINE.json configs[0]: DynamicNetworkLSM on Sampson's monks (T=3, N=18, d=2), 500
        iterations through the estimator's fit(); its posterior summary beside the reference's
        between-seed envelope (tests/golden/monks_envelopes.npz: the reference's own 8 seeds)"""
        from dynetlsm_amd import DynamicNetworkLSM
        gold = os.path.join(ROOT, 'tests', 'golden')
        Y = np.load(os.path.join(gold, 'monks.npz'))['Y_undirected']
        env = np.load(os.path.join(gold, 'monks_envelopes.npz'))
        cols = [str(c) for c in env['columns']]
        m = DynamicNetworkLSM(n_iter=500, tune=250, burn=250, random_state=0, device=local_rank)
        t0 = time.perf_counter()
        m.fit(Y)
        wall = time.perf_counter() - t0
        n_total = m.logps_.shape[0]
        keep = slice(500, None)
        d = np.sqrt(((m.Xs_[keep, :, :, None, :] - m.Xs_[keep, :, None, :, :]) ** 2).sum(-1))
        got = {'intercept_mean': float(m.intercepts_[keep, 0].mean()),
               'intercept_sd': float(m.intercepts_[keep, 0].std()),
               'logp_mean': float(m.logps_[keep].mean()), 'logp_sd': float(m.logps_[keep].std()),
               'mean_pairwise_distance': float(d.mean())}
        ref = {c: [round(float(env['summaries'][:, k].mean()), 4),
                   round(float(env['summaries'][:, k].std(ddof=1)), 4)] for k, c in enumerate(cols)}
        loop = getattr(m, 'loop_seconds_', None)
        lines.append({
            'metric': 'Gibbs iterations/sec, DynamicNetworkLSM on Sampson\'s monks T=3 N=18 d=2',
            'value': round((n_total - 1) / (loop if loop else wall), 1), 'unit': 'Gibbs iterations/s',
            'n_gpus': 1, 'steps': n_total - 1, 'dtype': 'f64', 'data': 'Sampson monks (tests/golden/monks.npz)',
            'config': {'workload': 'DynamicNetworkLSM(n_iter=500, tune=250, burn=250).fit(monks), '
                                   'undirected, 1 chain on 1 GPU (launch bound at N = 18)'},
            'fit_seconds': round(wall, 3),
            'posterior_summary': {k: round(v, 4) for k, v in got.items()},
            'reference_between_seed_[mean,sd]': ref,
            'within_4sd_of_reference': {k: bool(abs(got[k] - ref[k][0]) <= 4 * ref[k][1] + 0.02 * abs(ref[k][0]))
                                        for k in got if k in ref}})
        m.chain_.close()

    for i, name in enumerate(models):
        if i == 0 or world > 1:
            run_model(name)         # ranks must stay in step: a failure ends the job
            continue
        try:                        # single process: an attached config may fail without
            run_model(name)         # taking the headline line with it
        except Exception as e:      # noqa: BLE001
            import traceback
            traceback.print_exc(file=sys.stderr)
            lines.append({'metric': name, 'error': '%s: %s' % (type(e).__name__, e)})
    if rank == 0 and world == 1 and args.model == 'all':
        try:
            run_monks()
        except Exception as e:      # noqa: BLE001
            import traceback
            traceback.print_exc(file=sys.stderr)
            lines.append({'metric': 'monks', 'error': '%s: %s' % (type(e).__name__, e)})
    if rank == 0 and world == 1 and args.model == 'all' and args.chains_per_gpu == 1:
        # chains-per-GPU scaling is the sweep's defensible utilisation figure - config 2 again with four
        # chains on the GPU (one handle, stream and host thread each), the same window
        try:
            # (in a process of its own, as `python bench.py --model lsm --chains-per-gpu 4` measures it: inside this
            # one - behind the streams of five workloads - the four chains' queues overlapped far less: 3900 - 4200
            # it/s aggregate against 7000)
            import subprocess
            steps4 = max(args.steps, 200)       # (four host threads: a 20-step window measures their start)
            cmd = [sys.executable, os.path.abspath(__file__), '--model', 'lsm', '--chains-per-gpu', '4', '--no-cpu',
                   '--profile-steps', '0', '--steps', str(steps4), '--warmup', str(max(args.warmup, 20)),
                   '--settle-steps', str(args.settle_steps), '--T', str(args.T), '--N', str(args.N), '--D', str(args.D),
                   '--density', str(args.density)]
            out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600).stdout.decode()
            l4 = json.loads([ln for ln in out.splitlines() if ln.startswith('{')][-1])

            class _A4(object):
                steps = steps4
            a4 = _A4()
            c2 = lines[0]
            iv = c2.get('iteration_fp64_valu') or {}
            lines.append({'metric': l4['metric'], 'chains_per_gpu': 4, 'steps': a4.steps, 'aggregate_it_per_s': l4['value'],
                          'per_chain_it_per_s': round(l4['value'] / 4.0, 1),
                          'ratio_to_one_chain': round(l4['value'] / c2['value'], 3),
                          'ms_per_step_all_chains': l4['ms_per_step'],
                          'config': {'workload': l4['config']['workload']},
                          # the sweep kernel's issue-slot fraction and the iteration's float64 fraction at one
                          # chain, scaled by the aggregate rate: how busy the vector pipes are when the launch gaps
                          # of one chain are filled by the others
                          'frac_issue_slots': (round((c2.get('roofline') or {}).get('frac_issue_slots') *
                                                     l4['value'] / c2['value'], 4)
                                               if (c2.get('roofline') or {}).get('frac_issue_slots') else None),
                          'iteration_frac_fp64_valu': (round(iv['frac'] * l4['value'] / c2['value'], 4)
                                                       if 'frac' in iv else None)})
        except Exception as e:      # noqa: BLE001
            import traceback
            traceback.print_exc(file=sys.stderr)
            lines.append({'metric': 'lsm, 4 chains per GPU', 'error': '%s: %s' % (type(e).__name__, e)})
    if rank == 0:
        head = lines[0]
        if len(lines) > 1:
            head['extra_configs'] = lines[1:]
            # the other configurations' figures as flat numbers inside `config` (the driver's record keeps `config`
            # whole and only the tail of everything else)
            def val(pred):
                for ln in lines[1:]:
                    if pred(ln) and 'error' not in ln:
                        return ln.get('value', ln.get('aggregate_it_per_s'))
                return None
            head['config'].update({
                'also_c3_hdp_lpcm_it_per_s': val(lambda ln: 'HDP' in ln['metric'] or 'hdp' in ln['metric'].lower()),
                'also_c4_model_network_it_per_s': val(lambda ln: 'model-drawn' in ln['metric']),
                'also_c4_degree_regular_network_it_per_s': val(lambda ln: 'degree-regular' in ln['metric']),
                'also_c1_monks_it_per_s': val(lambda ln: 'monks' in ln['metric']),
                'also_c2_4_chains_per_gpu_aggregate_it_per_s': val(lambda ln: ln.get('chains_per_gpu') == 4)})
        print(json.dumps(head), flush=True)
    group.close()


def main():
    args = parse()
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(launch_ranks(args))
    run_rank(args)


if __name__ == '__main__':
    main()
