"""Independent MCMC chains sharded one per GPU.

Chains never talk inside an iteration.  Collectives (RCCL over xGMI when the
backend is ``nccl``; ``gloo`` on CPU for tests) appear only at the edges:

  * ``broadcast_chain_network``  the network of rank 0's chain -> the chains of the
    other ranks as the engine holds it, 1 bit per dyad (5 MB at T=10, N=2000 instead
    of 320 MB of float64), device to device: rank 0 uploads and packs once, the
    packed words travel over xGMI and land in the other chains without touching a
    host (``dlsm_get/set_network_packed``);
  * ``broadcast_network``  the float64 tensor itself for host-side consumers
    (sent as uint8, widened locally);
  * ``gather_arrays`` / ``gather_results``  per-chain summaries and traces
    (posterior mean positions, log-posterior and intercept traces ...) -> every
    rank (all_gather of equally shaped float64 tensors).

Launch with ``python -m torch.distributed.run --nproc-per-node N script.py``;
rank r drives GPU LOCAL_RANK with Philox chain id r.

``fit_chains(estimator, Y, n_chains)`` is the estimator-level call: N chains of one model,
one per GPU, in one call - what the reference does serially with ``for i in range(n_reps):
... random_state=i`` (examples/homogeneous_simulation.py:177-184).  It joins the ranks of a
launcher when there is one (RANK set), and otherwise spawns one child process per chain itself
BEFORE any GPU call of its own (``launch_ranks``: the parent never touches the device); rank 0's
network is broadcast over RCCL, every rank runs ``estimator.fit`` with Philox chain id and
``random_state`` offset by its rank, the scalar traces / posterior means / selected partitions
are all-gathered, and the result carries the split R-hat of the scalar traces across chains.
"""
import os
import pickle
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

__all__ = ['ChainGroup', 'init_chain_group', 'fit_chains', 'ChainsResult', 'launch_ranks',
           'split_rhat']


class ChainGroup(object):
    """Rank / device bookkeeping of one process of the chain group."""

    def __init__(self, rank, world, local_rank, backend, dist=None, torch=None, force=False, data_group=None):
        self.rank, self.world, self.local_rank = rank, world, local_rank
        self.backend = backend
        self._dist, self._torch = dist, torch
        # data_group: the RCCL group of the data collectives when the DEFAULT group is the gloo control group
        # (init_chain_group brings the control group up first and decides about RCCL over it); None: the default
        # group carries the data as well
        self._grp = data_group
        # force: go through the collectives even in a group of one (a single-GPU box can then
        # exercise the RCCL path: device tensors, stream ordering, library initialisation)
        self._solo = world == 1 and not force
        # Control traffic - barriers and the timing reduction around a timed region - goes
        # through a gloo group of the same ranks when the data backend is RCCL: a device
        # barrier is an all-reduce launch plus a device synchronisation (about a millisecond,
        # 3 - 4 % of a 100-iteration window), a host barrier over loopback a few tens of
        # microseconds.  The network broadcast and the final gather stay on the data backend.
        self._ctl = None
        if not self._solo and backend == 'nccl' and data_group is None:
            self._ctl = dist.new_group(backend='gloo')
        # what the data collectives of this rank moved and how long they took (host clock around
        # the call + the synchronisation that ends it): `describe()` puts them on the N > 1 bench
        # line, so that the first run on a real 8-GPU node can be checked from its JSON alone
        self.stats = {'broadcast_calls': 0, 'broadcast_bytes': 0, 'broadcast_ms': 0.0,
                      'network_broadcast_bytes': 0, 'network_broadcast_ms': 0.0,
                      'gather_calls': 0, 'gather_bytes_per_rank': 0, 'gather_ms': 0.0}

    @property
    def chain_id(self):
        return self.rank

    @property
    def device(self):
        return self.local_rank

    def _tensor_device(self):
        t = self._torch
        return t.device('cuda', self.local_rank) if self.backend == 'nccl' else t.device('cpu')

    def describe(self):
        """what the backend itself reports for this group (not what the caller asked for) and the
        traffic counters: {'world_size', 'rank', 'backend', 'control_backend', 'device', ...stats}"""
        d = {'world_size': self.world, 'rank': self.rank, 'backend': self.backend, 'control_backend': None,
             'device': None, 'solo': bool(self._solo)}
        if not self._solo:
            d['world_size'] = int(self._dist.get_world_size())
            d['rank'] = int(self._dist.get_rank())
            d['backend'] = str(self._dist.get_backend(self._grp) if self._grp is not None else self._dist.get_backend())
            d['control_backend'] = (str(self._dist.get_backend(self._ctl)) if self._ctl is not None
                                    else str(self._dist.get_backend()))
            d['device'] = str(self._tensor_device())
        d.update({k: (round(v, 4) if isinstance(v, float) else v) for k, v in self.stats.items()})
        if getattr(self, 'fallback_reason', None):
            d['requested_backend'] = 'nccl'
            d['fallback_reason'] = self.fallback_reason
        return d

    def _timed(self, kind, nbytes, t0):
        if self.backend == 'nccl':
            self._torch.cuda.current_stream().synchronize()
        ms = 1e3 * (time.perf_counter() - t0)
        if kind == 'gather':
            self.stats['gather_calls'] += 1; self.stats['gather_bytes_per_rank'] += int(nbytes)
            self.stats['gather_ms'] += ms
        else:
            self.stats['broadcast_calls'] += 1; self.stats['broadcast_bytes'] += int(nbytes)
            self.stats['broadcast_ms'] += ms
            if kind == 'network':
                self.stats['network_broadcast_bytes'] += int(nbytes); self.stats['network_broadcast_ms'] += ms

    def barrier(self):
        if not self._solo:
            if self._ctl is not None:
                self._dist.barrier(group=self._ctl)
            else:
                self._dist.barrier()

    def broadcast_network(self, Y, shape=None, src=0):
        """Y (T, N, N) float64 on ``src`` (None elsewhere) -> float64 on all."""
        if self._solo:
            return np.ascontiguousarray(Y, dtype=np.float64)
        t = self._torch
        dev = self._tensor_device()
        hdr = t.zeros(3, dtype=t.int64, device=dev)
        if self.rank == src:
            Y = np.ascontiguousarray(Y, dtype=np.float64)
            if not np.all((Y == 0) | (Y == 1) | (Y == -1)):      # -1: the reference's missing code
                raise ValueError('network entries must be 0 / 1 (or -1 for a missing dyad)')
            hdr = t.tensor(Y.shape, dtype=t.int64, device=dev)
        self._dist.broadcast(hdr, src, group=self._grp)
        shp = tuple(int(v) for v in hdr.cpu())
        if self.rank == src:
            buf = t.from_numpy(Y.astype(np.int8)).to(dev)
        else:
            buf = t.empty(shp, dtype=t.int8, device=dev)
        t0 = time.perf_counter()
        self._dist.broadcast(buf, src, group=self._grp)
        self._timed('network', buf.numel(), t0)
        return buf.cpu().numpy().astype(np.float64)

    def broadcast_chain_network(self, chain, src=0):
        """The packed network of rank ``src``'s chain (already uploaded there) into the
        chain of every other rank; same shape and model on every rank.  With the RCCL
        backend the words never leave device memory."""
        if self._solo:
            return
        t = self._torch
        n = chain.network_packed_words()
        buf = t.empty(n, dtype=t.int32, device=self._tensor_device())
        if self.rank == src:
            chain.get_network_packed(buf.data_ptr(), n)
        t0 = time.perf_counter()
        self._dist.broadcast(buf, src, group=self._grp)
        self._timed('network', 4 * n, t0)              # (synchronises: the chain copies on its own stream)
        if self.rank != src:
            chain.set_network_packed(buf.data_ptr(), n)

    def broadcast_array(self, a, src=0):
        """small float64 array (same shape known on every rank)"""
        if self._solo:
            return np.asarray(a, dtype=np.float64)
        t = self._torch
        buf = t.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self._tensor_device())
        t0 = time.perf_counter()
        self._dist.broadcast(buf, src, group=self._grp)
        self._timed('array', 8 * buf.numel(), t0)
        return buf.cpu().numpy()

    def gather_arrays(self, a):
        """list (one per rank) of equally shaped float64 arrays, on every rank"""
        a = np.ascontiguousarray(a, dtype=np.float64)
        if self._solo:
            return [a]
        t = self._torch
        mine = t.from_numpy(a).to(self._tensor_device())
        out = [t.empty_like(mine) for _ in range(self.world)]
        t0 = time.perf_counter()
        self._dist.all_gather(out, mine, group=self._grp)
        self._timed('gather', 8 * mine.numel(), t0)
        return [o.cpu().numpy() for o in out]

    def gather_results(self, results):
        """dict of equally shaped per-chain float arrays (e.g. ``X_mean`` (T, N, D),
        ``logps`` (n,), ``intercepts`` (n, 1)) -> dict of (world, ...) arrays holding every
        chain's entry, on every rank: the final gather of SURVEY.md 8e."""
        return {k: np.stack(self.gather_arrays(results[k])) for k in sorted(results)}

    def max_over_ranks(self, x):
        if self._solo:
            return float(x)
        t = self._torch
        if self._ctl is not None or self._grp is not None:      # (a host all-reduce over the gloo control group)
            v = t.tensor([float(x)], dtype=t.float64)
            self._dist.all_reduce(v, op=self._dist.ReduceOp.MAX, group=self._ctl)
            return float(v[0])
        v = t.tensor([float(x)], dtype=t.float64, device=self._tensor_device())
        self._dist.all_reduce(v, op=self._dist.ReduceOp.MAX)
        return float(v.cpu()[0])

    def close(self):
        if not self._solo and self._dist.is_initialized():
            self._dist.destroy_process_group()


def init_chain_group(backend=None, force=False):
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_ADDR / MASTER_PORT (torch.distributed.run sets them).  ``force``: initialise the
    backend and route through its collectives even when WORLD_SIZE is 1.

    ``backend='nccl'`` (RCCL): the gloo CONTROL group comes up first, over the rendezvous as given - barriers
    and the timing reductions use it anyway - then the RCCL group is made as a second group of the same ranks
    and probed (one element summed over the ranks), and the ranks AGREE over the control group whether it
    works: either every rank uses RCCL or none does (round-5 advice: a fallback each rank decided on its own
    left the others waiting in a collective, and looked for a second port).  RCCL that does not come up is an
    ERROR unless DLSM_ALLOW_BACKEND_FALLBACK=1: then the setup collectives - none of them is on the data path -
    go through the control group and ``describe()`` (the bench line's `collectives` block) says so."""
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    import datetime
    import torch
    import torch.distributed as dist
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    fallback_reason = None
    data_group = None
    if world > 1 or force:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('gloo', rank=rank, world_size=world)
        if backend == 'nccl':
            def agree(ok):
                """every rank's flag, the same answer everywhere (over the gloo control group)"""
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                return int(flag[0]) == 1
            # phase A: what a rank can check alone (its device) - no rank enters an RCCL call unless all can
            mine = None
            try:
                if not torch.cuda.is_available():
                    raise RuntimeError('no GPU visible to this rank')
                torch.cuda.set_device(local_rank)
                torch.zeros(1, device=torch.device('cuda', local_rank))
            except Exception as exc:        # noqa: BLE001
                mine = '%s: %s' % (type(exc).__name__, str(exc)[:300])
            ok = agree(mine is None)
            if ok:
                # phase B: the communicator and one all-reduce through it (a bounded wait: a rank whose
                # communicator fails leaves the others in this collective)
                try:
                    data_group = dist.new_group(backend='nccl', timeout=datetime.timedelta(
                        seconds=float(os.environ.get('DLSM_RCCL_PROBE_TIMEOUT', '120'))))
                    probe = torch.ones(1, device=torch.device('cuda', local_rank))
                    dist.all_reduce(probe, group=data_group)
                    torch.cuda.synchronize()
                    if int(probe.item()) != world:
                        raise RuntimeError('RCCL all_reduce probe returned %r for %d ranks' % (probe.item(), world))
                except Exception as exc:    # noqa: BLE001
                    mine = '%s: %s' % (type(exc).__name__, str(exc)[:300])
                ok = agree(mine is None)
            if not ok:
                fallback_reason = mine or 'RCCL did not come up on another rank'
                if os.environ.get('DLSM_ALLOW_BACKEND_FALLBACK') != '1':
                    dist.destroy_process_group()
                    raise RuntimeError('dynetlsm_amd: the RCCL process group did not come up (rank %d: %s); '
                                       'DLSM_ALLOW_BACKEND_FALLBACK=1 sends the setup collectives through gloo '
                                       '(no collective is on the chains\' data path)' % (rank, fallback_reason))
                print('dynetlsm_amd: RCCL process group failed on rank %d (%s); collectives go through gloo'
                      % (rank, fallback_reason), file=sys.stderr)
                data_group = None
                backend = 'gloo'
    elif backend == 'nccl':
        torch.cuda.set_device(local_rank)
    g = ChainGroup(rank, world, local_rank, backend, dist, torch, force=force, data_group=data_group)
    g.fallback_reason = fallback_reason
    return g


# ------------------------------------------------------------------------------------
# N chains of one estimator, one per GPU
# ------------------------------------------------------------------------------------
def split_rhat(chains):
    """split R-hat (Gelman et al., BDA3 11.4) of a scalar trace over chains (m, n): every chain
    cut in two halves, between- against within-sequence variance"""
    c = np.asarray(chains, dtype=np.float64)
    if c.ndim != 2 or c.shape[1] < 4:
        return float('nan')
    half = c.shape[1] // 2
    q = np.concatenate([c[:, :half], c[:, half:2 * half]], axis=0)
    n = q.shape[1]
    W = q.var(axis=1, ddof=1).mean()
    B = n * q.mean(axis=1).var(ddof=1)
    if not W > 0.0:
        return 1.0 if B == 0.0 else float('inf')
    return float(np.sqrt(((n - 1.0) / n * W + B / n) / W))


def visible_gpu_count():
    """GPUs this process would see, counted WITHOUT a HIP / HSA call (the parent of
    ``launch_ranks`` must not initialise the runtime its children are about to use:
    ``torch.cuda.device_count()`` falls back to hipGetDeviceCount when amdsmi is absent): the
    kernel driver's topology nodes with SIMDs whose render node (``/dev/dri/renderD<drm_render_minor>``) this
    process may open - a container may see the topology of GPUs it was not given - cut by every one of
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES that is set (HIP's filter applies on top
    of ROCR's: the smallest count)."""
    n = 0
    root = '/sys/class/kfd/kfd/topology/nodes'
    try:
        for node in sorted(os.listdir(root)):
            try:
                with open(os.path.join(root, node, 'properties')) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            except OSError:
                continue
            if int(props.get('simd_count', '0')) <= 0:
                continue
            minor = int(props.get('drm_render_minor', '-1'))
            if minor >= 0 and os.path.isdir('/dev/dri'):
                dev = '/dev/dri/renderD%d' % minor
                if not (os.path.exists(dev) and os.access(dev, os.R_OK | os.W_OK)):
                    continue                    # (a GPU of the host this container cannot open)
            n += 1
    except OSError:
        n = 0
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            ids = [x for x in v.split(',') if x.strip() != '']
            n = min(n, len(ids)) if n else len(ids)
    return n


def _free_port():
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def launch_ranks(argv, n, env_extra=None, relay_rank0=True, local_ranks=None, timeout=None):
    """Start ``argv`` n times as child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    set (one rank per GPU) and wait for them: the caller makes no GPU call, so nothing is
    inherited.  Rank 0 inherits stdout when ``relay_rank0`` (its output is the job's), the
    others' goes to stderr.  When a rank fails the others are terminated - exactly the PIDs
    started here - and the first non-zero exit code is returned.  ``timeout`` (seconds): ranks
    still alive after it are terminated, then killed, and 124 is returned (a rank wedged in a
    collective must not hold the caller for ever)."""
    port = _free_port()
    deadline = None if timeout is None else time.monotonic() + float(timeout)
    procs = []
    for r in range(n):
        lr = r if local_ranks is None else local_ranks[r]
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(lr), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # ranks that share a device keep the HDP-LPCM loop on one queue: its second queue (the
        # intercept's likelihood pass beside the label update and the conjugate draws) pays on a
        # device of its own and loses badly when the queues of two PROCESSES take turns on one GPU
        # (measured: two ranks on one MI355X 4442 it/s on one queue each, 1824 on two)
        if local_ranks is not None and list(local_ranks).count(lr) > 1:
            env.setdefault('DLSM_HDP_QUEUES', '1')
            # ... and no role waits INSIDE a launch for a workgroup another process's launches may keep off the
            # CUs: the case-control sweep's helper workgroups and the pipelined sweep's served cross products (the
            # engine decides by the chains alive in ITS process; it cannot see the other processes)
            env.setdefault('DLSM_CC_HELPERS', '0')
            env.setdefault('DLSM_PIPE_XSERVE', '0')
        if env_extra:
            env.update(env_extra)
        procs.append(subprocess.Popen(list(argv), env=env,
                                      stdout=None if (r == 0 and relay_rank0) else sys.stderr))
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print('rank %d exited with %d; stopping the others' % (r, code), file=sys.stderr)
                for q in alive:
                    procs[q].terminate()
        if alive and deadline is not None and time.monotonic() > deadline:
            print('launch_ranks: %d rank(s) still running after %.0f s; stopping them'
                  % (len(alive), float(timeout)), file=sys.stderr)
            for q in alive:
                procs[q].terminate()
            t_kill = time.monotonic() + 5.0
            while any(procs[q].poll() is None for q in alive) and time.monotonic() < t_kill:
                time.sleep(0.05)
            for q in alive:
                if procs[q].poll() is None:
                    procs[q].kill()
                procs[q].wait()
            return rc or 124
        time.sleep(0.05)
    return rc


class ChainsResult(object):
    """What ``fit_chains`` returns, the same on every rank:

    ``n_chains``; ``traces`` - dict of (n_chains, n_total[, k]) arrays (``logps``, ``intercepts``,
    ``lambdas`` when the model has them); ``X_mean`` (n_chains, T, N, D) posterior-mean (or
    selected) positions; ``z`` (n_chains, T, N) selected partitions when the model has labels;
    ``n_burn``; ``rhat`` - split R-hat of every scalar trace over the chains' kept iterations;
    ``logp_mean`` per chain and ``best_chain`` (the highest); ``seconds`` per chain;
    ``estimator`` - the fitted estimator of THIS rank (rank 0's when the chains were spawned:
    with its large traces dropped unless ``keep_traces``)."""

    def __init__(self, gathered, n_burn, estimator=None):
        self.traces = {k: gathered[k] for k in ('logps', 'intercepts', 'lambdas') if k in gathered}
        self.X_mean = gathered.get('X_mean')
        self.z = gathered['z'].astype(np.int64) if 'z' in gathered else None
        self.seconds = gathered['seconds'][:, 0]
        self.n_chains = int(self.seconds.shape[0])
        self.n_burn = int(n_burn)
        self.estimator = estimator
        self.rhat = {}
        kept = slice(self.n_burn, None)
        for name, tr in self.traces.items():
            tr = tr[:, kept]
            if tr.ndim == 2:
                self.rhat[name] = split_rhat(tr)
            else:
                for j in range(tr.shape[2]):
                    self.rhat['%s[%d]' % (name, j)] = split_rhat(tr[:, :, j])
        self.logp_mean = self.traces['logps'][:, kept].mean(axis=1)
        self.best_chain = int(np.argmax(self.logp_mean))

    def summary(self):
        return {'n_chains': self.n_chains, 'n_burn': self.n_burn,
                'rhat': {k: round(v, 4) for k, v in self.rhat.items()},
                'logp_mean': [round(float(v), 3) for v in self.logp_mean],
                'best_chain': self.best_chain, 'seconds': [round(float(v), 3) for v in self.seconds]}


def _chain_results(est, seconds):
    """the per-chain arrays that travel in the final gather (float64, equal shapes on every rank)"""
    out = {'logps': np.asarray(est.logps_, dtype=np.float64),
           'intercepts': np.asarray(est.intercepts_, dtype=np.float64),
           'seconds': np.array([float(seconds)])}
    if getattr(est, 'lambdas_', None) is not None:
        out['lambdas'] = np.asarray(est.lambdas_, dtype=np.float64)
    xm = getattr(est, 'X_mean_', None)
    out['X_mean'] = np.asarray(xm if xm is not None else est.X_, dtype=np.float64)
    if getattr(est, 'z_', None) is not None:
        out['z'] = np.asarray(est.z_, dtype=np.float64)
    return out


def _kept_from(est):
    """index of the first kept row of the estimator's STORED traces: ``n_burn_`` counts
    iterations, a thinned estimator stores every ``thin``-th of them (hdp_lpcm.py:1072-1083 thins
    the traces, :1085 on then works with ceil(n_burn / thin), hdp_lpcm.py:465 - as ``_finish`` and forecast.py do)"""
    n_burn = int(getattr(est, 'n_burn_', 0) or 0)
    thin = int(getattr(est, 'thin', None) or 1)
    n_rows = int(np.asarray(est.logps_).shape[0])
    return max(0, min(-(-n_burn // thin), n_rows - 1))          # ceil, as the reference's n_burn_ (hdp_lpcm.py:465)


def _fit_as_rank(estimator, Y, init, group, seed_stride=1):
    """this rank's part of ``fit_chains``: network from rank 0, fit with the rank's chain id and
    seed, final gather"""
    Y = group.broadcast_network(Y if group.rank == 0 else None)      # (shape header + int8 entries)
    est = estimator
    est.chain_id = int(getattr(est, 'chain_id', 0) or 0) + group.rank
    est.device = group.device
    rs = getattr(est, 'random_state', None)
    if rs is None or isinstance(rs, (int, np.integer)):
        est.random_state = (0 if rs is None else int(rs)) + seed_stride * group.rank
    t0 = time.perf_counter()
    est.fit(Y, init=init) if init is not None else est.fit(Y)
    secs = time.perf_counter() - t0
    gathered = group.gather_results(_chain_results(est, secs))
    return ChainsResult(gathered, _kept_from(est), est)


def fit_chains(estimator, Y, n_chains=None, init=None, backend=None, share_device0=False,
               keep_traces=False, timeout=None):
    """Fit ``n_chains`` independent chains of ``estimator`` (an unfitted DynamicNetworkLSM /
    DynamicNetworkHDPLPCM / DynamicNetworkLPCM) to the network ``Y``, one chain per GPU of this
    node, and gather them: returns a ``ChainsResult``.

    * Under a launcher (``RANK`` is set: ``torch.distributed.run`` or ``launch_ranks``) every rank
      calls this; only rank 0 needs ``Y`` (the others may pass None): it reaches them as one RCCL
      broadcast.  Chain r uses Philox chain id ``estimator.chain_id + r`` and ``random_state + r``.
    * Otherwise the call spawns its ranks itself, one child process per chain, before making any
      GPU call (``n_chains=None``: one per visible device), hands them the estimator, ``init`` and
      ``Y`` through a temporary directory, and returns rank 0's gathered result.
      ``share_device0`` puts every rank on GPU 0 over gloo (a single-GPU box; tests).

    ``backend``: 'nccl' (= RCCL; the default on a GPU box) or 'gloo'."""
    if 'RANK' in os.environ:
        if share_device0:
            os.environ['LOCAL_RANK'] = '0'
            if backend is None:
                backend = 'gloo'        # (RCCL refuses two ranks on one device)
        group = init_chain_group(backend=backend)
        try:
            return _fit_as_rank(estimator, Y, init, group)
        finally:
            group.close()
    if n_chains is None:
        n_chains = max(1, visible_gpu_count())
    n_chains = int(n_chains)
    if backend is None:
        backend = 'gloo' if share_device0 else 'nccl'
    with tempfile.TemporaryDirectory(prefix='dlsm_chains_') as job:
        Yarr = np.ascontiguousarray(Y, dtype=np.float64)
        np.save(os.path.join(job, 'Y.npy'), Yarr.astype(np.int8))      # 0 / 1 / -1 (missing)
        with open(os.path.join(job, 'job.pkl'), 'wb') as f:
            pickle.dump(dict(estimator=estimator, init=init, backend=backend,
                             share_device0=bool(share_device0), keep_traces=bool(keep_traces)), f)
        env = {'PYTHONPATH': os.pathsep.join([p for p in sys.path if p] +
                                             [os.environ.get('PYTHONPATH', '')])}
        # (the module is imported, not run as __main__: the pickled result must name
        # dynetlsm_amd.multichain.ChainsResult)
        code = 'import sys; from dynetlsm_amd.multichain import _rank_main; _rank_main(sys.argv[1])'
        rc = launch_ranks([sys.executable, '-c', code, job], n_chains,
                          env_extra=env, relay_rank0=False,
                          local_ranks=[0] * n_chains if share_device0 else None, timeout=timeout)
        if rc != 0:
            raise RuntimeError('fit_chains: a rank exited with code %d (its messages are on stderr)' % rc)
        with open(os.path.join(job, 'result.pkl'), 'rb') as f:
            return pickle.load(f)


def _rank_main(job):
    """entry point of a rank spawned by ``fit_chains``"""
    with open(os.path.join(job, 'job.pkl'), 'rb') as f:
        spec = pickle.load(f)
    group = init_chain_group(backend=spec['backend'])
    try:
        Y = None
        if group.rank == 0:
            Y = np.load(os.path.join(job, 'Y.npy')).astype(np.float64)
        res = _fit_as_rank(spec['estimator'], Y, spec['init'], group)
        if group.rank == 0:
            est = res.estimator
            if est is not None:
                ch = getattr(est, 'chain_', None)
                if hasattr(est, 'release_device_trace'):
                    est.release_device_trace(materialize=spec['keep_traces'])
                elif ch is not None:
                    ch.close()
                for name in ('chain_', '_sums', 'case_control_sampler_'):
                    est.__dict__.pop(name, None)           # device handles do not travel
                if not spec['keep_traces']:
                    for name in ('Xs_', 'zs_', 'weights_', 'cooccurrence_probas_', 'Y_fit_'):
                        est.__dict__.pop(name, None)
            with open(os.path.join(job, 'result.pkl'), 'wb') as f:
                pickle.dump(res, f)
    finally:
        group.close()

