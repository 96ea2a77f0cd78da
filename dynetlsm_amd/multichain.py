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
"""
import os

import numpy as np

__all__ = ['ChainGroup', 'init_chain_group']


class ChainGroup(object):
    """Rank / device bookkeeping of one process of the chain group."""

    def __init__(self, rank, world, local_rank, backend, dist=None, torch=None, force=False):
        self.rank, self.world, self.local_rank = rank, world, local_rank
        self.backend = backend
        self._dist, self._torch = dist, torch
        # force: go through the collectives even in a group of one (a single-GPU box can then
        # exercise the RCCL path: device tensors, stream ordering, library initialisation)
        self._solo = world == 1 and not force
        # Control traffic - barriers and the timing reduction around a timed region - goes
        # through a gloo group of the same ranks when the data backend is RCCL: a device
        # barrier is an all-reduce launch plus a device synchronisation (about a millisecond,
        # 3 - 4 % of a 100-iteration window), a host barrier over loopback a few tens of
        # microseconds.  The network broadcast and the final gather stay on the data backend.
        self._ctl = None
        if not self._solo and backend == 'nccl':
            self._ctl = dist.new_group(backend='gloo')

    @property
    def chain_id(self):
        return self.rank

    @property
    def device(self):
        return self.local_rank

    def _tensor_device(self):
        t = self._torch
        return t.device('cuda', self.local_rank) if self.backend == 'nccl' else t.device('cpu')

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
            if not np.all((Y == 0) | (Y == 1)):
                raise ValueError('network entries must be 0 / 1')
            hdr = t.tensor(Y.shape, dtype=t.int64, device=dev)
        self._dist.broadcast(hdr, src)
        shp = tuple(int(v) for v in hdr.cpu())
        if self.rank == src:
            buf = t.from_numpy(Y.astype(np.uint8)).to(dev)
        else:
            buf = t.empty(shp, dtype=t.uint8, device=dev)
        self._dist.broadcast(buf, src)
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
        self._dist.broadcast(buf, src)
        if self.backend == 'nccl':
            t.cuda.current_stream().synchronize()      # the chain copies on its own stream
        if self.rank != src:
            chain.set_network_packed(buf.data_ptr(), n)

    def broadcast_array(self, a, src=0):
        """small float64 array (same shape known on every rank)"""
        if self._solo:
            return np.asarray(a, dtype=np.float64)
        t = self._torch
        buf = t.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self._tensor_device())
        self._dist.broadcast(buf, src)
        return buf.cpu().numpy()

    def gather_arrays(self, a):
        """list (one per rank) of equally shaped float64 arrays, on every rank"""
        a = np.ascontiguousarray(a, dtype=np.float64)
        if self._solo:
            return [a]
        t = self._torch
        mine = t.from_numpy(a).to(self._tensor_device())
        out = [t.empty_like(mine) for _ in range(self.world)]
        self._dist.all_gather(out, mine)
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
        if self._ctl is not None:
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
    backend and route through its collectives even when WORLD_SIZE is 1."""
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    import torch
    import torch.distributed as dist
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl':
        torch.cuda.set_device(local_rank)
    if world > 1 or force:
        kw = {}
        if backend == 'nccl':
            kw['device_id'] = torch.device('cuda', local_rank)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return ChainGroup(rank, world, local_rank, backend, dist, torch, force=force)
