"""world_size-2 gloo test of the chain-sharding collectives (broadcast of the
network, gather of per-chain results).  CPU only: the chains themselves are
stand-ins, the collectives are the code under test."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, %r)
    from dynetlsm_amd.multichain import init_chain_group
    g = init_chain_group(backend='gloo')
    assert g.world == 2 and g.chain_id == g.rank
    Y = None
    if g.rank == 0:
        rng = np.random.RandomState(3)
        Y = (rng.rand(3, 17, 17) < 0.3).astype(np.float64)
    Y = g.broadcast_network(Y)
    assert Y.shape == (3, 17, 17) and Y.dtype == np.float64
    ref = (np.random.RandomState(3).rand(3, 17, 17) < 0.3).astype(np.float64)
    assert np.array_equal(Y, ref)
    x0 = g.broadcast_array(np.arange(4.0) if g.rank == 0 else np.zeros(4))
    assert np.array_equal(x0, np.arange(4.0))
    # the packed broadcast (dlsm_get/set_network_packed through a stand-in chain that keeps
    # its words in host memory; with gloo the buffer handed over is a host pointer)
    import ctypes
    class FakeChain(object):
        def __init__(self, words=None):
            self.words = None if words is None else np.ascontiguousarray(words, dtype=np.uint32)
            self.n = 3 * 17 * 4
        def network_packed_words(self):
            return self.n
        def get_network_packed(self, ptr, n):
            assert n == self.n
            ctypes.memmove(ptr, self.words.ctypes.data, 4 * n)
        def set_network_packed(self, ptr, n):
            assert n == self.n
            self.words = np.zeros(n, dtype=np.uint32)
            ctypes.memmove(self.words.ctypes.data, ptr, 4 * n)
    wref = np.random.RandomState(9).randint(0, 2 ** 32, size=3 * 17 * 4, dtype=np.uint64).astype(np.uint32)
    ch = FakeChain(wref if g.rank == 0 else None)
    g.broadcast_chain_network(ch)
    assert np.array_equal(ch.words, wref)
    res = g.gather_results(dict(X_mean=np.full((3, 17, 2), float(g.rank)),
                                logps=np.arange(5.0) + g.rank))
    assert res['X_mean'].shape == (2, 3, 17, 2) and res['logps'].shape == (2, 5)
    assert res['X_mean'][1].min() == 1.0 and res['logps'][1][0] == 1.0
    # stand-in for a chain: a summary that depends on the chain id
    summ = np.array([g.chain_id, Y.sum() + g.chain_id, 0.5])
    allsum = g.gather_arrays(summ)
    assert len(allsum) == 2
    for r, s in enumerate(allsum):
        assert s[0] == r and s[1] == ref.sum() + r
    assert g.max_over_ranks(10.0 + g.rank) == 11.0
    # what the backend reports and what went through it (the N > 1 bench line carries this)
    d = g.describe()
    assert d['world_size'] == 2 and d['rank'] == g.rank and d['backend'] == 'gloo' and d['device'] == 'cpu'
    assert d['network_broadcast_bytes'] == 3 * 17 * 17 + 4 * 3 * 17 * 4 and d['network_broadcast_ms'] > 0
    assert d['broadcast_calls'] == 3 and d['broadcast_bytes'] == d['network_broadcast_bytes'] + 32
    assert d['gather_calls'] == 3 and d['gather_bytes_per_rank'] == 8 * (3 * 17 * 2 + 5 + 3)
    g.barrier()
    g.close()
    print('rank %%d ok' %% g.rank)
''') % ROOT


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_broadcast_and_gather(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out.decode())
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert 'rank %d ok' % r in out


def test_single_process_group_is_a_no_op():
    sys.path.insert(0, ROOT)
    import numpy as np
    from dynetlsm_amd.multichain import ChainGroup
    g = ChainGroup(0, 1, 0, 'gloo')
    Y = np.eye(3)[None]
    assert np.array_equal(g.broadcast_network(Y), Y)
    assert g.gather_arrays(np.ones(2))[0].tolist() == [1.0, 1.0]
    assert g.max_over_ranks(3.0) == 3.0


def test_bench_self_launch_propagates_rank_failure():
    """`python bench.py --gpus 2` spawns its ranks itself; on a box without a GPU the ranks
    fail loudly (no CPU fallback) and the parent exits non-zero instead of hanging"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip('a GPU is present: covered by the gpu tests')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2',
                        '--backend', 'gloo', '--share-device0', '--steps', '2', '--warmup', '1',
                        '--model', 'lsm', '--N', '64', '--T', '2', '--no-cpu'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode != 0
    assert b'"metric"' not in p.stdout


def test_fit_chains_spawns_its_ranks_and_gathers(tmp_path):
    """multichain.fit_chains from a plain process: one child per chain (gloo, world size 2), the
    network from rank 0 (a missing dyad included), chain ids and seeds offset by the rank, traces and
    selected partitions gathered, split R-hat over the chains"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import numpy as np
    from dynetlsm_amd.multichain import fit_chains, split_rhat
    from standin_estimator import StandInEstimator
    rng = np.random.RandomState(0)
    Y = (rng.rand(3, 9, 9) < 0.3).astype(np.float64)
    Y[1, 2, 3] = -1
    res = fit_chains(StandInEstimator(n_iter=60, random_state=5, chain_id=2), Y, n_chains=2,
                     init=dict(shift=0.25), backend='gloo')
    assert res.n_chains == 2 and res.n_burn == 10
    assert res.traces['logps'].shape == (2, 60) and res.traces['intercepts'].shape == (2, 60, 1)
    assert res.X_mean.shape == (2, 3, 9, 2) and res.z.shape == (2, 3, 9) and res.z.dtype == np.int64
    # every rank ran ITS chain: id and seed offset by the rank, the same network and init everywhere
    for r in range(2):
        want = StandInEstimator(n_iter=60, random_state=5 + r, chain_id=2 + r).fit(Y, init=dict(shift=0.25))
        np.testing.assert_array_equal(res.traces['logps'][r], want.logps_)
        np.testing.assert_array_equal(res.z[r], want.z_)
    assert not np.array_equal(res.traces['logps'][0], res.traces['logps'][1])
    assert res.rhat['logps'] == split_rhat(res.traces['logps'][:, 10:])
    assert 0.9 < res.rhat['logps'] < 1.5 and 'intercepts[0]' in res.rhat
    assert res.best_chain == int(np.argmax(res.traces['logps'][:, 10:].mean(axis=1)))
    assert res.estimator.chain_id == 2 and res.estimator.seen_missing_ == 1      # rank 0's
    assert set(res.summary()) >= {'n_chains', 'rhat', 'best_chain'}


def test_fit_chains_reports_a_failing_rank():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import numpy as np
    import pytest
    from dynetlsm_amd.multichain import fit_chains
    from standin_estimator import StandInEstimator
    with pytest.raises(RuntimeError, match='rank exited'):
        fit_chains(StandInEstimator(n_iter=20, fail_on_chain=1), np.zeros((2, 5, 5)), n_chains=2,
                   backend='gloo')


def test_fit_chains_with_a_thinned_estimator_keeps_its_post_burn_in_rows():
    """round-4 advice: n_burn_ counts iterations, a thinned estimator stores every thin-th row - the
    kept slice starts at n_burn_ // thin (hdp_lpcm.py:1072-1085), not at n_burn_ (an empty slice,
    NaN R-hat and best_chain silently 0 with thin=10, n_iter=5000, burn=2500)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import numpy as np
    from dynetlsm_amd.multichain import fit_chains, split_rhat
    from standin_estimator import StandInEstimator
    Y = np.zeros((2, 6, 6))
    res = fit_chains(StandInEstimator(n_iter=400, random_state=1, thin=10, burn=300), Y, n_chains=2,
                     backend='gloo')
    assert res.traces['logps'].shape == (2, 40) and res.n_burn == 30
    assert np.isfinite(res.rhat['logps']) and np.isfinite(res.logp_mean).all()
    assert res.rhat['logps'] == split_rhat(res.traces['logps'][:, 30:])
    assert res.best_chain == int(np.argmax(res.traces['logps'][:, 30:].mean(axis=1)))
    # a burn-in longer than the chain still leaves the last row (as _finish does)
    res = fit_chains(StandInEstimator(n_iter=50, random_state=1, thin=5, burn=80), Y, n_chains=2,
                     backend='gloo')
    assert res.n_burn == 9 and np.isfinite(res.logp_mean).all()


def test_fit_chains_timeout_stops_a_wedged_rank():
    """a rank that never returns (wedged in a collective, say) does not hold the caller: after
    `timeout` seconds the ranks are terminated and the call raises"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import time
    import numpy as np
    import pytest
    from dynetlsm_amd.multichain import fit_chains
    from standin_estimator import StandInEstimator
    t0 = time.monotonic()
    with pytest.raises(RuntimeError, match='rank exited'):
        fit_chains(StandInEstimator(n_iter=20, hang_on_chain=1), np.zeros((2, 5, 5)), n_chains=2,
                   backend='gloo', timeout=8)
    assert time.monotonic() - t0 < 120


def test_visible_gpu_count_makes_no_runtime_call(monkeypatch):
    """the parent of launch_ranks counts devices from the driver's topology files and the
    *_VISIBLE_DEVICES variables: no HIP / HSA call before its children exist"""
    sys.path.insert(0, ROOT)
    from dynetlsm_amd import multichain
    monkeypatch.delenv('ROCR_VISIBLE_DEVICES', raising=False)
    monkeypatch.delenv('CUDA_VISIBLE_DEVICES', raising=False)
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,1,2')
    n = multichain.visible_gpu_count()
    assert 0 <= n <= 3 and (n == 3 or os.path.isdir('/sys/class/kfd/kfd/topology/nodes'))
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '')
    assert multichain.visible_gpu_count() == 0


def test_rccl_that_does_not_come_up_is_an_error_unless_the_fallback_is_allowed():
    """init_chain_group(backend='nccl') where RCCL cannot start (here: no GPU): an error by default - north_star
    says RCCL, and a first 8-GPU run must not "succeed" without it - and, with DLSM_ALLOW_BACKEND_FALLBACK=1, the
    setup collectives (none is on the chains' data path) go through the gloo control group, `describe()` (the
    bench line's `collectives` block) naming the requested backend and the reason.  The decision is COLLECTIVE:
    two ranks take the same branch (round-5 advice: per-rank fallbacks left the other ranks in a collective)."""
    code = textwrap.dedent('''
        import json, sys
        sys.path.insert(0, %r)
        import numpy as np
        from dynetlsm_amd.multichain import init_chain_group
        g = init_chain_group(backend='nccl', force=True)
        out = g.gather_arrays(np.arange(3.0) + g.rank)
        d = g.describe()
        g.barrier()
        g.close()
        print(json.dumps({'backend': g.backend, 'reason': g.fallback_reason, 'describe': d,
                          'gathered': [a.tolist() for a in out]}))
    ''') % ROOT
    import json

    def run(world, allow):
        port = _free_port()
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                       MASTER_PORT=str(port))
            env.pop('DLSM_ALLOW_BACKEND_FALLBACK', None)
            if allow:
                env['DLSM_ALLOW_BACKEND_FALLBACK'] = '1'
            procs.append(subprocess.Popen([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.PIPE, text=True))
        return [p.communicate(timeout=240) + (p.returncode,) for p in procs]
    for world in (1, 2):
        outs = run(world, allow=True)
        for r, (so, se, rc) in enumerate(outs):
            assert rc == 0, se[-2000:]
            res = json.loads(so.strip().splitlines()[-1])
            assert res['backend'] == 'gloo' and res['reason']
            assert res['describe']['requested_backend'] == 'nccl' and res['describe']['backend'] == 'gloo'
            assert res['gathered'] == [[0.0 + q, 1.0 + q, 2.0 + q] for q in range(world)]
            assert 'collectives go through gloo' in se
        outs = run(world, allow=False)
        assert all(rc != 0 for _, _, rc in outs)
        assert all('did not come up' in se for _, se, _ in outs)
