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
    # stand-in for a chain: a summary that depends on the chain id
    summ = np.array([g.chain_id, Y.sum() + g.chain_id, 0.5])
    allsum = g.gather_arrays(summ)
    assert len(allsum) == 2
    for r, s in enumerate(allsum):
        assert s[0] == r and s[1] == ref.sum() + r
    assert g.max_over_ranks(10.0 + g.rank) == 11.0
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
