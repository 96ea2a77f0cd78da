"""dlsm_batch_*: several chains of one network through shared launches (csrc/kernels_batch.hpp).
Every chain's trace must be bit for bit what dlsm_lsm_run produces for it alone - the items are
the single-chain launch's items, only dealt to other wavefronts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    import dynetlsm_amd
    return dynetlsm_amd


def _net(T, N, seed, density=0.1):
    rng = np.random.RandomState(seed)
    Y = (rng.rand(T, N, N) < density).astype(np.float64)
    Y = np.triu(Y, 1)
    Y = Y + Y.transpose(0, 2, 1)
    return Y, 0.7 * rng.randn(T, N, 2)


def _chain(eng, Y, X, cid, n_total, tune=None, nip=0, algo=0, b=0.3):
    T, N = Y.shape[:2]
    c = eng.Chain(T, N, 2, 'undirected', seed=77, chain_id=cid)
    c.upload_network(Y); c.set_positions(X); c.set_intercepts([b])
    c.set_prior_random_walk(2.0, 0.1)
    c.set_samplers(eng.SamplerGrid(T, N, 0.15, tune=tune, tune_interval=2))
    c.lsm_configure([b], 2.0, step_size_intercept=0.1, tune=tune, tune_interval=2, n_iter_procrustes=nip,
                    sweep_algo=algo)
    c.trace_alloc(n_total)
    return c


@pytest.mark.parametrize('T,N,nc,tune,pref', [(4, 600, 3, None, -1), (3, 700, 2, 6, 0), (1, 520, 4, None, 0),
                                              (5, 1300, 8, 5, 2)])
def test_batch_is_bitwise_the_single_chain_runs(eng, T, N, nc, tune, pref):
    Y, X = _net(T, N, 3)
    n_it = 9
    alone = []
    for cid in range(nc):
        with _chain(eng, Y, X, cid, n_it + 1, tune) as c:
            c.lsm_run(1, 3, procrustes_ref=-1)
            c.lsm_run(4, n_it - 3, procrustes_ref=pref)
            g = c.get_samplers(eng.SamplerGrid(T, N, 0.15, tune=tune, tune_interval=2))
            alone.append((c.trace_read(0, n_it + 1), g.step_size.copy(), g.n_accepted.copy(),
                          c.lsm_get_config().i_step_size[0]))
    chains = [_chain(eng, Y, X, cid, n_it + 1, tune) for cid in range(nc)]
    with eng.ChainBatch(chains) as b:
        b.lsm_run(1, 3, procrustes_ref=-1)
        b.lsm_run(4, n_it - 3, procrustes_ref=pref)
        b.synchronize()
        merged, single = b.stats()
        assert merged == (3 - 2) + (n_it - 3 - 2) and single == 4, (merged, single)
        for c, (tr, step, nacc, istep) in zip(chains, alone):
            got = c.trace_read(0, n_it + 1)                  # a member chain's own calls keep working
            for a, w in zip(got, tr):
                np.testing.assert_array_equal(a, w)
            g = c.get_samplers(eng.SamplerGrid(T, N, 0.15, tune=tune, tune_interval=2))
            np.testing.assert_array_equal(g.step_size, step)
            np.testing.assert_array_equal(g.n_accepted, nacc)
            assert c.lsm_get_config().i_step_size[0] == istep
    assert not np.array_equal(alone[0][0][0][-1], alone[1][0][0][-1])       # the chain id keys the draws
    # released from the batch the chains run on by themselves
    chains[0].lsm_run(n_it, 1, procrustes_ref=-1)
    chains[0].synchronize()
    for c in chains:
        c.close()


def test_batch_at_config_2_size(eng):
    """T=10, N=2000: four chains x 6 iterations through shared launches == the single-chain runs"""
    from dynetlsm_amd.synthetic import synthetic_lsm_network
    net = synthetic_lsm_network(T=10, N=2000, D=2, density=0.03, seed=0)
    n_it = 6

    def make(cid, src=None):
        c = eng.Chain(10, 2000, 2, 'undirected', seed=5, chain_id=cid)
        if src is None:
            c.upload_network(net['Y'])
        else:
            n = src.network_packed_words()
            buf = np.zeros(n, dtype=np.uint32)
            src.get_network_packed(buf.ctypes.data, n)
            c.set_network_packed(buf.ctypes.data, n)
        c.set_positions(net['X_init']); c.set_intercepts([net['intercept']])
        c.set_prior_random_walk(2.0, 0.1)
        c.set_samplers(eng.SamplerGrid(10, 2000, 0.1, tune=None))
        c.lsm_configure([net['intercept']], 2.0, tune=None, n_iter_procrustes=0)
        c.trace_alloc(n_it + 1)
        return c
    first = make(0)
    chains = [first] + [make(cid, first) for cid in range(1, 4)]
    for c in chains:
        c.lsm_run(1, n_it, procrustes_ref=0)
    alone = [c.trace_read(0, n_it + 1) for c in chains]
    for c in chains:
        c.close()
    first = make(0)
    chains = [first] + [make(cid, first) for cid in range(1, 4)]
    with eng.ChainBatch(chains) as b:
        b.lsm_run(1, n_it, procrustes_ref=0)
        b.synchronize()
        assert b.stats() == (n_it - 2, 2)
    for c, ref in zip(chains, alone):
        for a, w in zip(c.trace_read(0, n_it + 1), ref):
            np.testing.assert_array_equal(a, w)
        c.close()


def test_batch_refuses_what_it_cannot_share(eng):
    Y, X = _net(3, 520, 1)
    Y2, _ = _net(3, 520, 2)
    a = _chain(eng, Y, X, 0, 4)
    c = _chain(eng, Y2, X, 1, 4)
    with pytest.raises(eng.EngineError, match='different network'):
        eng.ChainBatch([a, c])
    with pytest.raises(eng.EngineError, match='listed twice'):
        eng.ChainBatch([a, a])
    d = eng.Chain(3, 520, 2, 'directed')
    with pytest.raises(eng.EngineError):
        eng.ChainBatch([a, d])
    e = _chain(eng, Y, X, 2, 4)
    with eng.ChainBatch([a, e]) as b:
        with pytest.raises(eng.EngineError, match='already belongs'):
            eng.ChainBatch([a])
        with pytest.raises(eng.EngineError, match='out of a chain'):
            b.lsm_run(1, 10)
        # a chain destroyed while in the batch: the batch notices, nothing dangles
        e.close()
        with pytest.raises(eng.EngineError, match='destroyed'):
            b.lsm_run(1, 2)
    a.lsm_run(1, 2)
    a.synchronize()
    for ch in (a, c, d):
        ch.close()


def test_batch_with_chains_configured_differently_runs_them_one_by_one(eng):
    """different sweep algorithms in one batch: nothing is merged, every trace is still the
    single-chain one"""
    Y, X = _net(3, 560, 4)
    ref = []
    for cid, algo in ((0, 4), (1, 2)):
        with _chain(eng, Y, X, cid, 6, algo=algo) as c:
            c.lsm_run(1, 5)
            ref.append(c.trace_read(0, 6))
    chains = [_chain(eng, Y, X, 0, 6, algo=4), _chain(eng, Y, X, 1, 6, algo=2)]
    with eng.ChainBatch(chains) as b:
        b.lsm_run(1, 5)
        assert b.stats() == (0, 5)
    for c, r in zip(chains, ref):
        for a, w in zip(c.trace_read(0, 6), r):
            np.testing.assert_array_equal(a, w)
        c.close()
