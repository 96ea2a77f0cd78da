"""SURVEY.md 8e on an MI355X: the packed-network hand-over between chains
(dlsm_get/set_network_packed) and `bench.py --gpus 2` launching its own ranks (both ranks on
cuda:0 with the gloo backend: the box has one GPU; on an 8-GPU node the same code runs one
rank per GPU over RCCL)."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch      # before the engine's library: one HIP runtime per process, torch's first

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def eng():
    import dynetlsm_amd
    return dynetlsm_amd


def _net(T, N, directed, seed):
    rng = np.random.RandomState(seed)
    Y = (rng.rand(T, N, N) < 0.15).astype(np.float64)
    for t in range(T):
        np.fill_diagonal(Y[t], 0.0)
    if not directed:
        Y = np.triu(Y, 1)
        Y = Y + Y.transpose(0, 2, 1)
    return Y, rng.randn(T, N, 2), rng.dirichlet(np.ones(N))


@pytest.mark.parametrize('model', ['undirected', 'directed'])
@pytest.mark.parametrize('N', [37, 300])
def test_packed_network_roundtrip(eng, model, N):
    T = 3
    Y, X, radii = _net(T, N, model == 'directed', 4)
    b = [0.3] if model == 'undirected' else [0.3, 0.6]
    with eng.Chain(T, N, 2, model) as a, eng.Chain(T, N, 2, model) as c:
        a.upload_network(Y)
        n = a.network_packed_words()
        W = ((N + 31) // 32 + 3) // 4 * 4
        assert n == T * N * W * (2 if model == 'directed' else 1)
        buf = np.zeros(n, dtype=np.uint32)
        a.get_network_packed(buf.ctypes.data, n)              # host pointer
        # the layout: bit i of row j of slice t is Y[t, j, i]
        rows = buf[:T * N * W].reshape(T, N, W)
        for (t, j, i) in [(0, 0, 1), (1, 5, N - 1), (2, N - 1, 0), (1, 7, 7)]:
            assert ((rows[t, j, i >> 5] >> (i & 31)) & 1) == int(Y[t, j, i])
        c.set_network_packed(buf.ctypes.data, n)
        for ch in (a, c):
            ch.set_positions(X); ch.set_intercepts(b)
            if model == 'directed':
                ch.set_radii(radii)
        assert a.loglik_full() == c.loglik_full()              # bit for bit
        # through a device buffer (what the RCCL broadcast hands over)
        dbuf = torch.zeros(n, dtype=torch.int32, device='cuda:0')
        a.get_network_packed(dbuf.data_ptr(), n)
        torch.cuda.synchronize()
        assert np.array_equal(dbuf.cpu().numpy().view(np.uint32), buf)
        with eng.Chain(T, N, 2, model) as d:
            d.set_network_packed(dbuf.data_ptr(), n)
            d.set_positions(X); d.set_intercepts(b)
            if model == 'directed':
                d.set_radii(radii)
            assert d.loglik_full() == a.loglik_full()
        # a buffer that violates the layout is refused: diagonal, padding, transpose
        for word, bit in ((0, 0), (W - 1, 31)):
            bad = buf.copy()
            bad[word] |= np.uint32(1 << bit)
            with pytest.raises(eng.EngineError) as e:
                c.set_network_packed(bad.ctypes.data, n)
            assert e.value.code == -4
        bad = buf.copy()
        bad[0] ^= np.uint32(2)                                  # Y[0, 0, 1] without Y[0, 1, 0]
        with pytest.raises(eng.EngineError) as e:
            c.set_network_packed(bad.ctypes.data, n)
        assert e.value.code == -4
        with pytest.raises(eng.EngineError):
            c.set_network_packed(buf.ctypes.data, n - 1)


def _run_bench(*extra):
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(extra), env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_bench_launches_two_ranks_by_itself():
    """`python bench.py --gpus 2` as typed: the parent spawns the ranks, rank 0 broadcasts the
    packed network, both chains run, the results are gathered; config 5's workload
    (HDP-LPCM chains) rides along as extra_configs"""
    line = _run_bench('--gpus', '2', '--backend', 'gloo', '--share-device0', '--steps', '5',
                      '--warmup', '2', '--profile-steps', '0', '--no-cpu', '--model', 'all')
    assert line['n_gpus'] == 2 and line['steps'] == 5 and line['scaling'] == 'weak'
    assert line['config']['chains'] == 2 and 'DynamicNetworkLSM' in line['config']['workload']
    assert line['value'] > 0 and line['dtype'] == 'f64'
    assert line['gathered']['X_mean'] == [2, 10, 2000, 2]
    assert line['gathered']['logps'] == [2, 5]
    summ = line['chain_summaries[intercept_mean,intercept_sd,logp_mean,logp_last]']
    assert len(summ) == 2 and summ[0] != summ[1]            # chain id keys the Philox streams
    assert abs(summ[0][2] - summ[1][2]) < 0.01 * abs(summ[0][2])    # same network, same posterior
    assert line['X_mean_rms_between_chains'] > 0
    hdp = line['extra_configs'][0]
    assert hdp['n_gpus'] == 2 and 'DynamicNetworkHDPLPCM' in hdp['config']['workload']
    assert hdp['value'] > 0 and hdp['gathered']['X_mean'] == [2, 10, 2000, 2]
    assert len(hdp['n_clusters_used_last']) == 2


def test_bench_single_gpu_line_has_the_contract_fields():
    line = _run_bench('--steps', '20', '--warmup', '5', '--profile-steps', '5', '--cpu-iters', '1',
                      '--model', 'lsm')
    assert line['n_gpus'] == 1 and line['metric'].startswith('Gibbs iterations/sec')
    r = line['roofline']
    assert r['bound'] == 'fp64_valu' and r['unit'] == 'TFLOP/s' and 0 < r['frac'] < 1
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert line['roofline_hbm']['bound'] == 'hbm' and line['roofline_hbm']['unit'] == 'GB/s'
    assert line['roofline_loglik']['kernel'].startswith('k_loglik_undirected')
    c = line['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] == 1 and c['value'] > 0
    assert c['engine_loglik_rel_err_vs_oracle'] < 1e-6          # north_star's tolerance
    assert 0 < line['iteration_fp64_valu']['frac'] < 1


def test_chains_sharing_a_gpu_are_bitwise_the_single_chain_runs(eng):
    """several chains per GPU (bench.py --chains-per-gpu): each chain is its own handle and
    stream, enqueued from its own host thread; what a chain computes does not depend on its
    neighbours - traces bitwise equal to the same chains run one after the other"""
    import threading
    from dynetlsm_amd.synthetic import synthetic_lsm_network
    T, N, n_it = 4, 700, 12
    net = synthetic_lsm_network(T, N, 2, density=0.05, seed=2)

    def make(cid):
        c = eng.Chain(T, N, 2, 'undirected', seed=99, chain_id=cid)
        c.upload_network(net['Y']); c.set_positions(net['X_init'])
        c.set_intercepts([net['intercept']])
        c.set_prior_random_walk(2.0, 0.1)
        c.set_samplers(eng.SamplerGrid(T, N, 0.1, tune=6, tune_interval=2))
        c.lsm_configure([net['intercept']], 2.0, tune=6, n_iter_procrustes=0)
        c.trace_alloc(n_it + 1)
        return c
    alone = []
    for cid in range(3):
        with make(cid) as c:
            c.lsm_run(1, n_it, procrustes_ref=0)
            alone.append(c.trace_read(0, n_it + 1))
    chains = [make(cid) for cid in range(3)]
    ths = [threading.Thread(target=c.lsm_run, args=(1, n_it, 0)) for c in chains]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for c, ref in zip(chains, alone):
        got = c.trace_read(0, n_it + 1)
        for a, b in zip(got, ref):
            np.testing.assert_array_equal(a, b)
        c.close()
    assert not np.array_equal(alone[0][0][-1], alone[1][0][-1])     # chain id keys the draws


def test_bench_through_rccl_in_a_group_of_one():
    """the RCCL code path on a single-GPU box: `--force-collectives` initialises the nccl
    backend with world size 1 and sends the packed network, the starting values, the timing
    reduction and the final gather through it (device tensors, stream ordering)"""
    common = ('--gpus', '1', '--model', 'lsm', '--steps', '200', '--warmup', '50',
              '--profile-steps', '0', '--no-cpu')
    line = _run_bench('--force-collectives', '--backend', 'nccl', *common)
    assert line['n_gpus'] == 1 and line['value'] > 0
    assert 'nccl' in line['config']['network_broadcast']
    co = line['collectives']            # what RCCL itself reported, and what went through it
    assert co['backend'] == 'nccl' and co['world_size'] == 1 and co['device'] == 'cuda:0'
    assert co['control_backend'] == 'gloo'
    assert co['network_broadcast_bytes'] >= 10 * 2000 * 64 * 4 and co['network_broadcast_ms'] > 0
    assert co['gather_calls'] >= 3 and co['gather_bytes_per_rank'] >= 10 * 2000 * 2 * 8
    assert line['gathered']['X_mean'] == [1, 10, 2000, 2]
    assert line['per_rank_value'] == [line['value']]
    # the collectives sit outside the timed region: the rate is the plain single-GPU run's
    plain = _run_bench(*common)
    assert plain['config']['network_broadcast'] == 'none'
    # (two separate processes on a shared pool: run-to-run spread is 1-2 %; the quantitative
    # evidence that the collectives are outside the window is `barrier_ms`, a fraction of the
    # 50 ms window)
    assert abs(line['value'] / plain['value'] - 1.0) < 0.10, (line['value'], plain['value'])
    assert line['barrier_ms'] < 5.0, line['barrier_ms']


def test_fit_chains_two_ranks_on_one_gpu_are_the_single_chain_fits(eng):
    """config 5 as ONE call (multichain.fit_chains, the estimator-level entry point): two ranks,
    both on cuda:0 over gloo (the box has one GPU; on an 8-GPU node the same call runs one rank per
    GPU over RCCL).  Every chain's trace is bit for bit what a single fit with that chain id and
    seed produces; the result carries the split R-hat over the chains."""
    from dynetlsm_amd.multichain import fit_chains
    rng = np.random.RandomState(3)
    T, N = 3, 150
    Y = (rng.rand(T, N, N) < 0.08).astype(np.float64)
    Y = np.triu(Y, 1); Y = Y + Y.transpose(0, 2, 1)
    kw = dict(n_iter=60, tune=20, burn=20, n_components=5, selection_type='map')
    res = fit_chains(eng.DynamicNetworkHDPLPCM(random_state=4, chain_id=0, **kw), Y, n_chains=2,
                     share_device0=True)
    assert res.n_chains == 2 and res.n_burn == 40
    assert res.traces['logps'].shape == (2, 100) and res.traces['lambdas'].shape == (2, 100, 1)
    assert res.X_mean.shape == (2, T, N, 2) and res.z.shape == (2, T, N)
    for r in range(2):
        m = eng.DynamicNetworkHDPLPCM(random_state=4 + r, chain_id=r, **kw).fit(Y)
        np.testing.assert_array_equal(res.traces['logps'][r], m.logps_)
        np.testing.assert_array_equal(res.traces['lambdas'][r], m.lambdas_)
        np.testing.assert_array_equal(res.z[r], m.z_)
        np.testing.assert_array_equal(res.X_mean[r], m.X_mean_)
        m.chain_.close()
    assert np.isfinite(res.rhat['logps']) and 'lambdas[0]' in res.rhat
    assert res.estimator.selected_id_ >= 40 and 'chain_' not in vars(res.estimator)
    # the LSM through the same call
    res = fit_chains(eng.DynamicNetworkLSM(n_iter=40, tune=10, burn=10, random_state=1), Y, n_chains=2,
                     share_device0=True)
    assert res.traces['logps'].shape == (2, 60) and res.z is None and res.X_mean.shape == (2, T, N, 2)
    assert not np.array_equal(res.traces['logps'][0], res.traces['logps'][1])


def test_bench_config5_eight_ranks_share_the_one_gpu():
    """BASELINE.json configs[4] - eight independent HDP-LPCM chains, one per rank - with the eight
    ranks of `bench.py --gpus 8` on the box's one GPU over gloo (one queue each): the launcher, the
    packed-network broadcast, eight chains with eight Philox chain ids, the gathers and the line's
    per-rank fields exactly as an 8-GPU node will produce them over RCCL (there the same command
    without --backend gloo --share-device0)"""
    line = _run_bench('--gpus', '8', '--backend', 'gloo', '--share-device0', '--steps', '10',
                      '--warmup', '2', '--profile-steps', '0', '--no-cpu', '--model', 'hdp')
    assert line['n_gpus'] == 8 and line['steps'] == 10 and line['scaling'] == 'weak'
    assert 'DynamicNetworkHDPLPCM' in line['config']['workload'] and line['config']['chains'] == 8
    assert line['config']['hdp_queues'] == 1           # ranks that share a device keep one queue each
    pr = line['per_rank_value']
    assert len(pr) == 8 and len(set(pr)) == 8 and min(pr) > 0
    assert abs(line['value'] - 8 * 10 / (line['ms_per_step'] * 10 / 1e3)) < 0.01 * line['value']
    assert line['value'] <= sum(pr) * 1.0001           # the slowest rank sets the aggregate
    summ = line['chain_summaries[intercept_mean,intercept_sd,logp_mean,logp_last]']
    assert len(summ) == 8 and len({tuple(x) for x in summ}) == 8
    # one network, one posterior: the intercepts agree (the HDP-LPCM's log-posterior itself jumps by 10^5
    # whenever a small extra cluster opens or closes - not a quantity to compare over 10 iterations)
    ic = np.array([x[0] for x in summ]); lp = np.array([x[2] for x in summ])
    assert np.isfinite(lp).all() and np.ptp(ic) < 0.02 * abs(ic.mean()), (ic, lp)
    assert line['X_mean_rms_between_chains'] > 0
    assert line['gathered']['X_mean'] == [8, 10, 2000, 2] and line['gathered']['logps'] == [8, 10]
    assert line['gathered']['lambdas'][:2] == [8, 10]
    assert len(line['n_clusters_used_last']) == 8
    co = line['collectives']
    assert co['world_size'] == 8 and co['backend'] == 'gloo' and co['rank'] == 0
    assert co['network_broadcast_bytes'] >= 10 * 2000 * 64 * 4 and co['network_broadcast_ms'] > 0
    assert co['gather_calls'] >= 4


def test_fit_chains_eight_ranks_on_one_gpu(eng):
    """multichain.fit_chains(..., n_chains=8): config 5 as one estimator-level call, eight ranks on
    cuda:0 over gloo; eight different chains of one posterior, split R-hat over them reported"""
    from dynetlsm_amd.multichain import fit_chains
    rng = np.random.RandomState(7)
    T, N = 3, 120
    Y = (rng.rand(T, N, N) < 0.1).astype(np.float64)
    Y = np.triu(Y, 1); Y = Y + Y.transpose(0, 2, 1)
    res = fit_chains(eng.DynamicNetworkHDPLPCM(n_iter=300, tune=100, burn=100, n_components=5,
                                               selection_type='map', random_state=2),
                     Y, n_chains=8, share_device0=True, timeout=900)
    assert res.n_chains == 8 and res.n_burn == 200
    assert res.traces['logps'].shape == (8, 500) and res.traces['lambdas'].shape == (8, 500, 1)
    assert res.X_mean.shape == (8, T, N, 2) and res.z.shape == (8, T, N)
    assert len({res.traces['logps'][r].tobytes() for r in range(8)}) == 8       # eight different chains
    for k in ('logps', 'intercepts[0]', 'lambdas[0]'):
        assert np.isfinite(res.rhat[k]), (k, res.rhat)
    assert res.rhat['logps'] < 1.5, res.rhat              # short chains of one small posterior
    assert 0 <= res.best_chain < 8 and len(res.summary()['seconds']) == 8
    print('fit_chains, 8 ranks on one GPU:', res.summary())
