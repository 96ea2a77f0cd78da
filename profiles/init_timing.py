"""Initialisation pipeline at the headline size: wall-clock of
every device stage, and of the host libraries the reference calls for the same
stage (scipy.sparse.csgraph.shortest_path, sklearn.manifold.MDS, scipy.linalg.eigh
on the explicit matrix, numpy gradient) on a bounded sample (one slice each).

    python profiles/init_timing.py [--T 10] [--N 2000] [--no-host]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import dynetlsm_amd as da                                  # noqa: E402
from dynetlsm_amd import initialization as init_mod        # noqa: E402
from dynetlsm_amd.synthetic import synthetic_lsm_network   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--T', type=int, default=10)
    ap.add_argument('--N', type=int, default=2000)
    ap.add_argument('--density', type=float, default=0.03)
    ap.add_argument('--no-host', action='store_true')
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'init_timing.json'))
    a = ap.parse_args()
    T, N, D = a.T, a.N, 2
    Y = synthetic_lsm_network(T, N, D, density=a.density, seed=0)['Y']
    res = dict(config='T=%d N=%d D=%d density=%.3f' % (T, N, D, a.density))

    c = da.Chain(T, N, D, 'undirected', seed=1)
    t0 = time.perf_counter(); c.upload_network(Y); res['upload_s'] = time.perf_counter() - t0
    c.init_shortest_paths(); c.init_release()            # warm-up (code objects, allocator)
    t0 = time.perf_counter(); c.init_shortest_paths(); res['hops_s'] = time.perf_counter() - t0
    rng = np.random.RandomState(0)
    X0 = np.stack([rng.uniform(size=N * D).reshape(N, D) for _ in range(4)])
    t0 = time.perf_counter()
    Xs, stress, n_iter = c.init_smacof(0, X0)
    res['smacof_s'] = time.perf_counter() - t0
    res['smacof_n_iter'] = [int(v) for v in n_iter]
    X = np.empty((T, N, D))
    X[0] = Xs[int(np.argmin(stress))]
    t0 = time.perf_counter()
    nl = []
    for t in range(1, T):
        X[t], ev, info = c.init_gmds_step(t, X[t - 1])
        nl.append(info['n_lanczos'])
    res['gmds_s'] = time.perf_counter() - t0
    res['gmds_n_lanczos'] = nl
    D1 = c.init_get_dissimilarity(1) if T > 1 else None
    D0 = c.init_get_dissimilarity(0)
    n_eval = [0]
    orig = c.init_mle_sums

    def counted(p0, p1):
        n_eval[0] += 1
        return orig(p0, p1)
    c.init_mle_sums = counted
    t0 = time.perf_counter()
    scale, b = init_mod.scale_intercept_mle(c, X)
    res['mle_s'] = time.perf_counter() - t0
    res['mle_evaluations'] = n_eval[0]
    res['mle'] = [float(scale), float(b)]
    res['device_total_s'] = (res['upload_s'] + res['hops_s'] + res['smacof_s'] +
                             res['gmds_s'] + res['mle_s'])
    # whole pipeline once more through the public function
    t0 = time.perf_counter()
    Xd = init_mod.generalized_mds(c, random_state=np.random.RandomState(0))
    res['generalized_mds_s'] = time.perf_counter() - t0
    res['generalized_mds_repeatable'] = bool(np.array_equal(Xd, X))

    if not a.no_host:
        from scipy.sparse import csgraph
        from scipy.linalg import eigh, orthogonal_procrustes
        from sklearn.manifold import MDS
        t0 = time.perf_counter()
        Dh = csgraph.shortest_path(Y[0], directed=False, unweighted=True)
        inf = np.isinf(Dh); Dh[inf] = Dh[~inf].max() + 1
        res['host_shortest_path_s_per_slice'] = time.perf_counter() - t0
        res['hops_equal_host'] = bool(np.array_equal(Dh, D0))
        t0 = time.perf_counter()
        Xh0 = MDS(dissimilarity='precomputed', n_components=D, n_init=4,
                  random_state=np.random.RandomState(0)).fit_transform(Dh)
        res['host_mds_s'] = time.perf_counter() - t0
        res['smacof_max_abs_diff_vs_sklearn'] = float(np.abs(Xh0 - X[0]).max())
        res['smacof_scale'] = float(np.abs(Xh0).max())
        if T > 1:
            t0 = time.perf_counter()
            H = np.eye(N) - np.ones((N, N)) / N
            G = (1 / 11.) * H.dot((-0.5 * D1 ** 2).dot(H)) + (10 / 11.) * X[0].dot(X[0].T)
            evals, evecs = eigh(G)
            Xh1 = evecs[:, ::-1][:, :D] * np.sqrt(evals[::-1][:D])
            R, _ = orthogonal_procrustes(Xh1, X[0])
            Xh1 = Xh1.dot(R)
            res['host_gmds_s_per_slice'] = time.perf_counter() - t0
            res['gmds_max_abs_diff_vs_eigh'] = float(np.abs(Xh1 - X[1]).max())
        res['host_total_s_extrapolated'] = (
            T * res['host_shortest_path_s_per_slice'] + res['host_mds_s'] +
            (T - 1) * res.get('host_gmds_s_per_slice', 0.0))
    c.close()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
