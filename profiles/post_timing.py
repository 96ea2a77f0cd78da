"""Post-loop processing at config C3's size: posterior co-occurrence
matrices and the expected-VI criterion of every kept sample on the device, against the
reference's numpy formulation timed on a bounded sample (a few samples / one slice) and
extrapolated.

    python profiles/post_timing.py [--S 2500]
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
from dynetlsm_amd import posterior as post                 # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--T', type=int, default=10)
    ap.add_argument('--N', type=int, default=2000)
    ap.add_argument('--K', type=int, default=20)
    ap.add_argument('--S', type=int, default=2500)
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'post_timing.json'))
    a = ap.parse_args()
    T, N, K, S = a.T, a.N, a.K, a.S
    rng = np.random.RandomState(0)
    base = rng.randint(0, 6, size=(T, N))
    zs = np.repeat(base[None], S, axis=0)
    flip = rng.rand(S, T, N) < 0.1
    zs[flip] = rng.randint(0, K, size=int(flip.sum()))
    zs = zs.astype(np.int64)
    res = dict(config='T=%d N=%d K=%d kept samples=%d' % (T, N, K, S))
    c = da.Chain(T, N, 2, 'undirected')
    c.post_cooccurrence(zs[:64], K, want_matrix=False); c.post_release()      # warm-up
    t0 = time.perf_counter()
    cooc = c.post_cooccurrence(zs, K)
    res['device_cooccurrence_s'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    sums = c.post_expected_vi_sums()
    res['device_vi_sums_s'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    vis = post.expected_vi(zs, cooc.sum(axis=2), sums)
    res['host_vi_assembly_s'] = time.perf_counter() - t0
    res['engine_total_s'] = (res['device_cooccurrence_s'] + res['device_vi_sums_s'] +
                             res['host_vi_assembly_s'])
    # the reference's formulation (label_utils.py:40-62, posterior_vi.py:23-43) on a sample
    eye = np.eye(K)
    t0 = time.perf_counter()
    acc = np.zeros((N, N))
    for z in zs[:8, 0]:
        ind = eye[z]
        acc += ind.dot(ind.T)
    per = (time.perf_counter() - t0) / 8
    res['host_cooccurrence_s_extrapolated'] = per * S * T

    def ref_vi(labels, cp):
        n, ng = labels.shape[0], labels.max() + 1
        resp = np.zeros((n, ng)); resp[np.arange(n), labels] = 1
        nk = resp.sum(axis=0); nz = nk != 0
        vi = np.sum(nk[nz] * np.log2(nk[nz]))
        vi -= 2 * np.log2(np.sum(cp * resp[:, labels].T, axis=1)).sum()
        vi += np.log2(np.sum(cp, axis=1)).sum()
        return vi / n
    t0 = time.perf_counter()
    v0 = np.mean([ref_vi(zs[3, t], cooc[t]) for t in range(T)])
    res['host_vi_s_extrapolated'] = (time.perf_counter() - t0) * S
    res['vi_rel_diff_vs_reference_formula'] = float(abs(v0 - vis[3]) / abs(v0))
    res['host_total_s_extrapolated'] = (res['host_cooccurrence_s_extrapolated'] +
                                        res['host_vi_s_extrapolated'])
    c.close()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, 'w'), indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
