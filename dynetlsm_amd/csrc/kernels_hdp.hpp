// Label-wise sums of the HDP-LPCM conjugate updates: everything in
// hdp_lpcm.py:901-954 and :1188-1280 that is O(T N) - sums over the nodes carrying a
// label - stays on the device, where X and the labels already are; the host keeps the
// O(T K) draws (and their MT19937 order).  One workgroup per (label k, slice t) walks the
// slice in a fixed order: bitwise reproducible, no atomics.
#pragma once
#include "chain.hpp"
#include "device_common.hpp"

namespace dlsm {

enum : int {
    HDP_SUMS_MEAN = 0,      // out[t][k][d] = sum_i V_ti,  V_0 = X_0, V_t = X_t - (1 - lm) X_{t-1}
    HDP_SUMS_RESIDUAL = 1,  // out[t][k]    = sum_i |X_ti - (1 - lm) X_{t-1,i} - lm mu_k|^2 (t = 0: |X - mu_k|^2)
    HDP_SUMS_LAMBDA = 2,    // out[t][k][2] = sum_i (mu_k - X_{t-1,i}) . (X_ti - X_{t-1,i}) / sigma_k,
                            //                sum_i |mu_k - X_{t-1,i}|^2 / sigma_k          (t >= 1)
    HDP_SUMS_LOGP = 3       // out[t][k]    = sum_i log w[t, z_{t-1,i}, k] (t = 0: log w[0, 0, k])
                            //                - 0.5 log sigma_k - 0.5 |res|^2 / sigma_k
                            //                - (0.5 a + 1) log sigma_k - 0.5 b / sigma_k
};
constexpr int HDP_THREADS = 256;

struct HdpParams {
    const double *mu;      // [K][D]
    const double *sigma;   // [K]
    const double *w;       // [T][K][K]
    double lmbda, a, b;
    // device-resident loop: the blending coefficient / the inverse-gamma scale are read from
    // device memory (they were drawn by the kernels just before); NULL = the values above
    const double *lmbda_p = nullptr, *b_p = nullptr;
};

// the sums of workgroup (k, t) given cluster k's mean `mk`, variance `sk` and the scalars
template <int D, int STAGE>
__device__ __forceinline__ void hdp_label_sums_wg(const ChainView &c, int k, int t,
                                                  const double (&mk)[D], double sk,
                                                  double lm, double a_, double hb_,
                                                  const double *__restrict__ w,
                                                  double *__restrict__ out) {
    constexpr int NV = STAGE == HDP_SUMS_MEAN ? D : (STAGE == HDP_SUMS_LAMBDA ? 2 : 1);
    __shared__ double buf[NV][HDP_THREADS / 64];
    const int tid = threadIdx.x;
    const int N = c.N, K = c.K;
    const int32_t *zt = c.z + (size_t)t * N;
    const int32_t *zp = t > 0 ? c.z + (size_t)(t - 1) * N : nullptr;
    const double *Xt = c.X + (size_t)t * N * D;
    const double *Xp = t > 0 ? c.X + (size_t)(t - 1) * N * D : nullptr;
    const double lsk = STAGE == HDP_SUMS_LOGP ? log(sk) : 0.0;
    double acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.0;
    // Four nodes per thread and trip: their labels are requested together, then the positions of
    // the cluster's members together (a node per trip was two memory round trips per node, one
    // behind the other, eight nodes in a row at config 3); a thread still adds its nodes in
    // ascending order: the sums are the same numbers.
    constexpr int NU = 4;
    for (int i0 = tid; i0 < N; i0 += NU * HDP_THREADS) {
      int zz[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
          const int i = i0 + u * HDP_THREADS;
          zz[u] = i < N ? zt[i] : -1;
      }
      double xs[NU][D], xps[NU][D];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
          const int i = i0 + u * HDP_THREADS;
          if (zz[u] == k) {
#pragma unroll
              for (int d = 0; d < D; ++d) {
                  xs[u][d] = Xt[(size_t)i * D + d];
                  xps[u][d] = t > 0 ? Xp[(size_t)i * D + d] : 0.0;
              }
          }
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int i = i0 + u * HDP_THREADS;
        if (zz[u] != k) continue;
        double x[D], xp[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { x[d] = xs[u][d]; xp[d] = xps[u][d]; }
        if (STAGE == HDP_SUMS_MEAN) {
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] += t > 0 ? x[d] - (1 - lm) * xp[d] : x[d];
        } else if (STAGE == HDP_SUMS_RESIDUAL || STAGE == HDP_SUMS_LOGP) {
            double ss = 0.0;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const double r = t > 0 ? x[d] - (1 - lm) * xp[d] - lm * mk[d] : x[d] - mk[d];
                ss += r * r;
            }
            if (STAGE == HDP_SUMS_RESIDUAL) {
                acc[0] += ss;
            } else {
                const int zprev = t > 0 ? zp[i] : 0;
                acc[0] += log(w[((size_t)t * K + zprev) * K + k]) - 0.5 * lsk - 0.5 * ss / sk -
                          (0.5 * a_ + 1.0) * lsk - 0.5 * hb_ / sk;
            }
        } else {
            if (t > 0) {
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const double dm = mk[d] - xp[d];
                    a0 += dm * (x[d] - xp[d]);
                    a1 += dm * dm;
                }
                acc[0] += a0 / sk;
                acc[1] += a1 / sk;
            }
        }
      }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const double s = block_sum_all<HDP_THREADS / 64>(acc[v], buf[v], tid);
        if (tid == 0) out[((size_t)t * K + k) * NV + v] = s;
    }
}

template <int D, int STAGE>
__global__ __launch_bounds__(HDP_THREADS) void k_hdp_label_sums(ChainView c, HdpParams hp,
                                                                double *__restrict__ out) {
    const int k = blockIdx.x;
    const double lm = hp.lmbda_p ? hp.lmbda_p[0] : hp.lmbda;
    const double hb_ = hp.b_p ? hp.b_p[0] : hp.b;
    double mk[D], sk = 1.0;
#pragma unroll
    for (int d = 0; d < D; ++d) mk[d] = STAGE == HDP_SUMS_MEAN ? 0.0 : hp.mu[(size_t)k * D + d];
    if (STAGE == HDP_SUMS_LAMBDA || STAGE == HDP_SUMS_LOGP) sk = hp.sigma[k];
    hdp_label_sums_wg<D, STAGE>(c, k, (int)blockIdx.y, mk, sk, lm, hp.a, hb_, hp.w, out);
}

}  // namespace dlsm
