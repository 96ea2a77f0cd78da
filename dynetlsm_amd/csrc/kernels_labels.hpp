// Label block update: per node, backward messages
// over K states and T steps, then forward categorical sampling.
// One wave per node, lanes over the K mixture components (K <= 64); the T x K
// tables of a node live in that wave's slice of LDS.
#pragma once
#include "chain.hpp"
#include "device_common.hpp"

namespace dlsm {

// gaussian_likelihood_fast.pyx:17-27
template <int D>
__device__ __forceinline__ double spherical_normal_log_pdf(const double *x,
                                                           const double *mean,
                                                           double var) {
    double ss = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) ss += (x[k] - mean[k]) * (x[k] - mean[k]);
    ss *= 0.5 * (1. / var);
    return -0.5 * D * log(2 * 3.14159265358979323846 * var) - ss;
}

// log N(X[t, i]; m_tk, sigma_k) for component k (gaussian_likelihood_fast.pyx:44-49)
template <int D>
__device__ __forceinline__ double gauss_loglik_tk(const ChainView &c, int t, int i,
                                                  int k) {
    const double *x = c.X + ((size_t)t * c.N + i) * D;
    const double *m = c.mu + (size_t)k * D;
    if (t == 0) return spherical_normal_log_pdf<D>(x, m, c.sigma[k]);
    const double *xp = c.X + ((size_t)(t - 1) * c.N + i) * D;
    double muk[D];
#pragma unroll
    for (int j = 0; j < D; ++j) muk[j] = c.lmbda * m[j] + (1 - c.lmbda) * xp[j];
    return spherical_normal_log_pdf<D>(x, muk, c.sigma[k]);
}

// a11 seam: T x K table of one node (one wave)
template <int D>
__global__ __launch_bounds__(64) void k_gauss_table(ChainView c, int node,
                                                    int normalize,
                                                    double *__restrict__ out) {
    const int lane = threadIdx.x;
    for (int t = 0; t < c.T; ++t) {
        double v = lane < c.K ? gauss_loglik_tk<D>(c, t, node, lane) : -INFINITY;
        if (normalize) {
            double m = v;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
            v -= m;
        }
        if (lane < c.K) out[t * c.K + lane] = exp(v);
    }
}

// a12: sample_labels.py:134-190.  Dynamic LDS: per wave 3*T*K doubles
// (likelihood, backward message, partial marginal).  Sums that decide a draw
// run in the reference's index order so that the oracle (same Philox uniform)
// gets the same label.
constexpr int LAB_WAVES = 4;

template <int D>
__global__ __launch_bounds__(64 * LAB_WAVES) void k_sample_labels(
    ChainView c, const double *__restrict__ w, uint32_t iter,
    int32_t *__restrict__ z_out, int32_t *__restrict__ n_cnt,
    int32_t *__restrict__ nk_cnt) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int T = c.T, K = c.K, N = c.N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * LAB_WAVES + wave;
    double *L = smem + (size_t)wave * 3 * T * K;
    double *bm = L + T * K;
    double *pm = bm + T * K;
    if (i >= N) return;                       // whole wave leaves together
    for (int t = 0; t < T; ++t)
        if (lane < K) L[t * K + lane] = exp(gauss_loglik_tk<D>(c, t, i, lane));
    if (lane < K) bm[(T - 1) * K + lane] = 1.0;
    __builtin_amdgcn_wave_barrier();
    // backward messages :164-170
    for (int t = T - 1; t > 0; --t) {
        if (lane < K) pm[t * K + lane] = L[t * K + lane] * bm[t * K + lane];
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): LDS writes landed
        __builtin_amdgcn_wave_barrier();
        double s = 0.0;
        if (lane < K) {
            const double *wr = w + ((size_t)t * K + lane) * K;
            for (int k = 0; k < K; ++k) s += wr[k] * pm[t * K + k];
            bm[(t - 1) * K + lane] = s;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        double tot = 0.0;
        for (int r = 0; r < K; ++r) tot += bm[(t - 1) * K + r];
        __builtin_amdgcn_wave_barrier();
        if (lane < K) bm[(t - 1) * K + lane] = s / tot;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
    if (lane < K) pm[lane] = L[lane] * bm[lane];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // forward sampling :173-188 (every lane walks the cdf; lane 0 records)
    int zprev = 0;
    for (int t = 0; t < T; ++t) {
        const double *wrow = t == 0 ? w : w + ((size_t)t * K + zprev) * K;
        double u0, u1;
        philox_uniform2(c.seed, (uint32_t)i, (uint32_t)t, iter,
                        stream_word(c.chain, STREAM_LABELS), u0, u1);
        double total = 0.0;
        for (int k = 0; k < K; ++k) total += wrow[k] * pm[t * K + k];
        const double u = u0 * total;
        double cdf = 0.0;
        int zt = 0;
        for (int k = 0; k < K; ++k) {
            cdf += wrow[k] * pm[t * K + k];
            zt += (u > cdf);
        }
        if (lane == 0) {
            z_out[(size_t)t * N + i] = zt;
            if (t == 0) atomicAdd(&n_cnt[zt], 1);
            else atomicAdd(&n_cnt[((size_t)t * K + zprev) * K + zt], 1);
            atomicAdd(&nk_cnt[t * K + zt], 1);
        }
        zprev = zt;
    }
}

}  // namespace dlsm
