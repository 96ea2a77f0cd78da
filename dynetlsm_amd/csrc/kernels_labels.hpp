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
    const double lm = c.lmbda_p[0];
    double muk[D];
#pragma unroll
    for (int j = 0; j < D; ++j) muk[j] = lm * m[j] + (1 - lm) * xp[j];
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

// a12: sample_labels.py:134-190.  Dynamic LDS: the T transition matrices (rows padded
// to an odd length, shared by the workgroup; WLDS = false reads them from global memory
// when they do not fit) and per wave 2*T*K doubles (likelihood, partial marginal).  Sums that decide a draw run in the reference's index order so that the
// oracle (same Philox uniform) gets the same label.
constexpr int LAB_WAVES = 8;

__host__ __device__ inline int lab_row_pad(int K) { return K | 1; }

#ifdef DLSM_PIPE_TIMING
__device__ unsigned long long g_lab_t[4096][6];    // per wavefront: phase stamps (profiles/labels_phases.py)
#define DLSM_LAB_STAMP(I_, DEP_) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(DEP_)); lts[I_] = t_; }
#else
#define DLSM_LAB_STAMP(I_, DEP_)
#endif
template <int D, bool WLDS>
__global__ __launch_bounds__(64 * LAB_WAVES) void k_sample_labels(
    ChainView c, const double *__restrict__ w, uint32_t iter,
    int32_t *__restrict__ z_out) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int T = c.T, K = c.K, N = c.N;
    const int KP = WLDS ? lab_row_pad(K) : K;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * LAB_WAVES + wave;
#ifdef DLSM_PIPE_TIMING
    unsigned long long lts[6] = {0, 0, 0, 0, 0, 0};
#endif
    DLSM_LAB_STAMP(0, (double)lane)
    const double *wt = w;                     // row (t, j) at wt + (t K + j) KP
    double *tables = smem;
    if (WLDS) {
        for (int q = threadIdx.x; q < T * K * K; q += 64 * LAB_WAVES) {
            const int r = q / K;
            smem[(size_t)r * KP + (q - r * K)] = w[q];
        }
        wt = smem;
        tables = smem + (size_t)T * K * KP;
    }
    DLSM_LAB_STAMP(1, (double)lane)
    double *L = tables + (size_t)wave * 2 * T * K;
    double *pm = L + T * K;
    const bool live = i < N;                  // a whole wave is live or not
    if (live)                                 // the T x K table, 64 entries at a time
        for (int q = lane; q < T * K; q += 64) {
            const int t = q / K;
            L[q] = exp(gauss_loglik_tk<D>(c, t, i, q - t * K));
        }
    __syncthreads();
    if (!live) return;
    DLSM_LAB_STAMP(2, (double)lane)
    // the uniform of time t is drawn by lane t (T <= 64 per pass), all times at once
    double u_all = 0.0;
    // Lane k owns component k.  The sums over the components run in index order (as the
    // reference's do) with the k-th term fetched from lane k's register, not through LDS.
    // backward messages :164-170
    double bmk = 1.0;                         // message of component `lane` at time t
    for (int t = T - 1; t > 0; --t) {
        const double pmk = lane < K ? L[t * K + lane] * bmk : 0.0;
        if (lane < K) pm[t * K + lane] = pmk;          // the forward pass needs it again
        double s = 0.0;
        const double *wr = wt + ((size_t)t * K + min(lane, K - 1)) * KP;
        // the lane's row of w, eight LDS reads in flight per trip (one read per term of the
        // chain would pay the LDS latency K times per step); same terms, same order
        // (no test of k0 + u < K inside a trip: lanes >= K hold pmk = 0, the clamped read is a
        // finite weight, and adding their product, +0, changes nothing.  A step is 1.4 us of one
        // wavefront's chain - profiles/labels_phases.py; requesting the next step's row and table
        // entry a step ahead changed nothing: 38.7 against 39.3 us for the two label kernels)
        for (int k0 = 0; k0 < K; k0 += 8) {
            double wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = wr[min(k0 + u, K - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += wv[u] * lane_value(pmk, (k0 + u) & 63);
        }
        double tot = 0.0;
        for (int r0 = 0; r0 < K; r0 += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double v = lane_value(s, (r0 + u) & 63);
                tot += r0 + u < K ? v : 0.0;
            }
        }
        bmk = s / tot;
    }
    if (lane < K) pm[lane] = L[lane] * bmk;
    DLSM_LAB_STAMP(3, bmk)
    __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): LDS writes landed
    __builtin_amdgcn_wave_barrier();
    // forward sampling :173-188: the cumulative sum runs in index order and lane k keeps its
    // k-th value; the label is the number of cumulative values below u * total
    int zprev = 0;
    for (int t = 0; t < T; ++t) {
        const double *wrow = wt + (t == 0 ? (size_t)0 : ((size_t)t * K + zprev) * KP);
        if ((t & 63) == 0) {
            double u1;
            philox_uniform2(c.seed, (uint32_t)i, (uint32_t)(t + lane), iter,
                            stream_word(c.chain, STREAM_LABELS), u_all, u1);
        }
        const double u0 = lane_value(u_all, t & 63);
        const double term = lane < K ? wrow[lane] * pm[t * K + lane] : 0.0;
        double cdf = 0.0, mine = 0.0;
        for (int k0 = 0; k0 < K; k0 += 8) {            // lanes >= K hold term = 0: no test per k
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                cdf += lane_value(term, (k0 + u) & 63);
                mine = k0 + u == lane ? cdf : mine;
            }
        }
        const double u = u0 * cdf;
        const int zt = __popcll(__ballot(lane < K && u > mine));
        if (lane == 0) z_out[(size_t)t * N + i] = zt;
        zprev = zt;
    }
#ifdef DLSM_PIPE_TIMING
    DLSM_LAB_STAMP(4, (double)zprev)
    if (lane == 0 && i < 4096) for (int q = 0; q < 5; ++q) g_lab_t[i][q] = lts[q];
#endif
}

// The counts the conjugate updates need (sample_labels.py:176-188): n[0][0][k] initial labels,
// n[t][j][k] transitions j -> k into time t, nk[t][k] labels in use.  One workgroup per time
// step, histogram in LDS (same-address global atomics from 2000 wavefronts cost ~170 ns each).
// `trace_row` (may be NULL): the labels are also filed as bytes, [T][N], the device-resident
// HDP-LPCM loop's trace row of this sample.
__global__ __launch_bounds__(256) void k_label_counts(const int32_t *__restrict__ z, int N, int K,
                                                      int32_t *__restrict__ n_cnt,
                                                      int32_t *__restrict__ nk_cnt,
                                                      uint8_t *__restrict__ trace_row) {
    extern __shared__ int32_t hist[];         // K * K + K
    const int t = blockIdx.x;
    for (int q = threadIdx.x; q < K * K + K; q += 256) hist[q] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += 256) {
        const int zt = z[(size_t)t * N + i];
        const int zp = t == 0 ? 0 : z[(size_t)(t - 1) * N + i];
        atomicAdd(&hist[zp * K + zt], 1);
        atomicAdd(&hist[K * K + zt], 1);
        if (trace_row) trace_row[(size_t)t * N + i] = (uint8_t)zt;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < K * K; q += 256) n_cnt[(size_t)t * K * K + q] = hist[q];
    for (int q = threadIdx.x; q < K; q += 256) nk_cnt[t * K + q] = hist[K * K + q];
}

}  // namespace dlsm
