// Post-loop processing of the label trace: posterior co-occurrence
// matrices (label_utils.py:40-62) and the sample-dependent sum of the expected-VI
// criterion (model_selection/posterior_vi.py:23-52).  Both are O(n_samples T N^2); the
// labels are kept on the device as bytes, sample-minor: zt[t][i][s], rows padded to 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.hpp"

namespace dlsm {

// int64 (the caller's zs_) or uint8 (the device-resident trace) [S][T][N] (a chunk of samples
// s0 .. s0 + ns) -> uint8 [T][N][Spad]
template <typename ZT>
__global__ __launch_bounds__(256) void k_post_pack_labels(const ZT *__restrict__ zs, int ns,
                                                          int s0, int T, int N, int Spad,
                                                          uint8_t *__restrict__ zt) {
    const size_t total = (size_t)ns * T * N;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (size_t)gridDim.x * 256) {
        const int i = (int)(q % N);
        const int t = (int)((q / N) % T);
        const int s = (int)(q / ((size_t)N * T));
        zt[((size_t)t * N + i) * Spad + s0 + s] = (uint8_t)zs[q];
    }
}

// counts[t][i][j] = #{s : z_sti == z_stj}.  64 x 64 pairs per workgroup, 4 x 4 per thread;
// the labels of 64 samples at a time go through LDS as 32-bit words of 4 samples.
constexpr int PC_TILE = 64;
__global__ __launch_bounds__(256) void k_post_cooccurrence(const uint8_t *__restrict__ zt, int N,
                                                           int S, int Spad,
                                                           uint32_t *__restrict__ counts) {
    __shared__ uint32_t sI[PC_TILE][17];        // 64 rows x 16 words (+1: bank spread)
    __shared__ uint32_t sJ[PC_TILE][17];
    const int t = blockIdx.z;
    const int i0 = blockIdx.y * PC_TILE, j0 = blockIdx.x * PC_TILE;
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    const uint8_t *base = zt + (size_t)t * N * Spad;
    uint32_t cnt[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) cnt[a][b] = 0u;
    for (int s0 = 0; s0 < S; s0 += 64) {
        for (int q = tid; q < PC_TILE * 16; q += 256) {
            const int r = q >> 4, w = q & 15;
            const int gi = min(i0 + r, N - 1), gj = min(j0 + r, N - 1);
            sI[r][w] = *(const uint32_t *)(base + (size_t)gi * Spad + s0 + 4 * w);
            sJ[r][w] = *(const uint32_t *)(base + (size_t)gj * Spad + s0 + 4 * w);
        }
        __syncthreads();
        const int nw = min(16, (S - s0 + 3) / 4);
        for (int w = 0; w < nw; ++w) {
            const int nb = min(4, S - s0 - 4 * w);       // valid samples in this word
            uint32_t wi[4], wj[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { wi[a] = sI[4 * ti + a][w]; wj[a] = sJ[4 * tj + a][w]; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint32_t x = wi[a] ^ wj[b];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        cnt[a][b] += (k < nb && ((x >> (8 * k)) & 0xFFu) == 0u) ? 1u : 0u;
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = i0 + 4 * ti + a, j = j0 + 4 * tj + b;
            if (i < N && j < N) counts[((size_t)t * N + i) * N + j] = cnt[a][b];
        }
}

__global__ __launch_bounds__(256) void k_post_counts_to_proba(const uint32_t *__restrict__ counts,
                                                              size_t n, double n_samples,
                                                              double *__restrict__ out) {
    // a division, as label_utils.py:62 (count / n_iter), not a multiplication by 1 / n
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (size_t)gridDim.x * 256)
        out[q] = (double)counts[q] / n_samples;
}

// part[t][g][s] = sum over the 16 rows i of group g of log2( sum_j C_t[i][j] [z_stj == z_sti] )
// lanes = samples; one wavefront carries 4 rows at a time (the byte of node j is loaded once
// for the 4 rows; C_t[i][j] is wave-uniform).
constexpr int PV_ROWS_PER_WAVE = 4;
constexpr int PV_ROWS_PER_WG = 16;
__global__ __launch_bounds__(256) void k_post_vi_rows(const uint8_t *__restrict__ zt,
                                                      const double *__restrict__ cooc, int N, int S,
                                                      int Spad, double *__restrict__ part) {
    const int t = blockIdx.z, chunk = blockIdx.y, g = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = chunk * 64 + lane;
    const uint8_t *zb = zt + (size_t)t * N * Spad + (size_t)chunk * 64 + lane;
    const double *C = cooc + (size_t)t * N * N;
    const int ibase = g * PV_ROWS_PER_WG + wave * PV_ROWS_PER_WAVE;
    int zi[PV_ROWS_PER_WAVE];
    const double *Crow[PV_ROWS_PER_WAVE];
    double acc[PV_ROWS_PER_WAVE];
#pragma unroll
    for (int r = 0; r < PV_ROWS_PER_WAVE; ++r) {
        const int i = min(ibase + r, N - 1);
        zi[r] = zb[(size_t)i * Spad];
        Crow[r] = C + (size_t)i * N;
        acc[r] = 0.0;
    }
#pragma unroll 4
    for (int j = 0; j < N; ++j) {
        const int zj = zb[(size_t)j * Spad];
#pragma unroll
        for (int r = 0; r < PV_ROWS_PER_WAVE; ++r) acc[r] += zj == zi[r] ? Crow[r][j] : 0.0;
    }
    double tot = 0.0;
#pragma unroll
    for (int r = 0; r < PV_ROWS_PER_WAVE; ++r)
        if (ibase + r < N) tot += log2(acc[r]);
    __shared__ double sW[4][64];
    sW[wave][lane] = tot;
    __syncthreads();
    if (wave == 0 && s < S) {
        const double v = sW[0][lane] + sW[1][lane] + sW[2][lane] + sW[3][lane];
        part[((size_t)t * gridDim.x + g) * Spad + s] = v;
    }
}

// out[t][s] = sum_g part[t][g][s] (fixed order)
__global__ __launch_bounds__(256) void k_post_vi_reduce(const double *__restrict__ part, int ngroups,
                                                        int S, int Spad, double *__restrict__ out) {
    const int t = blockIdx.y;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    double v = 0.0;
    for (int g = 0; g < ngroups; ++g) v += part[((size_t)t * ngroups + g) * Spad + s];
    out[(size_t)t * S + s] = v;
}

// ---- post-loop processing on the device-resident trace (hdp_lpcm.py:1085-1162) ------------------
// nk[s][t][k] = nodes carrying label k at time t of stored sample s (approx_bic.py:26-51,
// posterior_vi.py:31-36 and label_utils.py:73-81 all start from these counts)
__global__ __launch_bounds__(256) void k_post_label_counts(const uint8_t *__restrict__ z, int N, int K,
                                                           int32_t *__restrict__ nk) {
    __shared__ int hist[256];
    const int t = blockIdx.x, s = blockIdx.y, T = gridDim.x;
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint8_t *row = z + ((size_t)s * T + t) * N;
    for (int i = threadIdx.x; i < N; i += 256) atomicAdd(&hist[row[i]], 1);      // integers: order free
    __syncthreads();
    if ((int)threadIdx.x < K) nk[((size_t)s * T + t) * K + threadIdx.x] = hist[threadIdx.x];
}

// out[r] = sum_j cooc[r][j], one wavefront per row, fixed order
__global__ __launch_bounds__(256) void k_post_row_sums(const double *__restrict__ cooc, size_t rows,
                                                       int N, double *__restrict__ out) {
    const size_t r = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    double v = 0.0;
    for (int j = lane; j < N; j += 64) v += cooc[r * N + j];
    v = wave_sum_all(v);
    if (lane == 0) out[r] = v;
}

// Procrustes alignment of every stored sample onto a reference configuration
// (hdp_lpcm.py:1141-1146 -> procrustes.py:20-35 -> scipy.linalg.orthogonal_procrustes):
// M = X_s^T X_ref over the T N rows, R = U V^T of its SVD, X_s <- X_s R, mu_s <- mu_s R.
// One workgroup per sample: pass 1 reduces M in a fixed order, every thread then holds R (the
// d x d one-sided Jacobi of the sweep's own Procrustes step), pass 2 rotates the rows.
template <int D>
__device__ void jacobi_polar(const double (&M)[D][D], double (&R)[D][D]);   // kernels_sweep.hpp
template <int D>
__global__ __launch_bounds__(256) void k_post_align(double *__restrict__ Xs, double *__restrict__ mus,
                                                    const double *__restrict__ ref, int rows, int K) {
    __shared__ double sM[4][D * D];
    __shared__ double sR[D * D];
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *X = Xs + (size_t)s * rows * D;
    double m[D][D];
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) m[a][b] = 0.0;
    for (int r = tid; r < rows; r += 256) {
        double x[D], y[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { x[d] = X[(size_t)r * D + d]; y[d] = ref[(size_t)r * D + d]; }
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) m[a][b] = fma(x[a], y[b], m[a][b]);
    }
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) {
            const double v = wave_sum_all(m[a][b]);
            if (lane == 0) sM[wave][a * D + b] = v;
        }
    __syncthreads();
    if (tid == 0) {
        double M[D][D], R[D][D];
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b)
                M[a][b] = (sM[0][a * D + b] + sM[1][a * D + b]) + (sM[2][a * D + b] + sM[3][a * D + b]);
        jacobi_polar<D>(M, R);
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) sR[a * D + b] = R[a][b];
    }
    __syncthreads();
    double R[D][D];
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) R[a][b] = sR[a * D + b];
    auto rotate = [&](double *row) {
        double x[D], y[D];
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = row[d];
#pragma unroll
        for (int b = 0; b < D; ++b) {
            double acc = 0.0;
#pragma unroll
            for (int a = 0; a < D; ++a) acc = fma(x[a], R[a][b], acc);
            y[b] = acc;
        }
#pragma unroll
        for (int d = 0; d < D; ++d) row[d] = y[d];
    };
    for (int r = tid; r < rows; r += 256) rotate(X + (size_t)r * D);
    if (mus != nullptr && tid < K) rotate(mus + ((size_t)s * K + tid) * D);
}

// partial[c][e] = sum over the samples s = c, c + C, .. of Xs[s][e]; then mean[e] = sum_c / S
constexpr int PM_CHUNKS = 64;
__global__ __launch_bounds__(256) void k_post_mean_partial(const double *__restrict__ Xs, size_t len,
                                                           int S, double *__restrict__ partial) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (e >= len) return;
    double v = 0.0;
    for (int s = c; s < S; s += PM_CHUNKS) v += Xs[(size_t)s * len + e];
    partial[(size_t)c * len + e] = v;
}
__global__ __launch_bounds__(256) void k_post_mean_final(const double *__restrict__ partial, size_t len,
                                                         int S, double *__restrict__ out) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= len) return;
    double v = 0.0;
    for (int c = 0; c < PM_CHUNKS; ++c) v += partial[(size_t)c * len + e];
    out[e] = v / (double)S;
}

// latent_marginal_loglikelihood (model_selection/approx_bic.py:54-76): the forward algorithm over
// the label chain of every node at positions X, one wavefront per node, lane k = component k
// (K <= 64); spherical_normal_log_pdf of gaussian_likelihood_fast.pyx:17-27 for the densities.
// part[workgroup] = sum over its 4 nodes of sum_t log c_t.
template <int D>
__global__ __launch_bounds__(256) void k_post_forward_loglik(const double *__restrict__ X, int T, int N,
                                                             const double *__restrict__ init_w,
                                                             const double *__restrict__ trans_w,
                                                             const double *__restrict__ mu,
                                                             const double *__restrict__ sigma,
                                                             double lmbda, int K,
                                                             double *__restrict__ part) {
    __shared__ double sW[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave;
    double ll = 0.0;
    if (i < N) {
        const bool on = lane < K;
        const int kc = min(lane, K - 1);
        double m[D];
#pragma unroll
        for (int d = 0; d < D; ++d) m[d] = mu[(size_t)kc * D + d];
        const double var = sigma[kc];
        const double lognorm = -0.5 * D * log(2.0 * 3.14159265358979323846 * var);
        double f = 0.0, xprev[D];
        for (int t = 0; t < T; ++t) {
            double x[D], ss = 0.0;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                x[d] = X[((size_t)t * N + i) * D + d];
                const double mean = t == 0 ? m[d] : lmbda * m[d] + (1.0 - lmbda) * xprev[d];
                const double df = x[d] - mean;
                ss += df * df;
            }
            const double g = on ? exp(lognorm - 0.5 * ss / var) : 0.0;
            if (t == 0) {
                f = on ? init_w[kc] * g : 0.0;
            } else {
                double acc = 0.0;            // sum_j f_j trans_w[t][j][k], j ascending
                for (int j = 0; j < K; ++j)
                    acc = fma(lane_value(f, j), on ? trans_w[((size_t)t * K + j) * K + kc] : 0.0, acc);
                f = g * acc;
            }
            const double c = wave_sum_all(f);
            ll += log(c);
            f /= c;
#pragma unroll
            for (int d = 0; d < D; ++d) xprev[d] = x[d];
        }
    }
    if (lane == 0) sW[wave] = ll;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sW[0] + sW[1]) + (sW[2] + sW[3]);
}

}  // namespace dlsm
