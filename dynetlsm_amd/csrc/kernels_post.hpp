// Post-loop processing of the label trace: posterior co-occurrence
// matrices (label_utils.py:40-62) and the sample-dependent sum of the expected-VI
// criterion (model_selection/posterior_vi.py:23-52).  Both are O(n_samples T N^2); the
// labels are kept on the device as bytes, sample-minor: zt[t][i][s], rows padded to 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.hpp"

namespace dlsm {

// int64 [S][T][N] (a chunk of samples s0 .. s0 + ns) -> uint8 [T][N][Spad]
__global__ __launch_bounds__(256) void k_post_pack_labels(const int64_t *__restrict__ zs, int ns,
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

}  // namespace dlsm
