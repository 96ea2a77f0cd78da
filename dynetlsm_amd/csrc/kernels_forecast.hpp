// One-step-ahead forecasts: the O(n_samples N^2) accumulations behind
// DynamicNetworkHDPLPCM.forecast_probas / forecast_probas_pp_ (hdp_lpcm.py:555-626) and
// marginal_forecast (forecast.pyx:79-128).  64 x 64 node pairs per workgroup, 4 x 4 per
// thread, samples streamed through LDS; only tiles on or above the diagonal are
// computed and mirrored (the probabilities are symmetric).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_common.hpp"

namespace dlsm {

constexpr int FC_TILE = 64;
constexpr int FC_CHUNK = 16;      // samples per LDS refill

// out[i][j] = (1 / S) sum_s expit(b_s - |x_si - x_sj|),  Xs [S][N][D]
template <int D>
__global__ __launch_bounds__(256) void k_forecast_mean(const double *__restrict__ Xs,
                                                       const double *__restrict__ b, int S, int N,
                                                       int zero_diag, double *__restrict__ out) {
    const int ti0 = blockIdx.y, tj0 = blockIdx.x;
    if (tj0 < ti0) return;
    __shared__ double sXi[FC_CHUNK][FC_TILE * D];
    __shared__ double sXj[FC_CHUNK][FC_TILE * D];
    __shared__ double sB[FC_CHUNK];
    const int i0 = ti0 * FC_TILE, j0 = tj0 * FC_TILE;
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = 0.0;
    for (int s0 = 0; s0 < S; s0 += FC_CHUNK) {
        const int ns = min(FC_CHUNK, S - s0);
        for (int q = tid; q < ns * FC_TILE * D; q += 256) {
            const int s = q / (FC_TILE * D), r = q % (FC_TILE * D);
            const int gi = min(i0 * D + r, N * D - 1), gj = min(j0 * D + r, N * D - 1);
            sXi[s][r] = Xs[(size_t)(s0 + s) * N * D + gi];
            sXj[s][r] = Xs[(size_t)(s0 + s) * N * D + gj];
        }
        if (tid < ns) sB[tid] = b[s0 + tid];
        __syncthreads();
        for (int s = 0; s < ns; ++s) {
            const double bs = sB[s];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double d = dist_fast<D>(&sXi[s][(4 * ti + a) * D], &sXj[s][(4 * tj + c) * D], 0);
                    acc[a][c] += 1.0 / (1.0 + fast_exp(d - bs));
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = i0 + 4 * ti + a, j = j0 + 4 * tj + c;
            if (i < N && j < N) {
                const double v = (zero_diag && i == j) ? 0.0 : acc[a][c] / (double)S;
                out[(size_t)i * N + j] = v;
                out[(size_t)j * N + i] = v;
            }
        }
}

// marginal_forecast: out[i][j] = sum_s w_si w_sj expit(b_s - d_ij) / sum_s w_si w_sj,
// d_ij = |x_i - x_j| of the plug-in positions x [N][D]; W [S][N]; diagonal 0.
template <int D>
__global__ __launch_bounds__(256) void k_forecast_marginal(const double *__restrict__ x,
                                                           const double *__restrict__ W,
                                                           const double *__restrict__ b, int S,
                                                           int N, double *__restrict__ out) {
    const int ti0 = blockIdx.y, tj0 = blockIdx.x;
    if (tj0 < ti0) return;
    __shared__ double sWi[FC_CHUNK][FC_TILE];
    __shared__ double sWj[FC_CHUNK][FC_TILE];
    __shared__ double sB[FC_CHUNK];
    const int i0 = ti0 * FC_TILE, j0 = tj0 * FC_TILE;
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    double num[4][4], den[4][4], dd[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = min(i0 + 4 * ti + a, N - 1), j = min(j0 + 4 * tj + c, N - 1);
            dd[a][c] = dist_fast<D>(x + (size_t)i * D, x + (size_t)j * D, 0);
            num[a][c] = 0.0; den[a][c] = 0.0;
        }
    for (int s0 = 0; s0 < S; s0 += FC_CHUNK) {
        const int ns = min(FC_CHUNK, S - s0);
        for (int q = tid; q < ns * FC_TILE; q += 256) {
            const int s = q / FC_TILE, r = q % FC_TILE;
            sWi[s][r] = W[(size_t)(s0 + s) * N + min(i0 + r, N - 1)];
            sWj[s][r] = W[(size_t)(s0 + s) * N + min(j0 + r, N - 1)];
        }
        if (tid < ns) sB[tid] = b[s0 + tid];
        __syncthreads();
        for (int s = 0; s < ns; ++s) {
            const double bs = sB[s];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double w = sWi[s][4 * ti + a] * sWj[s][4 * tj + c];
                    num[a][c] += w / (1.0 + fast_exp(dd[a][c] - bs));
                    den[a][c] += w;
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = i0 + 4 * ti + a, j = j0 + 4 * tj + c;
            if (i < N && j < N) {
                // forecast.pyx accumulates w / n_iter in both sums: the ratio is the same
                const double v = i == j ? 0.0 : num[a][c] / den[a][c];
                out[(size_t)i * N + j] = v;
                out[(size_t)j * N + i] = v;
            }
        }
}

}  // namespace dlsm
