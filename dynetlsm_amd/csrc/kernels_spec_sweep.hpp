// Speculative-batch sweep (algo 2): the same Gauss-Seidel random-walk
// Metropolis scan as k_sweep_slice, restructured so the O(N) work of every MH
// step runs on the whole chip instead of one CU per time slice.
//
// Within a slice, step j depends on steps < j only through the positions they
// accepted.  For a batch of B consecutive nodes:
//
//   eval    (chip-wide, one workgroup per (slice, batch node k, part)):
//           full0_k = sum_{i != k} delta(x_k -> x_k' | X_i as of batch start)
//           H[k][m] = delta(x_k -> x_k' | X_m = x_m') - delta(.. | X_m = x_m)
//                     for m < k in the batch  (what node m's acceptance changes)
//   resolve (one wave per slice): for k = 0..B-1 in order
//           ratio_k = full0_k + prior_k + sum_{m<k accepted} H[k][m]
//           accept iff log u_k < ratio_k  ->  corr_{k'>k} += H[k'][k]
//
// which is algebraically the sequential scan (differences are rounding only).
// Proposals are a pure function of (seed, iter, t, j): every workgroup
// regenerates the ones it needs, nothing is communicated.
#pragma once
#include "chain.hpp"
#include "device_common.hpp"
#include "kernels_sweep.hpp"

namespace dlsm {

constexpr int SP_THREADS = 256;
constexpr int SP_BMAX = 128;

struct SpecBuf {
    double *full0;   // [nsl][B][parts]
    double *prop;    // [nsl][B][D + 2] : x1[D], logu, prior delta
    double *Ht;      // [nsl][B][B]     : Ht[m][k] = H[k][m], k > m
    int B, parts;
};

template <int D, int MODEL>
__global__ __launch_bounds__(SP_THREADS) void k_spec_eval(ChainView c, SpecBuf sb,
                                                          uint32_t iter, int parity,
                                                          int j0, int nb) {
    __shared__ double sx0[SP_BMAX * D];
    __shared__ double sx1[SP_BMAX * D];
    __shared__ double sRed[SP_THREADS / 64];
    const int tid = threadIdx.x;
    const int N = c.N, W = c.W;
    const int p = blockIdx.x % sb.parts;
    const int k = (blockIdx.x / sb.parts) % nb;
    const int s = blockIdx.x / (sb.parts * nb);
    const int t = 2 * s + parity;
    const double *Xt = c.X + (size_t)t * N * D;
    const int jk = j0 + k;
    // proposals of the batch nodes m <= k
    if (tid <= k) {
        const int j = j0 + tid;
        double x0[D], x1[D], logu;
#pragma unroll
        for (int d = 0; d < D; ++d) x0[d] = Xt[(size_t)j * D + d];
        make_proposal<D>(c, iter, t, j, x0, c.step[(size_t)t * N + j], x1, logu);
#pragma unroll
        for (int d = 0; d < D; ++d) { sx0[tid * D + d] = x0[d]; sx1[tid * D + d] = x1[d]; }
        if (tid == k && p == 0) {
            double *pr = sb.prop + ((size_t)s * sb.B + k) * (D + 2);
#pragma unroll
            for (int d = 0; d < D; ++d) pr[d] = x1[d];
            pr[D] = logu;
            pr[D + 1] = node_log_prior<D>(c, t, j, x1) - node_log_prior<D>(c, t, j, x0);
        }
    }
    __syncthreads();
    double xk0[D], xk1[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { xk0[d] = sx0[k * D + d]; xk1[d] = sx1[k * D + d]; }
    const uint32_t *yr = c.ybits + ((size_t)t * N + jk) * W;
    const uint32_t *yc = MODEL == DLSM_DIRECTED ? c.ytbits + ((size_t)t * N + jk) * W
                                                : nullptr;
    double E, bin = 0.0, bout = 0.0, irk = 0.0;
    if (MODEL == DLSM_UNDIRECTED) {
        E = exp(c.intercept[0]);
    } else {
        bin = c.intercept[0]; bout = c.intercept[1];
        E = exp(bin + bout);
        irk = 1.0 / c.radii[jk];
    }
    const int per = (N + sb.parts - 1) / sb.parts;
    const int lo = p * per, hi = min(N, lo + per);
    double acc = 0.0;
    for (int i = lo + tid; i < hi; i += SP_THREADS) {
        if (i == jk) continue;
        double xi[D];
#pragma unroll
        for (int d = 0; d < D; ++d) xi[d] = Xt[(size_t)i * D + d];
        const double d0 = dist_of<D>(xi, xk0, c.squared);
        const double d1 = dist_of<D>(xi, xk1, c.squared);
        if (MODEL == DLSM_UNDIRECTED) {
            acc += delta_undirected(d0, d1, bit_of(yr, i), E);
        } else {
            const double iri = 1.0 / c.radii[i];
            acc += delta_directed(d0, d1, bit_of(yr, i), bit_of(yc, i),
                                  bin * iri + bout * irk, bin * irk + bout * iri, E);
        }
    }
    const double total = block_sum_all<SP_THREADS / 64>(acc, sRed, tid);
    if (tid == 0) sb.full0[((size_t)s * sb.B + k) * sb.parts + p] = total;
    // effect of an earlier batch node's acceptance on this node's ratio
    if (p == 0 && tid < k) {
        const int m = tid, jm = j0 + m;
        const double a0 = dist_of<D>(&sx0[m * D], xk0, c.squared);
        const double a1 = dist_of<D>(&sx0[m * D], xk1, c.squared);
        const double b0 = dist_of<D>(&sx1[m * D], xk0, c.squared);
        const double b1 = dist_of<D>(&sx1[m * D], xk1, c.squared);
        double g0, g1;
        if (MODEL == DLSM_UNDIRECTED) {
            const int y = bit_of(yr, jm);
            g0 = delta_undirected(a0, a1, y, E);
            g1 = delta_undirected(b0, b1, y, E);
        } else {
            const double irm = 1.0 / c.radii[jm];
            const int y1 = bit_of(yr, jm), y2 = bit_of(yc, jm);
            const double aa = bin * irm + bout * irk, cc = bin * irk + bout * irm;
            g0 = delta_directed(a0, a1, y1, y2, aa, cc, E);
            g1 = delta_directed(b0, b1, y1, y2, aa, cc, E);
        }
        sb.Ht[((size_t)s * sb.B + m) * sb.B + k] = g1 - g0;
    }
}

template <int D>
__global__ __launch_bounds__(SP_THREADS) void k_spec_resolve(ChainView c, SpecBuf sb,
                                                             int parity, int j0, int nb) {
    extern __shared__ __attribute__((aligned(16))) double sH[];   // nb * nb
    const int tid = threadIdx.x;
    const int s = blockIdx.x;
    const int t = 2 * s + parity;
    const int N = c.N;
    const double *Ht = sb.Ht + (size_t)s * sb.B * sb.B;
    for (int q = tid; q < nb * nb; q += SP_THREADS) {
        const int m = q / nb, k = q % nb;
        sH[q] = k > m ? Ht[(size_t)m * sb.B + k] : 0.0;
    }
    __syncthreads();
    if (tid >= 64) return;
    const int lane = tid;
    constexpr int VPL = SP_BMAX / 64;
    double ratio0[VPL], corr[VPL], logu[VPL];
    int accf[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        const int k = lane + 64 * v;
        ratio0[v] = 0.0; corr[v] = 0.0; logu[v] = 0.0; accf[v] = 0;
        if (k < nb) {
            const double *f = sb.full0 + ((size_t)s * sb.B + k) * sb.parts;
            double tot = 0.0;
            for (int p = 0; p < sb.parts; ++p) tot += f[p];
            const double *pr = sb.prop + ((size_t)s * sb.B + k) * (D + 2);
            ratio0[v] = tot + pr[D + 1];
            logu[v] = pr[D];
        }
    }
    for (int k = 0; k < nb; ++k) {
        const int owner = k & 63, slot = k >> 6;
        double h[VPL];
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const int kk = lane + 64 * v;
            h[v] = (kk > k && kk < nb) ? sH[k * nb + kk] : 0.0;
        }
        double r = ratio0[0] + corr[0], lu = logu[0];
#pragma unroll
        for (int v = 1; v < VPL; ++v)
            if (slot == v) { r = ratio0[v] + corr[v]; lu = logu[v]; }
        int a = !(lu >= r);                       // metropolis.py:50
        a = __shfl(a, owner, 64);
        if (a) {
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                corr[v] += h[v];
                if (lane == owner && slot == v) accf[v] = 1;
            }
        }
    }
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        const int k = lane + 64 * v;
        if (k < nb) {
            const size_t tj = (size_t)t * N + j0 + k;
            if (accf[v]) {
                const double *pr = sb.prop + ((size_t)s * sb.B + k) * (D + 2);
#pragma unroll
                for (int d = 0; d < D; ++d) c.X[tj * D + d] = pr[d];
            }
            double st = c.step[tj];
            int32_t na = c.nacc[tj], ns = c.nsteps[tj], un = c.until[tj];
            metropolis_bookkeeping(st, na, ns, un, c.tune, c.tune_interval, accf[v]);
            c.step[tj] = st; c.nacc[tj] = na; c.nsteps[tj] = ns; c.until[tj] = un;
        }
    }
}

}  // namespace dlsm
