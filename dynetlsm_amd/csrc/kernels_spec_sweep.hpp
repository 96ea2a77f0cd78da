// Speculative-batch sweep (algo 2): the same Gauss-Seidel random-walk
// Metropolis scan as k_sweep_slice, restructured so the O(N) work of every MH
// step runs on the whole chip instead of one CU per time slice.
//
// Within a slice, step j depends on steps < j only through the positions they
// accepted.  Per parity:
//
//   propose (once): x_j', log u_j and the prior delta of every (t, j) of the
//           parity.  Valid up front: X[t, j] and its step size change only at
//           step (t, j) itself, the neighbouring slices are of the other
//           parity, and the draws are a pure function of (seed, iter, t, j).
//   then for each batch of B consecutive nodes:
//   eval    (chip-wide, one workgroup per (slice, batch node k, part)):
//           full0_k = sum_{i != k} delta(x_k -> x_k' | X_i as of batch start)
//           H[k][m] = delta(x_k -> x_k' | X_m = x_m') - delta(.. | X_m = x_m)
//                     for m < k in the batch  (what node m's acceptance changes)
//   resolve (one wave per slice): walk the batch in order,
//           ratio_k = full0_k + prior_k + sum_{m<k accepted} H[k][m]
//           accept iff log u_k < ratio_k.  Lanes test their own node in
//           parallel; a ballot finds the first acceptance at or after the
//           cursor (everything before it is a final rejection), its row of H
//           is added to the later lanes, and the cursor jumps past it.
//
// which is algebraically the sequential scan (differences are rounding only).
#pragma once
#include "chain.hpp"
#include "device_common.hpp"
#include "kernels_sweep.hpp"

namespace dlsm {

constexpr int SP_THREADS = 256;
constexpr int SP_BMAX = 128;

struct SpecBuf {
    double *full0;   // [nsl][B][parts]
    double *prop;    // [nsl][N][D + 2] : x1[D], logu, prior delta
    double *Ht;      // [nsl][B][B]     : Ht[m][k] = H[k][m], k > m
    int B, parts;
};

template <int D>
__global__ __launch_bounds__(256) void k_spec_propose(ChainView c, SpecBuf sb,
                                                      uint32_t iter, int parity) {
    const int N = c.N;
    const int s = blockIdx.y;
    const int t = 2 * s + parity;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    double x0[D], x1[D], logu;
#pragma unroll
    for (int d = 0; d < D; ++d) x0[d] = c.X[((size_t)t * N + j) * D + d];
    make_proposal<D>(c, iter, t, j, x0, c.step[(size_t)t * N + j], x1, logu);
    double *pr = sb.prop + ((size_t)s * N + j) * (D + 2);
#pragma unroll
    for (int d = 0; d < D; ++d) pr[d] = x1[d];
    pr[D] = logu;
    pr[D + 1] = node_log_prior<D>(c, t, j, x1) - node_log_prior<D>(c, t, j, x0);
}

// Running products of (1 + E e^{-d}) for the current (P0) and the proposed (P1)
// position: sum_i log(p0_i / p1_i) = log(prod p0_i / prod p1_i); one log per
// `nflush` neighbours instead of one per neighbour.  nflush keeps the products
// far below the double range.
struct RatioAcc {
    double lin = 0.0, lg = 0.0, P0 = 1.0, P1 = 1.0;
    int cnt = 0;
    __device__ __forceinline__ void flush() {
        lg += log(P0 / P1);
        P0 = 1.0; P1 = 1.0; cnt = 0;
    }
    __device__ __forceinline__ double value() { flush(); return lin + lg; }
};

__device__ __forceinline__ int flush_interval(double E_max) {
    // (1 + E)^n < e^600
    const double l = log1p(E_max);
    if (!(l > 0.0)) return 1 << 20;
    const double n = 600.0 / l;
    return n < 1.0 ? 1 : (n > 1048576.0 ? 1 << 20 : (int)n);
}

template <int D, int MODEL>
__global__ __launch_bounds__(SP_THREADS) void k_spec_eval(ChainView c, SpecBuf sb,
                                                          int parity, int j0, int nb) {
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
    const bool do_h = (p == 0);
    // positions / proposals of the batch nodes m <= k (only k itself if no H)
    if (do_h ? (tid <= k) : (tid == k)) {
        const int j = j0 + tid;
        const double *pr = sb.prop + ((size_t)s * N + j) * (D + 2);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            sx0[tid * D + d] = Xt[(size_t)j * D + d];
            sx1[tid * D + d] = pr[d];
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
    int nflush;
    if (MODEL == DLSM_UNDIRECTED) {
        E = exp(c.intercept[0]);
        nflush = flush_interval(E);
    } else {
        bin = c.intercept[0]; bout = c.intercept[1];
        E = exp(bin + bout);
        irk = 1.0 / c.radii[jk];
        nflush = 0;                  // directed: exponents are not bounded by b
    }
    const int per = (N + sb.parts - 1) / sb.parts;
    const int lo = p * per, hi = min(N, lo + per);
    double acc = 0.0;
    RatioAcc ra;
    for (int i = lo + tid; i < hi; i += SP_THREADS) {
        if (i == jk) continue;
        double xi[D];
#pragma unroll
        for (int d = 0; d < D; ++d) xi[d] = Xt[(size_t)i * D + d];
        const double d0 = dist_of<D>(xi, xk0, c.squared);
        const double d1 = dist_of<D>(xi, xk1, c.squared);
        if (MODEL == DLSM_UNDIRECTED) {
            if (bit_of(yr, i)) ra.lin += d0 - d1;
            ra.P0 *= 1.0 + E * exp(-d0);
            ra.P1 *= 1.0 + E * exp(-d1);
            if (++ra.cnt >= nflush) ra.flush();
        } else {
            const double iri = 1.0 / c.radii[i];
            acc += delta_directed(d0, d1, bit_of(yr, i), bit_of(yc, i),
                                  bin * iri + bout * irk, bin * irk + bout * iri, E);
        }
    }
    if (MODEL == DLSM_UNDIRECTED) acc = ra.value();
    const double total = block_sum_all<SP_THREADS / 64>(acc, sRed, tid);
    if (tid == 0) sb.full0[((size_t)s * sb.B + k) * sb.parts + p] = total;
    // effect of an earlier batch node's acceptance on this node's ratio
    if (do_h && tid < k) {
        const int m = tid, jm = j0 + m;
        const double a0 = dist_of<D>(&sx0[m * D], xk0, c.squared);
        const double a1 = dist_of<D>(&sx0[m * D], xk1, c.squared);
        const double b0 = dist_of<D>(&sx1[m * D], xk0, c.squared);
        const double b1 = dist_of<D>(&sx1[m * D], xk1, c.squared);
        double h;
        if (MODEL == DLSM_UNDIRECTED) {
            // g1 - g0 with the four softplus terms under one log
            const double num = (1.0 + E * exp(-b0)) * (1.0 + E * exp(-a1));
            const double den = (1.0 + E * exp(-b1)) * (1.0 + E * exp(-a0));
            h = log(num / den);
            if (bit_of(yr, jm)) h += (b0 - b1) - (a0 - a1);
        } else {
            const double irm = 1.0 / c.radii[jm];
            const int y1 = bit_of(yr, jm), y2 = bit_of(yc, jm);
            const double aa = bin * irm + bout * irk, cc = bin * irk + bout * irm;
            h = delta_directed(b0, b1, y1, y2, aa, cc, E) -
                delta_directed(a0, a1, y1, y2, aa, cc, E);
        }
        sb.Ht[((size_t)s * sb.B + m) * sb.B + k] = h;
    }
}

template <int D>
__global__ __launch_bounds__(SP_THREADS) void k_spec_resolve(ChainView c, SpecBuf sb,
                                                             int parity, int j0, int nb) {
    extern __shared__ __attribute__((aligned(16))) double sH[];   // nb rows, stride B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.x;
    const int t = 2 * s + parity;
    const int N = c.N;
    const int ldh = sb.B;
    const double *Ht = sb.Ht + (size_t)s * sb.B * sb.B;
    // stage rows 0..nb-1 of H^T as one flat, fully coalesced copy with many loads
    // in flight per lane (entries at or below the diagonal are never read)
    {
        const int n2 = (nb * ldh) / 2;                 // double2 elements (B is even)
        const double2 *src = (const double2 *)Ht;
        double2 *dst = (double2 *)sH;
        for (int q0 = 0; q0 < n2; q0 += SP_THREADS * 8) {
            // unconditional loads (clamped index): a predicated load makes hipcc
            // branch around it and wait vmcnt(0) per element
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[min(q0 + u * SP_THREADS + tid, n2 - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) dst[min(q0 + u * SP_THREADS + tid, n2 - 1)] = v[u];
        }
    }
    static_assert(SP_BMAX == 128, "the scan below handles two 64-node halves");
    // lane l owns batch nodes l (half 0) and l + 64 (half 1)
    double r0 = 0.0, r1 = 0.0, lu0 = 0.0, lu1 = 0.0;
    const bool valid0 = lane < nb, valid1 = lane + 64 < nb;
    if (wave == 0) {
        const int p1 = sb.parts;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int k = min(lane + 64 * v, nb - 1);          // clamped: no branches
            const double *f = sb.full0 + ((size_t)s * sb.B + k) * sb.parts;
            double fp[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) fp[p] = f[min(p, p1 - 1)];
            double tot = 0.0;
#pragma unroll
            for (int p = 0; p < 8; ++p) tot += p < p1 ? fp[p] : 0.0;
            const double *pr = sb.prop + ((size_t)s * N + j0 + k) * (D + 2);
            const double rr = tot + pr[D + 1], ll = pr[D];
            if (v == 0) { r0 = rr; lu0 = ll; } else { r1 = rr; lu1 = ll; }
        }
    }
    __syncthreads();
    if (wave != 0) return;
    // accepted nodes of each half as wave-uniform bit masks
    unsigned long long acc0 = 0ull, acc1 = 0ull;
    {
        unsigned long long live = ~0ull;                       // lanes at / after the cursor
        const bool two = nb > 64;                              // wave-uniform
        const double *row = sH + lane;
        while (true) {
            const unsigned long long m = __ballot(valid0 && !(lu0 >= r0)) & live;
            if (!m) break;                                     // the rest of the half rejects
            const int f = __builtin_ctzll(m);
            const double h0 = row[f * ldh], h1 = two ? row[f * ldh + 64] : 0.0;
            r0 += lane > f ? h0 : 0.0;      // entries at / below the diagonal are junk
            r1 += h1;
            acc0 |= 1ull << f;
            live = f == 63 ? 0ull : ~0ull << (f + 1);
        }
        live = ~0ull;
        row = sH + (size_t)64 * ldh + 64 + lane;
        while (two) {
            const unsigned long long m = __ballot(valid1 && !(lu1 >= r1)) & live;
            if (!m) break;
            const int f = __builtin_ctzll(m);
            const double h1 = row[f * ldh];
            r1 += lane > f ? h1 : 0.0;
            acc1 |= 1ull << f;
            live = f == 63 ? 0ull : ~0ull << (f + 1);
        }
    }
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int k = lane + 64 * v;
        if (k < nb) {
            const int accepted = (int)(((v == 0 ? acc0 : acc1) >> lane) & 1ull);
            const size_t tj = (size_t)t * N + j0 + k;
            if (accepted) {
                const double *pr = sb.prop + ((size_t)s * N + j0 + k) * (D + 2);
#pragma unroll
                for (int d = 0; d < D; ++d) c.X[tj * D + d] = pr[d];
            }
            double st = c.step[tj];
            int32_t na = c.nacc[tj], ns = c.nsteps[tj], un = c.until[tj];
            metropolis_bookkeeping(st, na, ns, un, c.tune, c.tune_interval, accepted);
            c.step[tj] = st; c.nacc[tj] = na; c.nsteps[tj] = ns; c.until[tj] = un;
        }
    }
}

}  // namespace dlsm
