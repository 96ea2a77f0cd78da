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
    double *consts;  // [2] : E = exp(sum of intercepts), flush interval
    int B, parts;
    int s0;          // first slice (of the parity) served by this launch
    int per;         // neighbours per part = ceil(N / parts)
    unsigned long long *stamps;   // profiling only: [start, end] per workgroup of this
                                  // launch in 100 MHz wall-clock ticks (else nullptr)
};

__device__ __forceinline__ int flush_interval(double E_max) {
    // (1 + E)^n < e^600
    const double l = log1p(E_max);
    if (!(l > 0.0)) return 1 << 20;
    const double n = 600.0 / l;
    return n < 1.0 ? 1 : (n > 1048576.0 ? 1 << 20 : (int)n);
}

template <int D>
__global__ __launch_bounds__(256) void k_spec_propose(ChainView c, SpecBuf sb,
                                                      IterRef ir, int parity) {
    const uint32_t iter = ir.get();
    const int N = c.N;
    const int s = sb.s0 + blockIdx.y;
    const int t = 2 * s + parity;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        const double E = c.model == DLSM_UNDIRECTED ? exp(c.intercept[0])
                                                    : exp(c.intercept[0] + c.intercept[1]);
        sb.consts[0] = E;
        sb.consts[1] = (double)flush_interval(E);
    }
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



constexpr int SP_EV_THREADS = 256;     // eval workgroup (>= SP_BMAX so tid covers m < k)

// SBM = capacity of a super-batch (S sub-batches of SP_BMAX nodes): sizes the LDS copy
// of the batch positions / proposals.
template <int D, int MODEL, int SBM>
__global__ __launch_bounds__(SP_EV_THREADS) void k_spec_eval(ChainView c, SpecBuf sb,
                                                          int parity, int j0, int nb) {
    __shared__ double sx0[SBM * D];
    __shared__ double sx1[SBM * D];
    __shared__ double sRed[SP_EV_THREADS / 64];
    __shared__ double sLin[SP_EV_THREADS], sP0[SP_EV_THREADS], sP1[SP_EV_THREADS];
    const int tid = threadIdx.x;
    const int N = c.N, W = c.W;
    const unsigned bl = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (sb.stamps && threadIdx.x == 0) sb.stamps[2 * (size_t)bl] = wall_clock64();
    // grid (parts, nb, slices): no integer divisions in the prologue
    const int p = blockIdx.x;
    const int k = blockIdx.y;
    const int s = sb.s0 + blockIdx.z;
    const int t = 2 * s + parity;
    const double *Xt = c.X + (size_t)t * N * D;
    const int jk = j0 + k;
    const uint32_t *yr = c.ybits + ((size_t)t * N + jk) * W;
    const uint32_t *yc = MODEL == DLSM_DIRECTED ? c.ytbits + ((size_t)t * N + jk) * W
                                                : nullptr;
    const int per = sb.per;
    const int lo = p * per, hi = min(N, lo + per);
    // issue the first neighbours' loads before the staging barrier so that they are
    // in flight together with the prologue's (the kernel is latency bound)
    constexpr int NPRE = 4;
    double xpre[NPRE][D];
    uint32_t wpre[NPRE], wcpre[NPRE];
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
        const int ic = min(lo + tid + u * SP_EV_THREADS, N - 1);
#pragma unroll
        for (int d = 0; d < D; ++d) xpre[u][d] = Xt[(size_t)ic * D + d];
        wpre[u] = yr[ic >> 5];
        wcpre[u] = MODEL == DLSM_DIRECTED ? yc[ic >> 5] : 0u;
    }
    const double E = sb.consts[0];
    const int nflush = (int)sb.consts[1];
    // positions / proposals of the batch nodes m <= k
    for (int m = tid; m <= k; m += SP_EV_THREADS) {
        const int j = j0 + m;
        const double *pr = sb.prop + ((size_t)s * N + j) * (D + 2);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            sx0[m * D + d] = Xt[(size_t)j * D + d];
            sx1[m * D + d] = pr[d];
        }
    }
    __syncthreads();
    double xk0[D], xk1[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { xk0[d] = sx0[k * D + d]; xk1[d] = sx1[k * D + d]; }
    double bin = 0.0, bout = 0.0, irk = 0.0;
    if (MODEL == DLSM_DIRECTED) {
        bin = c.intercept[0]; bout = c.intercept[1];
        irk = 1.0 / c.radii[jk];
    }
    double acc = 0.0;
    RatioAcc ra;
    auto term = [&](int i, const double *xi, int ybit, int ycbit) {
        if (MODEL == DLSM_UNDIRECTED) {
            const double d0 = dist_fast<D>(xi, xk0, c.squared);
            const double d1 = dist_fast<D>(xi, xk1, c.squared);
            if (ybit) ra.lin += d0 - d1;
            ra.P0 *= fma(E, fast_exp(-d0), 1.0);
            ra.P1 *= fma(E, fast_exp(-d1), 1.0);
            if (++ra.cnt >= nflush) ra.flush();
        } else {
            const double d0 = dist_of<D>(xi, xk0, c.squared);
            const double d1 = dist_of<D>(xi, xk1, c.squared);
            const double iri = 1.0 / c.radii[i];
            acc += delta_directed(d0, d1, ybit, ycbit, bin * iri + bout * irk,
                                  bin * irk + bout * iri, E);
        }
    };
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
        const int i = lo + tid + u * SP_EV_THREADS;
        if (i < hi && i != jk)
            term(i, xpre[u], (wpre[u] >> (i & 31)) & 1, (wcpre[u] >> (i & 31)) & 1);
    }
    for (int i = lo + tid + NPRE * SP_EV_THREADS; i < hi; i += SP_EV_THREADS) {
        if (i == jk) continue;
        double xi[D];
#pragma unroll
        for (int d = 0; d < D; ++d) xi[d] = Xt[(size_t)i * D + d];
        term(i, xi, bit_of(yr, i), MODEL == DLSM_DIRECTED ? bit_of(yc, i) : 0);
    }
    // Workgroup total.  Undirected and the products of the whole workgroup stay in
    // range (few factors per thread, (1 + E) small): sum the linear parts and MULTIPLY
    // the products across the workgroup, one log at the end (lane 0 of wave 0) instead
    // of a division + log per thread and a shuffle tree per wave.
    // (1 + E)^(hi - lo) < e^600  <=>  flush interval >= number of factors
    const bool prod_path = MODEL == DLSM_UNDIRECTED && nflush >= hi - lo;
    if (prod_path) {
        sLin[tid] = ra.lin + ra.lg; sP0[tid] = ra.P0; sP1[tid] = ra.P1;
        __syncthreads();
        if (tid < 64) {
            double l = 0.0, q0 = 1.0, q1 = 1.0;
#pragma unroll
            for (int w = 0; w < SP_EV_THREADS / 64; ++w) {
                l += sLin[tid + 64 * w]; q0 *= sP0[tid + 64 * w]; q1 *= sP1[tid + 64 * w];
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                l += __shfl_xor(l, off, 64);
                q0 *= __shfl_xor(q0, off, 64);
                q1 *= __shfl_xor(q1, off, 64);
            }
            if (tid == 0) sb.full0[((size_t)s * sb.B + k) * sb.parts + p] = l + log(q0 / q1);
        }
    } else {
        if (MODEL == DLSM_UNDIRECTED) acc = ra.value();
        const double total = block_sum_all<SP_EV_THREADS / 64>(acc, sRed, tid);
        if (tid == 0) sb.full0[((size_t)s * sb.B + k) * sb.parts + p] = total;
    }
    // effect of an earlier batch node's acceptance on this node's ratio; the parts of
    // node k share the rows: part p takes m = p, p + parts, ...
    for (int m = tid * sb.parts + p; m < k; m += SP_EV_THREADS * sb.parts) {
        const int jm = j0 + m;
        const double a0 = dist_fast<D>(&sx0[m * D], xk0, c.squared);
        const double a1 = dist_fast<D>(&sx0[m * D], xk1, c.squared);
        const double b0 = dist_fast<D>(&sx1[m * D], xk0, c.squared);
        const double b1 = dist_fast<D>(&sx1[m * D], xk1, c.squared);
        double h;
        if (MODEL == DLSM_UNDIRECTED) {
            // g1 - g0 with the four softplus terms under one log
            const double num = fma(E, fast_exp(-b0), 1.0) * fma(E, fast_exp(-a1), 1.0);
            const double den = fma(E, fast_exp(-b1), 1.0) * fma(E, fast_exp(-a0), 1.0);
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
    if (sb.stamps) {
        __syncthreads();
        if (threadIdx.x == 0) sb.stamps[2 * (size_t)bl + 1] = wall_clock64();
    }
}

// ---------------------------------------------------------------------------
// Speculative-batch eval for the case-control likelihood (a3 inside a9/a10)
// (sub-batches of one: the launcher uses S = 1 for this model):
// one wave per (slice, batch node k).  Node k's ratio has O(deg + 2C) gathered
// terms; H[k][m] is non-zero only for the few earlier batch nodes m that appear
// in k's edge / control lists, so the column of H is zero-filled and the hits
// are accumulated in term order (deterministic) by lane 0.
//   term kinds: 0 in-edge, 1 out-edge (eta - softplus(eta)),
//               2 in-control, 3 out-control (- adj * softplus(eta))
// ---------------------------------------------------------------------------
constexpr int SPCC_MAXHIT = 2 * SP_BMAX;   // an earlier batch node can sit in at most one
                                           // in-list and one out-list of node k

template <int D>
__global__ __launch_bounds__(64) void k_spec_eval_cc(ChainView c, SpecBuf sb,
                                                     const int32_t *__restrict__ nctrl,
                                                     int parity, int j0, int nb) {
    __shared__ int sHitM[SPCC_MAXHIT];
    __shared__ double sHitV[SPCC_MAXHIT];
    __shared__ double sCol[SP_BMAX];
    __shared__ int sNhit;
    const int lane = threadIdx.x;
    const int N = c.N;
    const int k = blockIdx.x % nb;
    const int s = sb.s0 + blockIdx.x / nb;
    const int t = 2 * s + parity;
    const int jk = j0 + k;
    const size_t node = (size_t)t * N + jk;
    const double *Xt = c.X + (size_t)t * N * D;
    const double *prk = sb.prop + ((size_t)s * N + jk) * (D + 2);
    double xk0[D], xk1[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { xk0[d] = Xt[(size_t)jk * D + d]; xk1[d] = prk[d]; }
    const double bin = c.intercept[0], bout = c.intercept[1];
    const double rj = c.radii[jk];
    const int in_deg = c.degree[node * 2], out_deg = c.degree[node * 2 + 1];
    const int nci = nctrl[node * 2], nco = nctrl[node * 2 + 1];
    const double adj_in = (double)(N - in_deg - 1) / (double)nci;
    const double adj_out = (double)(N - out_deg - 1) / (double)nco;
    const int total_terms = in_deg + out_deg + nci + nco;
    if (lane == 0) sNhit = 0;
    for (int m = lane; m < SP_BMAX; m += 64) sCol[m] = 0.0;      // column k of H
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    double acc = 0.0;
    for (int q0 = 0; q0 < total_terms; q0 += 64) {
        const int q = q0 + lane;
        int e = -1, kind = 0;
        if (q < total_terms) {
            int r = q;
            if (r < in_deg) { e = c.in_edges[node * c.Din + r]; kind = 0; }
            else if ((r -= in_deg) < out_deg) { e = c.out_edges[node * c.Dout + r]; kind = 1; }
            else if ((r -= out_deg) < nci) { e = c.ctrl_in[node * c.C + r]; kind = 2; }
            else { r -= nci; e = c.ctrl_out[node * c.C + r]; kind = 3; }
        }
        double contrib = 0.0, hval = 0.0;
        bool hit = false;
        if (e >= 0) {
            double xe[D];
#pragma unroll
            for (int d = 0; d < D; ++d) xe[d] = Xt[(size_t)e * D + d];
            const double re = c.radii[e];
            const bool in_dir = (kind == 0 || kind == 2);
            const double wsp = kind < 2 ? 1.0 : (kind == 2 ? adj_in : adj_out);
            auto eta_of = [&](double dd) {
                return in_dir ? bin * (1 - dd / rj) + bout * (1 - dd / re)
                              : bin * (1 - dd / re) + bout * (1 - dd / rj);
            };
            // delta of this term when k moves, the neighbour at position xn
            auto delta = [&](const double *xn, bool self) {
                const double d0 = self ? 0.0 : dist_of<D>(xn, xk0, c.squared);
                const double d1 = self ? 0.0 : dist_of<D>(xn, xk1, c.squared);
                const double e0 = eta_of(d0), e1 = eta_of(d1);
                const double sp = log((1.0 + exp(e1)) / (1.0 + exp(e0)));
                return (kind < 2 ? (e1 - e0) : 0.0) - wsp * sp;
            };
            contrib = delta(xe, e == jk);
            if (e >= j0 && e < jk) {        // an earlier node of this batch
                const double *pre = sb.prop + ((size_t)s * N + e) * (D + 2);
                double xe1[D];
#pragma unroll
                for (int d = 0; d < D; ++d) xe1[d] = pre[d];
                hval = delta(xe1, false) - contrib;
                hit = true;
            }
        }
        acc += contrib;
        // record the hits of this chunk in term order
        const unsigned long long hm = __ballot(hit);
        if (hit) {
            const int pos = sNhit + __popcll(hm & ((1ull << lane) - 1ull));
            if (pos < SPCC_MAXHIT) { sHitM[pos] = e - j0; sHitV[pos] = hval; }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) sNhit = min(sNhit + (int)__popcll(hm), 1 << 20);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
    const double total = wave_sum_all(acc);
    if (lane == 0) {
        sb.full0[((size_t)s * sb.B + k) * sb.parts] = total;
        const int nh = min(sNhit, SPCC_MAXHIT);
        for (int q = 0; q < nh; ++q) sCol[sHitM[q]] += sHitV[q];     // term order
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (int m = lane; m < k; m += 64) sb.Ht[((size_t)s * sb.B + m) * sb.B + k] = sCol[m];
}

constexpr int SP_RES_THREADS = 1024;    // 16 waves: wide staging + parallel passes
constexpr int SP_SMAX = 4;              // sub-batches per super-batch (launch pair)

// Resolve one super-batch of S <= SP_SMAX sub-batches of <= 128 nodes, in order.
//
// The in-order accept/reject rule of a batch is the triangular system
//     a_k = [ log u_k < r_k + sum_{m<k} a_m H[k][m] ],   k = 0, 1, ...
// Instead of walking it node by node (one dependent LDS read + ballot per acceptance)
// iterate a -> F(a): given a guess of the accepted set, add the H rows of the guessed
// nodes to the ratios of the later nodes and re-test.  A fixed point satisfies the
// triangular system, whose solution is unique (forward substitution), so it IS the
// sequential result; |H| is tiny next to the ratios, so the guess "accept iff the
// uncorrected ratio accepts" is stable after 1-2 passes.  The pass is spread over the
// 16 waves: wave w serves the nodes of half (w & 1) and the guessed rows
// m in [16 (w >> 1), +16); waves 0 / 1 own the ratios of half 0 / 1, combine the 8
// partial sums in ascending-m order and re-ballot.
// For sub-batches after the first, the (final) acceptances of the earlier sub-batches
// enter as a gathered row sum of the off-diagonal part of H, also done by all waves.
template <int D>
__global__ __launch_bounds__(SP_RES_THREADS) void k_spec_resolve(ChainView c, SpecBuf sb,
                                                                 int parity, int j0, int nsb) {
    extern __shared__ __attribute__((aligned(16))) double sH[];   // 128 x 128 diagonal block
    __shared__ double sPart[16 * 64];
    __shared__ unsigned long long sMask[2][2];
    __shared__ int sAcc[SP_SMAX * SP_BMAX];                       // accepted nodes so far
    __shared__ int sNacc;
    static_assert(SP_BMAX == 128, "two 64-node halves");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = sb.s0 + blockIdx.x;
    const int t = 2 * s + parity;
    const int N = c.N;
    const int ldg = sb.B;                                         // leading dimension of H
    const double *Ht = sb.Ht + (size_t)s * sb.B * sb.B;
    const int half = wave & 1, part = wave >> 1;
    const int k = 64 * half + lane;
    const bool owner = wave < 2;
    if (tid == 0) sNacc = 0;
    // Software pipeline over the sub-batches: the diagonal block and the owners' per-node
    // data of sub-batch b + 1 are loaded into registers while sub-batch b is resolved
    // (plain global loads stay in flight across the barriers), so only the first
    // sub-batch of a launch exposes its global-load latency.
    // One launch stages at most 128 x 128 doubles: 8 double2 per thread.
    // (written as macros, not lambdas: arrays captured by reference end up in scratch)
    double2 blk0, blk1, blk2, blk3, blk4, blk5, blk6, blk7;
    double o_r = 0.0, o_lu = 0.0, o_st = 0.0, o_x1[D];
    int32_t o_na = 0, o_ns = 0, o_un = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) o_x1[d] = 0.0;
    // rows o..o+nb-1, columns o..o+127 of H^T; entries at or below the diagonal and
    // columns beyond the batch are never read (addresses clamped, no predication: a
    // predicated load makes hipcc branch around it and wait vmcnt(0) per element)
#define DLSM_LOAD1(U, REG, O_, NB_)                                                    \
    {                                                                                   \
        const int q_ = min((U) * SP_RES_THREADS + tid, (NB_) * (SP_BMAX / 2) - 1);      \
        const int col_ = min((O_) + 2 * (q_ & 63), ldg - 2);                            \
        REG = *(const double2 *)(Ht + (size_t)((O_) + (q_ >> 6)) * ldg + col_);         \
    }
#define DLSM_LOAD_BLOCK(O_, NB_)                                                        \
    DLSM_LOAD1(0, blk0, O_, NB_) DLSM_LOAD1(1, blk1, O_, NB_) DLSM_LOAD1(2, blk2, O_, NB_) \
    DLSM_LOAD1(3, blk3, O_, NB_) DLSM_LOAD1(4, blk4, O_, NB_) DLSM_LOAD1(5, blk5, O_, NB_) \
    DLSM_LOAD1(6, blk6, O_, NB_) DLSM_LOAD1(7, blk7, O_, NB_)
#define DLSM_STORE1(U, REG, NB_)                                                        \
    ((double2 *)sH)[min((U) * SP_RES_THREADS + tid, (NB_) * (SP_BMAX / 2) - 1)] = REG;
#define DLSM_LOAD_OWNER(O_, NB_)                                                        \
    if (owner) {                                                                        \
        const int kc_ = (O_) + min(k, (NB_) - 1);                                       \
        const double *f_ = sb.full0 + ((size_t)s * sb.B + kc_) * sb.parts;              \
        const int p1_ = sb.parts;                                                       \
        double t0_ = f_[0], t1_ = f_[min(1, p1_ - 1)], t2_ = f_[min(2, p1_ - 1)],        \
               t3_ = f_[min(3, p1_ - 1)], t4_ = f_[min(4, p1_ - 1)],                      \
               t5_ = f_[min(5, p1_ - 1)], t6_ = f_[min(6, p1_ - 1)],                      \
               t7_ = f_[min(7, p1_ - 1)];                                               \
        double tot_ = t0_;                                                              \
        tot_ += 1 < p1_ ? t1_ : 0.0; tot_ += 2 < p1_ ? t2_ : 0.0;                        \
        tot_ += 3 < p1_ ? t3_ : 0.0; tot_ += 4 < p1_ ? t4_ : 0.0;                        \
        tot_ += 5 < p1_ ? t5_ : 0.0; tot_ += 6 < p1_ ? t6_ : 0.0;                        \
        tot_ += 7 < p1_ ? t7_ : 0.0;                                                    \
        const double *pr_ = sb.prop + ((size_t)s * N + j0 + kc_) * (D + 2);             \
        o_r = tot_ + pr_[D + 1];                                                        \
        o_lu = pr_[D];                                                                  \
        _Pragma("unroll") for (int d = 0; d < D; ++d) o_x1[d] = pr_[d];                 \
        const size_t tjc_ = (size_t)t * N + j0 + kc_;                                   \
        o_st = c.step[tjc_]; o_na = c.nacc[tjc_]; o_ns = c.nsteps[tjc_];                \
        o_un = c.until[tjc_];                                                           \
    }
    {
        const int nb0 = min(SP_BMAX, nsb);
        DLSM_LOAD_OWNER(0, nb0)
        DLSM_LOAD_BLOCK(0, nb0)
    }
    for (int o = 0; o < nsb; o += SP_BMAX) {
        const int nb = min(SP_BMAX, nsb - o);
        const bool valid = k < nb;
        // this sub-batch's data (loaded one iteration ago) -> working registers / LDS
        double r = o_r, lu = o_lu, st = o_st, x1[D];
        int32_t na = o_na, ns = o_ns, un = o_un;
#pragma unroll
        for (int d = 0; d < D; ++d) x1[d] = o_x1[d];
        DLSM_STORE1(0, blk0, nb) DLSM_STORE1(1, blk1, nb) DLSM_STORE1(2, blk2, nb)
        DLSM_STORE1(3, blk3, nb) DLSM_STORE1(4, blk4, nb) DLSM_STORE1(5, blk5, nb)
        DLSM_STORE1(6, blk6, nb) DLSM_STORE1(7, blk7, nb)
        if (o + SP_BMAX < nsb) {                                  // prefetch the next one
            const int nbn = min(SP_BMAX, nsb - o - SP_BMAX);
            DLSM_LOAD_OWNER(o + SP_BMAX, nbn)
            DLSM_LOAD_BLOCK(o + SP_BMAX, nbn)
        }
        __syncthreads();                                          // sNacc / sAcc visible
        if (o > 0) {
            // nodes accepted in earlier sub-batches: gathered row sums of H from global
            const int nacc = sNacc;
            const double *colp = Ht + min(o + k, ldg - 1);
            double sum = 0.0;
            int a = part;
            for (; a + 24 < nacc; a += 32) {
                const double h0 = colp[(size_t)sAcc[a] * ldg];
                const double h1 = colp[(size_t)sAcc[a + 8] * ldg];
                const double h2 = colp[(size_t)sAcc[a + 16] * ldg];
                const double h3 = colp[(size_t)sAcc[a + 24] * ldg];
                sum += h0; sum += h1; sum += h2; sum += h3;
            }
            for (; a < nacc; a += 8) sum += colp[(size_t)sAcc[a] * ldg];
            sPart[wave * 64 + lane] = sum;
            __syncthreads();
            if (owner) {
#pragma unroll
                for (int p = 0; p < 8; ++p) r += sPart[(2 * p + half) * 64 + lane];
            }
            __syncthreads();
        }
        if (owner) {
            const unsigned long long g = __ballot(valid && !(lu >= r));
            if (lane == 0) sMask[0][half] = g;
        }
        __syncthreads();
        int cur = 0;
        for (int pass = 0; pass < 2 * SP_BMAX + 2; ++pass) {
            // partial correction of node k from the guessed rows of this wave's range
            const unsigned long long gm = sMask[cur][part >> 2];
            unsigned int bits = (unsigned int)(gm >> (16 * (part & 3))) & 0xFFFFu;
            const int mbase = 16 * part;
            double sum = 0.0;
            if (half == 1 || part < 4) {                          // rows >= 64 never touch half 0
                const double *col = sH + k;
                while (bits) {
                    int f[4];
                    double h[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        f[u] = bits ? mbase + __builtin_ctz(bits) : 1 << 20;
                        bits &= bits - 1u;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        h[u] = col[(f[u] < (1 << 20) ? f[u] : 0) * SP_BMAX];
#pragma unroll
                    for (int u = 0; u < 4; ++u) sum += k > f[u] ? h[u] : 0.0;
                }
            }
            sPart[wave * 64 + lane] = sum;
            __syncthreads();
            if (owner) {
                double q = r;
#pragma unroll
                for (int p = 0; p < 8; ++p) q += sPart[(2 * p + half) * 64 + lane];
                const unsigned long long g = __ballot(valid && !(lu >= q));
                if (lane == 0) sMask[cur ^ 1][half] = g;
            }
            __syncthreads();
            const bool same = sMask[cur ^ 1][0] == sMask[cur][0] &&
                              sMask[cur ^ 1][1] == sMask[cur][1];
            cur ^= 1;
            if (same) break;
        }
        if (owner) {
            const unsigned long long m0 = sMask[cur][0], m1 = sMask[cur][1];
            const unsigned long long mine = half == 0 ? m0 : m1;
            const int accepted = (int)((mine >> lane) & 1ull);
            if (valid) {
                const size_t tj = (size_t)t * N + j0 + o + k;
                if (accepted) {
#pragma unroll
                    for (int d = 0; d < D; ++d) c.X[tj * D + d] = x1[d];
                }
                metropolis_bookkeeping(st, na, ns, un, c.tune, c.tune_interval, accepted);
                c.step[tj] = st; c.nacc[tj] = na; c.nsteps[tj] = ns; c.until[tj] = un;
            }
            // append the accepted nodes (super-batch local index, ascending)
            if (accepted) {
                const int base = sNacc + (half == 0 ? 0 : __popcll(m0));
                sAcc[base + __popcll(mine & ((1ull << lane) - 1ull))] = o + k;
            }
        }
        __syncthreads();
        if (tid == 0) sNacc += __popcll(sMask[cur][0]) + __popcll(sMask[cur][1]);
        __syncthreads();
    }
}

#undef DLSM_LOAD1
#undef DLSM_LOAD_BLOCK
#undef DLSM_STORE1
#undef DLSM_LOAD_OWNER

}  // namespace dlsm
