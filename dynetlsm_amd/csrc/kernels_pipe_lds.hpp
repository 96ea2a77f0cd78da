// The undirected evaluators of the pipelined sweep (k_pipe_step, algo 4), round 6: neighbour rows staged in
// LDS once per workgroup, parts that INTERLEAVE the trips of 64 neighbours, and the H factors computed inside
// the trips that hold their operands.  (Included by kernels_spec_pipe.hpp in front of k_pipe_step.)
//
// What it replaces (pipe_eval_item, still the directed model's evaluator): every wavefront prefetched the
// rows of its part into registers - the 16 wavefronts of a workgroup the SAME rows, eleven 1 KB requests
// each through the CU's vector L1 - and computed, at its tail, one H entry per lane from a flat list: index
// decode, a gather of two proposal rows and a bit word, FOUR distances and exponentials, a scattered 8-byte
// store; the last wavefront's entry was the launch's tail (profiles/r04_h_entry_ablation.md: a launch without
// H entries ran 2.4 us shorter).
//
//   * H[k][m] (how node m's acceptance changes node k's ratio, m in k's window: the previous batch and the
//     earlier nodes of its own) needs d(m0, k0), d(m0, k1), d(m1, k0), d(m1, k1).  The first two - and their
//     exponentials, and the factors 1 + E e^{-d} - are what the item of node k computes anyway in the trip
//     that holds neighbour m at its snapshot position.  That trip now also reads m's PROPOSAL (staged
//     beside the rows) and finishes the factor: two distances and exponentials instead of four, no decode,
//     no gather, and a wavefront's 64 factors are 512 contiguous bytes of row k.
//   * The window is 2 .. 4 consecutive trips (128 + k nodes), so with contiguous parts one item of a node
//     would carry all of them.  A part is therefore a LIST of trips (pipe_plan): the window's trips dealt
//     round robin over the parts, then runs of the other trips sized so that the parts cost the same with a
//     window trip counted as two (it is 2.3: 61 + 49 vector instructions against 49).
//   * The 16 wavefronts of a workgroup are 16 nodes of one (slice, part): one copy of the part's rows in
//     LDS (11 KB at config 2) serves them all, the registers the prefetch held are free, and the rows are
//     requested once, by the whole workgroup, in front of the exp table's barrier.
//   * A node's row of the network: the 64 bits under a trip are one aligned 8-byte word at a wave-uniform
//     address - a scalar load straight into the lane mask of "y = 1" (was: a word per lane and two
//     v_readlane per trip).
// Values: the same arithmetic on the same operands as pipe_h_entry / the old trips - H blocks bit for bit,
// records equal up to the order in which a node's neighbours are summed (parts interleaved, not contiguous).
#pragma once

namespace dlsm {

#define DLSM_CONSTANT_AS __attribute__((address_space(4)))

// wave-uniform 8-byte word through the scalar cache (read-only data: the packed network)
__device__ __forceinline__ unsigned long long scalar_load_u64(const unsigned long long *p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return *(const DLSM_CONSTANT_AS unsigned long long *)p;
#pragma clang diagnostic pop
}

}  // namespace dlsm
#include "pipe_plan.hpp"        // PL_WIN, pipe_lds_trip_cap / _eval_bytes, PipePlan, pipe_plan_entry (plain C++)
namespace dlsm {
static_assert(PL_EXPTAB_DOUBLES == EXPTAB11_N, "pipe_plan.hpp sizes the LDS with the evaluators' exp table");

// (pb: the kernel argument itself - the plan entry is a scalar load at a computed offset)
__device__ __forceinline__ PipePlan pipe_plan(const PipeBuf &pb, int ntrip, int p, int be, int k0) {
    const int nwin = min(2 * be + (k0 >= 64 ? 1 : 0), ntrip - 1) - (be > 0 ? 2 * be - 2 : 0) + 1;
    return pipe_plan_from_entry(pb.lds.plan[nwin - 1][p], ntrip, be, k0);
}

// the rest of H[k][m] from what the trip holds: a0 = d(m0, k0), a1 = d(m0, k1), their exponentials ea0, ea1
// and factors fa = 1 + E ea (pipe_h_entry's arithmetic on the same operands)
template <int D, bool SQ>
__device__ __forceinline__ double pipe_h_finish(const double (&xm1)[D], const double (&xk0)[D],
                                                const double (&xk1)[D], double a0, double a1, double ea0,
                                                double ea1, double fa0, double fa1, bool y1, double E,
                                                const double *etab) {
    const double b0 = dist_fast<D>(xm1, xk0, SQ ? 1 : 0);
    const double b1 = dist_fast<D>(xm1, xk1, SQ ? 1 : 0);
    const double eb0 = SQ ? tab_exp11_clamped(-b0, etab) : tab_exp11(-b0, etab);
    const double eb1 = SQ ? tab_exp11_clamped(-b1, etab) : tab_exp11(-b1, etab);
    double num = fma(E, eb0, 1.0) * fa1;
    double den = fma(E, eb1, 1.0) * fa0;
    // the edge's factor e^{(b0 - b1) - (a0 - a1)} from the four exponentials at hand; a fifth one only
    // when their product left the normal range (distances > 300)
    const double fn = eb1 * ea0, fd = eb0 * ea1;
    const bool tiny = y1 && !(fd > 1e-290 && fn > 1e-290);
    if (y1 && !tiny) { num *= fn; den *= fd; }
    double h = num * fast_rcp(den);
    // (the table exponential here too: fast_exp's dozen constants would sit in registers through every trip)
    if (__builtin_amdgcn_ballot_w64(tiny)) { if (tiny) h *= tab_exp11_clamped((b0 - b1) - (a0 - a1), etab); }
    return h;
}

// Rows beyond the slice's last node (the last trip's idle lanes) are staged as a point this far away on the first
// axis: e^{-d} underflows to 0, the factor 1 + E e^{-d} is exactly 1 and the row's (zero) padding bits keep the
// linear term out - so the trips need no "i < N" lane mask.  (Inside the table exponential's range, |x| < 7.2e5.)
constexpr double PL_FAR = 3.0e5;

// the trips of one item: part p of node k (batch be, slice t); rows in sX, window proposals in sM.
// The SIMD issues one scalar instruction per cycle slot just as it issues one vector instruction, and the four
// items of a SIMD share both ports: what a trip spends on masks, trip numbers and addresses competes with the
// other items' arithmetic (the first form of this loop: 60 scalar instructions beside 47 vector ones, and no
// faster than the item it replaced).  So the part's window trips - the only ones that hold the node itself, an H
// entry or a partial lane mask - run first, in a loop of their own, and the other trips run bare: full exec
// mask, consecutive bit words, a countdown for the issue priority.
template <int D, bool FLUSH, bool SQ>
__device__ __forceinline__ void pipe_lds_trips(int nw64, int P, const PipePlan &pl, int p, int be,
                                               int k, int jk, int lane, const double (&xk0)[D],
                                               const double (&xk1)[D], double E, int nflush,
                                               const unsigned long long *yrow, unsigned long long ym,
                                               const double *etab, const double *sX, const double *sM, double *hrow,
                                               RatioAcc &ra
#ifdef DLSM_PIPE_TIMING
                                               , unsigned long long *ts
#endif
                                               ) {
    const int ntp = pl.trips();
    // (nw64: 8-byte words of a row of the network)
    const double *row = sX + lane * D;                      // this lane's neighbour of trip 0, 1, ..: stride 64 D
    double xi[D];
#pragma unroll
    for (int d = 0; d < D; ++d) xi[d] = row[d];
    // first trip behind the window's: rank pl.s of the other trips
    int g = pl.w > 0 ? pl.glo + p : (pl.s < pl.glo ? pl.s : pl.s + pl.nwin);   // (ym: its bits, requested by the caller)
    const int greg = pl.s < pl.glo ? pl.s : pl.s + pl.nwin;
// the edges' linear term: lin += y (d0 - d1) with y as a 0.0 / 1.0 select of one word (sub, select, fma: 49.5 vector
// instructions per trip; "if (y) lin += d0 - d1" became sub, add and a two-word select: 50.5, C2 4903 against 4923 it/s)
#define DLSM_LDS_LIN() ra.lin = fma(yb ? 1.0 : 0.0, d0 - d1, ra.lin);
#define DLSM_LDS_TERM()                                                                                   \
        const bool yb = __builtin_amdgcn_inverse_ballot_w64(ym);                                          \
        const double d0 = dist_fast<D>(xi, xk0, SQ ? 1 : 0);                                              \
        const double d1 = dist_fast<D>(xi, xk1, SQ ? 1 : 0);                                              \
        const double e0 = SQ ? tab_exp11_clamped(-d0, etab) : tab_exp11(-d0, etab);                       \
        const double e1 = SQ ? tab_exp11_clamped(-d1, etab) : tab_exp11(-d1, etab);                       \
        const double f0 = fma(E, e0, 1.0), f1 = fma(E, e1, 1.0);                                          \
        DLSM_LDS_LIN()                                                                                    \
        ra.P0 *= f0;                                                                                      \
        ra.P1 *= f1;                                                                                      \
        if (FLUSH) if (++ra.cnt >= nflush) ra.flush();
    // ---- the window's trips (previous batch: g = 2 be - 2, 2 be - 1; own batch: 2 be, 2 be + 1) ----------------
    for (int u = 0; u < pl.w; ++u) {
        const int gn = u + 1 < pl.w ? g + P : greg;
        double xn[D];
#pragma unroll
        for (int d = 0; d < D; ++d) xn[d] = row[(size_t)(u + 1) * 64 * D + d];
        const unsigned long long ymn = scalar_load_u64(yrow + min(gn, nw64 - 1));
        const int self = jk - 64 * g;                       // the node itself sits in one of its own batch's trips
        const unsigned long long vm = self >= 0 && self < 64 ? ~(1ull << self) : ~0ull;
        const int wq = g - 2 * be;
        const int nh = wq < 0 ? 64 : k - 64 * wq;           // lanes with m < k
        if (__builtin_amdgcn_inverse_ballot_w64(vm)) {
            DLSM_LDS_TERM()
            if (nh > 0) {
                const unsigned long long hm = nh >= 64 ? ~0ull : (1ull << nh) - 1ull;
                if (__builtin_amdgcn_inverse_ballot_w64(hm)) {
                    double xm1[D];
                    const double *rowm = sM + ((size_t)u * 64 + lane) * D;
#pragma unroll
                    for (int d = 0; d < D; ++d) xm1[d] = rowm[d];
#ifdef DLSM_X_LDS_H1
                    const double h = fma(0.0, xm1[0], 1.0);
#else
                    const double h = pipe_h_finish<D, SQ>(xm1, xk0, xk1, d0, d1, e0, e1, f0, f1, yb, E, etab);
#endif
                    hrow[(wq + 2) * 64 + lane] = h;         // [previous batch (128) | own batch (128)]
                }
            }
        }
        if (u == 0) { DLSM_STAMP(1, ra.P0) }
#pragma unroll
        for (int d = 0; d < D; ++d) xi[d] = xn[d];
        ym = ymn; g = gn;
    }
    // ---- the other trips: ascending, consecutive but for the jump over the window ---------------------------
    row += (size_t)pl.w * 64 * D;
    const int quarter = max((ntp + 3) >> 2, 1);
    int left = max(quarter - pl.w, 1), level = 3;           // issue priority 3 -> 0 by quarters of the item
    const int glast = nw64 - 1;
    // two trips per iteration, the rows and bit words of one requested under the other's arithmetic, in registers
    // of their own (the single-trip form handed the next row over through four copies per trip)
#if DLSM_TRIP_PRIO
#define DLSM_LDS_PRIO_STEP()                                                                              \
        if (--left == 0) {                                                                                \
            left = quarter;                                                                               \
            if (level == 3) __builtin_amdgcn_s_setprio(2);                                                \
            else if (level == 2) __builtin_amdgcn_s_setprio(1);                                           \
            else __builtin_amdgcn_s_setprio(0);                                                           \
            --level;                                                                                      \
        }
#else
#define DLSM_LDS_PRIO_STEP()
#endif
#define DLSM_LDS_NEXT(G_) { ++G_; if (G_ == pl.glo) G_ += pl.nwin; }
    int i = 0;
    for (; i + 1 < pl.r; i += 2) {
        double xb[D];
        int gb = g;
        DLSM_LDS_NEXT(gb)
#pragma unroll
        for (int d = 0; d < D; ++d) xb[d] = row[64 * D + d];
        const unsigned long long ymb = scalar_load_u64(yrow + min(gb, glast));
        {
            DLSM_LDS_TERM()
        }
        if (pl.w == 0 && i == 0) { DLSM_STAMP(1, ra.P0) }
        DLSM_LDS_PRIO_STEP()
        g = gb;
        DLSM_LDS_NEXT(g)
        row += 2 * 64 * D;                                  // (the slots behind the last trip are LDS of this launch too)
#pragma unroll
        for (int d = 0; d < D; ++d) xi[d] = row[d];
        ym = scalar_load_u64(yrow + min(g, glast));
        {
            const double (&xi)[D] = xb;
            const unsigned long long ym = ymb;
            DLSM_LDS_TERM()
        }
        DLSM_LDS_PRIO_STEP()
    }
    if (i < pl.r) {
        DLSM_LDS_TERM()
        if (pl.w == 0 && i == 0) { DLSM_STAMP(1, ra.P0) }
    }
#undef DLSM_LDS_NEXT
#undef DLSM_LDS_PRIO_STEP
#undef DLSM_LDS_TERM
#undef DLSM_LDS_LIN
    DLSM_STAMP(2, ra.P0)
}

// ---- the resolvers' cross products, by the evaluators ------------------------------------------------------
// Row k of the slice-t batch that THIS launch resolves needs prod_{m accepted in the batch before} Hx[k][m]: 128
// factors the previous launch stored and two mask words it left.  The resolver workgroup read all 128 rows (128
// KB through one CU: 1.8 of the 4.1 us before its fixed point could start) while every evaluator wavefront sat
// out a memory round trip behind its staging requests.  Now one wavefront in `xstride` takes a row: its 128
// factors are one 16-byte request per lane in front of the staging requests, the product is reduced across
// the wavefront while those are in flight, and lane 0 stores it past the L1 into the row's slot (PipeBuf::xprod).
// A wavefront waits for nothing here: the resolver can only be kept waiting by evaluators that have not started.
struct PipeXServe { double2 v; unsigned long long m; double *slot; bool on; };
__device__ __forceinline__ double wave_prod_tp(double v, int lane) {     // product over the wavefront, every lane
    v *= dpp_move<0xB1>(v);
    v *= dpp_move<0x4E>(v);
    v *= dpp_move<0x141>(v);
    v *= dpp_move<0x140>(v);
    v *= lane_get(v, (lane ^ 16) << 2);
    v *= lane_get(v, (lane ^ 32) << 2);
    return v;
}
// workgroup `wg` of the evaluators, wavefront `wave`: the first pb.xstride wavefronts of a workgroup serve rows
// wg * xstride + wave (xstride = ceil(rows / evaluator workgroups): no quotient to compute)
__device__ __forceinline__ void pipe_xserve_request(const PipeLds &a, int l, int wg, int wave,
                                                    int lane, PipeXServe &xs) {
    xs.on = false;
    if (!a.xserve || wave >= a.xstride) return;
    const int row = wg * a.xstride + wave;
    if (row >= a.T * PP_B) return;
    const int t = row >> 7, k = row & (PP_B - 1);
    const int b = l - (t & 1);                                            // the batch slice t resolves in this launch
    if (b < 1 || b >= a.nbat || k >= min(PP_B, a.N - b * PP_B)) return;
    // (32-bit offsets from the launch's base pointers: T < 2^7 slices of 2 x 128 x 256 factors)
    const uint32_t moff = (uint32_t)(t * 2 + ((b - 1) & 1)) * PP_ACC + PP_ACC_MASK;
    // (the lane's half of the mask as a VECTOR load: a scalar load from memory this cold would hold up every
    // scalar wait behind it - the kernel arguments the staging addresses are made of)
    const unsigned long long *pmg = (const unsigned long long *)(a.acc + moff);
    xs.m = __builtin_nontemporal_load(pmg + (lane >> 5));
    const uint32_t hoff = (uint32_t)((((b & 1) * a.T + t) * PP_B + k) * (2 * PP_B) + 2 * lane);
    xs.v = *(const double2 *)(a.Hd + hoff);                              // factors of window nodes 2 lane, 2 lane + 1
    xs.slot = a.xprod + (uint32_t)row;
    xs.on = true;
}
__device__ __forceinline__ void pipe_xserve_finish(const PipeXServe &xs, int lane) {
    if (!xs.on) return;
    const unsigned long long w = xs.m >> (2 * (lane & 31));
    const double f = ((w & 1ull) ? xs.v.x : 1.0) * ((w & 2ull) ? xs.v.y : 1.0);
    const double prod = wave_prod_tp(f, lane);
    if (lane == 0) __hip_atomic_store(xs.slot, prod, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifdef DLSM_PIPE_TIMING
#define DLSM_LDS_TS , ts
#else
#define DLSM_LDS_TS
#endif
template <int D>
__device__ __forceinline__ void pipe_eval_lds(const ChainView &c, const PipeBuf &pb, int l, double *lds
#ifdef DLSM_PIPE_TIMING
                                              , unsigned long long t_kernel
#endif
                                              ) {
    constexpr int PW = 2 * D + 2;
    // the launch's arguments in one piece (PipeLds): the pointers and scalars below are 128 contiguous bytes of the
    // kernel's argument block, pinned here so that they are requested together by the first instructions
    PipeLds a;
    a.X = pb.lds.X; a.ybits = pb.lds.ybits; a.prop = pb.lds.prop; a.full0 = pb.lds.full0; a.Hd = pb.lds.Hd;
    a.acc = pb.lds.acc; a.consts = pb.lds.consts; a.xprod = pb.lds.xprod;
    a.T = pb.lds.T; a.N = pb.lds.N; a.W = pb.lds.W; a.squared = pb.lds.squared; a.parts = pb.lds.parts;
    a.nbat = pb.lds.nbat; a.lds_cap = pb.lds.lds_cap; a.xserve = pb.lds.xserve;
    a.beE = pb.lds.beE; a.beO = pb.lds.beO; a.nbE = pb.lds.nbE; a.nbO = pb.lds.nbO; a.nslE = pb.lds.nslE;
    a.nslO = pb.lds.nslO; a.xstride = pb.lds.xstride; a.nsl_magic = pb.lds.nsl_magic;
    asm volatile("" :: "s"(a.X), "s"(a.ybits), "s"(a.prop), "s"(a.full0), "s"(a.Hd), "s"(a.acc), "s"(a.consts), "s"(a.xprod));
    asm volatile("" :: "s"(a.T), "s"(a.N), "s"(a.W), "s"(a.squared), "s"(a.parts), "s"(a.nbat), "s"(a.lds_cap), "s"(a.xserve),
                 "s"(a.beE), "s"(a.beO), "s"(a.nbE), "s"(a.nbO), "s"(a.nslE), "s"(a.nslO), "s"(a.xstride), "s"(a.nsl_magic));
    const int T = a.T, N = a.N;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P = a.parts;
    const int ntrip = (N + 63) >> 6;
    double *sTab = lds;
    double *sX = lds + EXPTAB11_N;                                      // [trip cap][64][D]
    double *sM = sX + (size_t)a.lds_cap * 64 * D;                       // [PL_WIN][64][D]
    // the workgroup's items: 16 consecutive nodes (wg mod 8) of one active slice and part, wg / 8 = p nsl + si;
    // batches, sizes and slice counts of the launch from the host (PipeBuf) - the four wavefronts of a SIMD issue
    // this prologue one after the other, so what it does not compute is time the barrier below comes earlier
    constexpr bool first = true;
    const int wg = (int)blockIdx.x - T;                                  // (also the row server's index)
    const int wgx = wg & (PP_B / PP_WAVES - 1), wr = wg >> 3;
    const int p = (int)(((uint32_t)wr * a.nsl_magic) >> 16);
    const int si = wr - p * (a.nslE + a.nslO);
    PipeXServe xs;
    {
        const int k0 = wgx * PP_WAVES;
        const int nslE = a.nslE;
        const bool odd = si >= nslE;
        const int be = odd ? a.beO : a.beE;
        const int nb = p < P ? (odd ? a.nbO : a.nbE) : 0;              // (workgroups behind the last item: p = P)
        if (k0 >= nb) {         // (workgroup-uniform) no items here: the serving wavefronts have nothing else to do
            pipe_xserve_request(a, l, wg, wave, lane, xs);
            pipe_xserve_finish(xs, lane);
            return;
        }
        const int t = odd ? 2 * (si - nslE) + 1 : 2 * si;
        const int k = k0 + wave;
        const bool live = k < nb;
#ifdef DLSM_PIPE_TIMING
        // stamps (profiles/pipe_timing.py): 0 entry, 4 rows + table staged (behind the barrier), 1 first trip done,
        // 2 last trip done, 3 = 5 record stored
        unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
#endif
        DLSM_STAMP(0, (double)lane)
        const int j0 = be * PP_B, jk = j0 + min(k, nb - 1);
        const int jprev = max(j0 - PP_B, 0);                             // nodes >= jprev: snapshot positions
        // (T N < 2^31: node indices in 32 bits, one 64-bit product per base)
        const uint32_t tN = (uint32_t)t * (uint32_t)N;
        const double *Xt = a.X + (size_t)tN * D;
        const double *props = a.prop + (size_t)tN * PW;
#if DLSM_TRIP_PRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        // ---- every request of the round, then the LDS stores, then ONE barrier ----------------------------
        double2 tabv = make_double2(0.0, 0.0);
        if (first) tabv = ((const double2 *)c_exp2_tab11)[tid];          // 1024 threads x 2 entries
        static_assert(PP_THREADS * 2 == EXPTAB11_N, "two table entries per thread");
        // staging list: the part's trips u = 0 .. ntp - 1 (rows), then the proposals of its window trips (its
        // first pl.w trips)
        const PipePlan pl = pipe_plan(pb, ntrip, p, be, k0);
        const int ntp = pl.trips();
        const int nst = ntp + pl.w;
        // (one entry per wavefront when the list has <= 16: config 2 holds 10 + 2 or 11 + 1)
        double stg[D];
        int s = wave, g_stg = 0;            // (g_stg: the trip of the entry in flight - the store needs it again)
        auto stage_request = [&](int s_) {
            const bool win = s_ >= ntp;
            const int g = pl.trip(win ? s_ - ntp : s_, p, P);
            g_stg = g;
            const int n = min(64 * g + lane, N - 1);
            // a trip's 64 neighbours are all on one side of jprev (both multiples of 64)
            const bool snap = win || 64 * g >= jprev;
            const char *src = snap ? (const char *)(props + (win ? 0 : D + 2)) : (const char *)Xt;
            const uint32_t off = __umul24((uint32_t)n, (uint32_t)((snap ? PW : D) * sizeof(double)));
            coh_load_row<D, false>(src, off, stg);
        };
        auto stage_store = [&](int s_) {
            const int g = g_stg;
            if (64 * g + 63 >= N) {             // the slice's last trip: its idle lanes hold the far point
                const bool idle = 64 * g + lane >= N;
#pragma unroll
                for (int d = 0; d < D; ++d) stg[d] = idle ? (d == 0 ? PL_FAR : 0.0) : stg[d];
            }
            double *dst = (s_ >= ntp ? sM + (size_t)(s_ - ntp) * 64 * D : sX + (size_t)s_ * 64 * D) + lane * D;
#pragma unroll
            for (int d = 0; d < D; ++d) dst[d] = stg[d];
        };
        if (s < nst) stage_request(s);
        // the wavefront's own node and the launch constants
        double xk0[D], xk1[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            xk0[d] = props[(size_t)jk * PW + D + 2 + d];
            xk1[d] = props[(size_t)jk * PW + d];
        }
        const double E = a.consts[0];
        const int nflush = (int)a.consts[1];
        // the node's row of the network: the bits under its first trip
        const unsigned long long *yrow = (const unsigned long long *)(a.ybits + (size_t)(tN + (uint32_t)jk) * (uint32_t)a.W);
        // (a part without trips - more parts than trips at tiny N - names a trip behind the row: clamped, unused)
        const unsigned long long ym0 = scalar_load_u64(yrow + min(pl.trip(0, p, P), (a.W >> 1) - 1));
        double *hrow = a.Hd + (size_t)((uint32_t)(((be & 1) * T + t) * PP_B + min(k, nb - 1)) * (uint32_t)(2 * PP_B));     // (< 2^31 doubles: T < 128)
#if defined(DLSM_PIPE_TIMING) && DLSM_PIPE_TIMING == 2      // (prologue probe: slot 1 = requests issued, slot 2 = at the barrier)
        DLSM_STAMP(1, (double)lane)
#endif
        if (first) ((double2 *)sTab)[tid] = tabv;
        if (s < nst) stage_store(s);
        for (s += PP_WAVES; s < nst; s += PP_WAVES) { stage_request(s); stage_store(s); }
#if defined(DLSM_PIPE_TIMING) && DLSM_PIPE_TIMING == 2
        DLSM_STAMP(2, xk0[0])
        unsigned long long ts1 = ts[1], ts2 = ts[2];
#endif
        __syncthreads();
        // The whole of it behind the barrier: in front of it the four wavefronts of a SIMD issue their prologues
        // one after the other and every instruction of a serving wavefront keeps the workgroup's other fifteen
        // waiting (+0.35 us on the barrier, measured); behind it a wavefront that sits out its request's round
        // trip costs nothing - the SIMD's other three have trips to issue.
        pipe_xserve_request(a, l, wg, wave, lane, xs);
        pipe_xserve_finish(xs, lane);
        DLSM_STAMP(4, xk0[0])
        if (live) {
            RatioAcc ra;
            const bool noflush = nflush >= 64 * ntp && !a.squared;
            if (noflush) pipe_lds_trips<D, false, false>(a.W >> 1, P, pl, p, be, k, jk, lane, xk0, xk1, E, nflush, yrow, ym0, sTab, sX, sM, hrow, ra DLSM_LDS_TS);
            else if (a.squared) pipe_lds_trips<D, true, true>(a.W >> 1, P, pl, p, be, k, jk, lane, xk0, xk1, E, nflush, yrow, ym0, sTab, sX, sM, hrow, ra DLSM_LDS_TS);
            else pipe_lds_trips<D, true, false>(a.W >> 1, P, pl, p, be, k, jk, lane, xk0, xk1, E, nflush, yrow, ym0, sTab, sX, sM, hrow, ra DLSM_LDS_TS);
            double tot_l, tot_r;
            if (noflush) {
                // the products of the whole wave stay in range: multiply across lanes
                pipe_reduce(ra.lin + ra.lg, ra.P0, ra.P1, lane, tot_l, tot_r);
            } else {
                tot_l = wave_sum_all(ra.value()); tot_r = 1.0;
            }
            if (lane == 0) {
                double2 *f = (double2 *)a.full0 + (uint32_t)((((be & 1) * T + t) * PP_B + k) * P + p);
                *f = make_double2(tot_l, tot_r);
            }
#ifdef DLSM_PIPE_TIMING
            DLSM_STAMP(3, tot_r)
            ts[5] = ts[3];
#if DLSM_PIPE_TIMING == 2
            ts[1] = ts1; ts[2] = ts2;
#endif
            ts[0] = t_kernel;           // (the round's entry stamp gives way to the wavefront's first stamp in the kernel)
            const int tgw = wg * PP_WAVES + wave;
            if (lane == 0 && l + 1 >= 0 && l + 1 < 24 && tgw < 4096)
                for (int i = 0; i < 6; ++i) g_pipe_item_t[l + 1][tgw][i] = ts[i];
#endif
        }
    }
}

}  // namespace dlsm
