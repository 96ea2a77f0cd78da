// Pipelined speculative sweep for the case-control likelihood with SPARSE corrections
// (algo 5; a3 inside a9 / a10: directed_likelihoods_fast.pyx:83-182 under
// sample_latent_positions.py:92-206).
//
// The dense form (kernels_spec_pipe.hpp, MODEL = case-control) keeps the corrections H[k][m]
// - how the acceptance of node m changes node k's ratio - in 128 x 128 blocks, which caps a
// batch at 128 nodes: 81 latency-bound launches per sweep at T = 5, N = 10 000.  But node k's
// ratio only depends on the O(deg + 2C) nodes in its edge / control lists, so of the nodes not
// yet resolved when k is evaluated (the window: the previous batch and the earlier nodes of
// k's own batch) only a few matter: ~18 of 1024 at C4 with batches of 512.  Here the
// evaluator files exactly those as two LISTS per node - corrections for nodes of the previous
// batch (whose acceptances are final when k's batch is resolved) and for earlier nodes of k's
// own batch - and the resolver runs the same fixed-point solve of the in-order accept / reject
// rule over the lists:
//
//     a_k = [ log u_k < r_k + sum_{(m, h) in own_k, m accepted} h ]
//
// A fixed point of a -> F(a) satisfies the triangular system, whose solution is unique, so it
// is the sequential scan's result.  Batches of CP_B = 512 nodes: ceil(N / 512) + 2 launches per
// sweep (22 at C4; measured 1517 it/s against 1506 / 1475 / 1471 for batches of 640 / 768 / 1024
// and 1359 for 384: larger batches lengthen the lists and the resolver, smaller ones pay the
// launch floor more often), each with T resolver workgroups (thread = node) beside ~250 evaluator
// workgroups (one wavefront per node, every load of a gather level in flight together).  The lists are stored entry-major ([entry][node]) so
// that the resolver's thread-per-node walks are coalesced; a node's first CP_OWN_REGS own
// entries stay in registers across the passes of the solve.  Same snapshot rule, same
// one-batch lag of the odd slices, same decisions as the dense form and the CPU oracle.
#pragma once
#include "kernels_spec_pipe.hpp"
#include "cc_rows.hpp"

namespace dlsm {

constexpr int CP_B = 512;               // nodes per batch (<= CP_THREADS: the resolver's thread = node)
static_assert(CP_B == CC_SORT_B, "the term rows are sorted inside the sweep's batches (cc_rows.hpp)");
constexpr int CP_THREADS = 1024;
constexpr int CP_WAVES = CP_THREADS / 64;
// (12 / 24 up front measured best at config 4: 8 / 16 1637, 10 / 20 1679, 12 / 24 1731-1744,
// 14 / 24 1739, 12 / 32 1700, 16 / 32 1557 it/s - every own entry is an LDS read per pass)
constexpr int CP_XCH = 24;              // cross entries of a node requested up front (see ccpipe_resolve)
constexpr int CP_OWN_REGS = 12;         // own entries a resolver thread keeps in registers
constexpr int CP_OWN_LDS = 4;           // ... and the next ones in LDS (one node in a hundred has more than 12,
                                        // so every other wavefront holds one: a trip to memory per pass otherwise)

struct CcPipeBuf {
    double *prop;            // [T][N][2D + 2] : x1[D], u, (unused), x0[D] (snapshot)
    double *tot;             // [2][T][CP_B] : log-ratio of node k with its partners at their snapshot / final positions
    double *xval, *oval;     // [2][T][cap][CP_B] : cross / own corrections, entry-major
    int32_t *xidx, *oidx;    // same shape: index of the node inside its batch
    int32_t *cnt;            // [2][T][CP_B][2] : entries of the two lists
    double *cur, *snap;      // [T][N][RW] : (x[D], r) records of the current / the snapshot positions
    unsigned long long *accmask;   // [T][CP_WAVES] : accepted nodes of the last resolved batch
    const int32_t *nctrl;    // valid controls per (t, i, direction)
    const int32_t *terms;    // [T][N][tw] : (in_deg, out_deg, nci, nco, adj_in, adj_out | in-edges, out-edges, in-controls, out-controls)
    int cap, nbat, tw;
    // the cross-sum helpers (ccpipe_cross_helper): helpers != 0 puts T such workgroups between the resolvers
    // and the evaluators
    double *xsum;            // [T][CP_B] : a slice's cross sums, from its helper to its resolver (CC_XS_EMPTY: not yet)
    int32_t *err;            // sticky error word (mapped host memory): a hand-over ran out of its poll budget
    int helpers, budget;
};
constexpr int CC_ERR_HELPER = 1 << 30;
constexpr int CC_ERR_FIXPOINT = 1 << 28;    // a resolver's fixed point did not settle inside its spin bound
// the process's word for it (mapped host memory; set by the first dlsm_create, read by check_pipe_err)
__device__ int32_t *g_cc_fixpoint_err = nullptr;
// "no sum yet": a NaN payload no arithmetic produces (the hardware's own NaN is 0x7FF8000000000000, and the
// entries that are summed carry no payloads)
constexpr unsigned long long CC_XS_EMPTY = 0x7FF8C0DE5EED0001ull;

// A gathered term needs its partner's position and radius: one record (32 bytes up to d = 3)
// instead of two arrays halves the cache-line requests the evaluator is bound by.  The record holds
// 1 / r (round 5): the linear predictor needs b / r, and a reciprocal per gathered term was five of
// the evaluator's ~110 float64 instructions per term (the radii change once per iteration).  `cur` follows
// the chain (the resolver writes accepted positions into it), `snap` keeps the positions of
// the sweep's start for the nodes that are not resolved yet.
__host__ __device__ constexpr int cp_record_width(int D) { return D + 1 <= 4 ? 4 : (D + 1 <= 8 ? 8 : 12); }
template <int D>
__global__ __launch_bounds__(256) void k_ccpipe_pack(ChainView c, CcPipeBuf pb) {
    constexpr int RW = cp_record_width(D);
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= (long)c.T * c.N) return;
    double rec[RW];
#pragma unroll
    for (int d = 0; d < RW; ++d) rec[d] = 0.0;
#pragma unroll
    for (int d = 0; d < D; ++d) rec[d] = c.X[q * D + d];
    rec[D] = 1.0 / c.radii[q % c.N];
#pragma unroll
    for (int d = 0; d < RW; d += 2) {
        *(double2 *)(pb.cur + q * RW + d) = make_double2(rec[d], rec[d + 1]);
        *(double2 *)(pb.snap + q * RW + d) = make_double2(rec[d], rec[d + 1]);
    }
}

// a node's term lists: kinds 0 in-edge, 1 out-edge, 2 in-control, 3 out-control
struct CcNode {
    size_t node;
    int in_deg, out_deg, nci, nco, total;
    double adj_in, adj_out;
};
// log-ratio contribution of one gathered term of node k when k moves x0 -> x1, its partner
// at xn (directed_likelihoods_fast.pyx:107-180):
//   edges    (eta1 - eta0) - [softplus(eta1) - softplus(eta0)]
//   controls - adj [softplus(eta1) - softplus(eta0)],  adj = (N - deg - 1) / n_controls
// With the lean forms of device_common.hpp: eta = B - d a with B = b_in + b_out and
// a = b_in / r + b_out / r' from reciprocals (irj = 1 / r_k is the caller's, ire = 1 / r_e comes with
// the partner's record), the lean root, the table exponential and one reciprocal instead of the
// division: a third of the instructions of the plain expressions, equal to rounding.
template <int D>
__device__ __forceinline__ double cc_term_delta_fast(const double *xn, const double *xk0,
                                                     const double *xk1, bool self, bool in_dir,
                                                     bool edge, double wsp, double bin, double bout,
                                                     double irj, double ire, int squared, const double *etab) {
    const double d0 = self ? 0.0 : dist_fast<D>(xn, xk0, squared);
    const double d1 = self ? 0.0 : dist_fast<D>(xn, xk1, squared);
    const double a = in_dir ? fma(bin, irj, bout * ire) : fma(bin, ire, bout * irj);
    const double B = bin + bout;
    const double e0 = fma(-d0, a, B), e1 = fma(-d1, a, B);
    // (the 256-entry table exponential of device_common.hpp: 13 instructions + an LDS read for 18;
    // the clamp keeps the table's index inside its range wherever a chain could go, and e^x is 0 in
    // double below -745 anyway.  Its fill - a trip to memory and a workgroup barrier in front of the
    // evaluators' first loads - costs less than the ten instructions per term: +1.7 % on config 4)
    const double sp = fast_log((1.0 + tab_exp(fmax(e1, -1000.0), etab)) *
                               fast_rcp(1.0 + tab_exp(fmax(e0, -1000.0), etab)));
    return (edge ? (e1 - e0) : 0.0) - wsp * sp;
}

// per-wavefront scratch of the evaluator: the window terms of a node, compacted in term order
// (128 window terms before a flush; 64 at d >= 6, where sixteen wavefronts' proposals would not fit the LDS)
template <int D> struct CcWcap { static constexpr int N = D <= 5 ? 128 : 64; };
template <int D>
struct CcWin {
    static constexpr int WCAP = CcWcap<D>::N;
    int e[WCAP];            // partner | kind << 28
    double contrib[WCAP];   // the term with the partner at its snapshot position
    double re[WCAP];        // 1 / the partner's radius
    double xp[WCAP][D];     // the partner's proposal (requested with its record: round 5)
};

// One wavefront: node k of batch `be` in slice t, its terms four 64-term chunks at a time with
// every load of a level issued together (list indices, then the partners' records): the item
// is a chain of gather latencies, and a wavefront that owns the whole node needs neither a second
// pass over the indices nor a barrier to place its entries in the node's two lists.  The terms
// whose partner sits in the window (~15 % at C4) are collected in LDS and get their second
// evaluation - the partner at its proposal - together, a full wavefront at a time, instead of one
// mostly idle evaluation per chunk.
// (round 6: the item is the batch's `rank`-th row - the rows of a batch are stored by descending term count,
// cc_rows.hpp - and learns its node from the row's header, which arrives with the counts and the first indices)
template <int D>
__device__ __forceinline__ void ccpipe_eval_item(const ChainView &c, const CcPipeBuf &pb, int be,
                                                 int t, int rank, int lane, CcWin<D> &sw, const double *etab) {
    constexpr int PW = 2 * D + 2;
    constexpr int RW = cp_record_width(D);
    constexpr int NCH = D <= 2 ? 4 : 2;      // 64-term chunks in flight (d = 3, 4: two - their records and proposals would spill)
    const int N = c.N;
    const int j0 = be * CP_B;
    const int jprev = max(0, j0 - CP_B);       // nodes >= jprev: snapshot positions
    const int bb = be & 1;
    // the node's row: counts and the first 64 * NCH indices leave together
    // (every gather of the item is a 32-bit lane offset from a wave-uniform base: 64-bit lane addresses cost
    // three vector instructions and two registers each, and the compiler hoisted and spilled the row's)
    const char *row = (const char *)(pb.terms + ((size_t)t * N + j0 + rank) * pb.tw);
    const int4 hdr = *(const int4 *)row;
    const double2 adj = *(const double2 *)(row + 16);
    // (every lane's copy: as a scalar it cost the kernel its last free scalar registers - 36 bytes of scratch)
    const int jk = *(const int32_t *)(row + 32);       // the row's node
    const int k = jk - j0;
    // (the lane as the row's offsets see it, opaque per item: the compiler otherwise hoists the four offsets
    // out of the workgroup's item loop - a wavefront runs one item, rarely two - and spills them)
    int lane_r = lane;
    asm volatile("" : "+v"(lane_r));
    constexpr uint32_t HB = CP_HDR * sizeof(int32_t);
    int e_first[NCH];
#pragma unroll
    for (int u = 0; u < NCH; ++u)
        e_first[u] = *(const int32_t *)(row + (HB + 4u * (uint32_t)min(64 * u + lane_r, pb.tw - CP_HDR - 1)));
    CcNode nd;
    nd.node = (size_t)t * N + jk;
    nd.in_deg = hdr.x; nd.out_deg = hdr.y; nd.nci = hdr.z; nd.nco = hdr.w;
    nd.adj_in = adj.x; nd.adj_out = adj.y;
    nd.total = nd.in_deg + nd.out_deg + nd.nci + nd.nco;
    // (the row's order: out-edges, out-controls, in-edges, in-controls - cc_rows.hpp)
    const int k_oc = nd.out_deg, k_ie = nd.out_deg + nd.nco, k_ic = nd.out_deg + nd.nco + nd.in_deg;
    const double *cur = pb.cur + (size_t)t * N * RW, *snap = pb.snap + (size_t)t * N * RW;
    const uint32_t snap_off = (uint32_t)((const char *)snap - (const char *)cur);
    const double *props = pb.prop + (size_t)t * N * PW;
    double xk0[D], xk1[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        xk0[d] = props[(size_t)jk * PW + D + 2 + d];
        xk1[d] = props[(size_t)jk * PW + d];
    }
    const double bin = c.intercept[0], bout = c.intercept[1];
    const double irj = snap[(size_t)jk * RW + D];          // 1 / r_k (k_ccpipe_pack)
    const size_t lbase = ((size_t)bb * c.T + t) * pb.cap * CP_B + k;      // + entry * CP_B
    const unsigned long long below = (1ull << lane) - 1ull;
    double acc = 0.0;
    int bx = 0, bo = 0, wcnt = 0;
    // second evaluation of the collected window terms: lane i takes the i-th of them
    auto flush = [&]() {
        __builtin_amdgcn_wave_barrier();
        for (int i0 = 0; i0 < wcnt; i0 += 64) {
            const int i = i0 + lane;
            const bool live = i < wcnt;
            const int packed = live ? sw.e[i] : 0;
            const int e = packed & 0x0FFFFFFF, kind = (packed >> 28) & 3;
            double h = 0.0;
            if (live) {
                double xe1[D];
#pragma unroll
                for (int d = 0; d < D; ++d) xe1[d] = sw.xp[i][d];
                const bool in_dir = (kind == 0 || kind == 2);
                const double wsp = kind < 2 ? 1.0 : (kind == 2 ? nd.adj_in : nd.adj_out);
                h = cc_term_delta_fast<D>(xe1, xk0, xk1, false, in_dir, kind < 2, wsp, bin, bout, irj,
                                          sw.re[i], c.squared, etab) - sw.contrib[i];
            }
            const bool isx = live && e < j0, iso = live && e >= j0;
            const unsigned long long mx = __ballot(isx), mo = __ballot(iso);
            if (isx) {
                const size_t p = lbase + (size_t)(bx + __popcll(mx & below)) * CP_B;
                pb.xidx[p] = e - jprev; pb.xval[p] = h;
            }
            if (iso) {
                const size_t p = lbase + (size_t)(bo + __popcll(mo & below)) * CP_B;
                pb.oidx[p] = e - j0; pb.oval[p] = h;
            }
            bx += __popcll(mx); bo += __popcll(mo);
        }
        wcnt = 0;
        __builtin_amdgcn_wave_barrier();
    };
    for (int q0 = 0; q0 < nd.total; q0 += 64 * NCH) {
        int e[NCH], kind[NCH];
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int q = q0 + 64 * u + lane;
            kind[u] = q < k_oc ? 1 : (q < k_ie ? 3 : (q < k_ic ? 0 : 2));
            const int er = q0 == 0 ? e_first[u]
                                   : *(const int32_t *)(row + (HB + 4u * (uint32_t)min(q, pb.tw - CP_HDR - 1)));
            e[u] = q < nd.total ? er : -1;
        }
        double xe[NCH][D], re[NCH], xp[NCH][D];
        bool win[NCH];
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            const int ee = max(e[u], 0);
            // (`snap` lies behind `cur` in one allocation: the choice is part of the lane's offset)
            const double *src = (const double *)((const char *)cur + (__umul24((uint32_t)ee, (uint32_t)(RW * sizeof(double))) +
                                                                     (ee < jprev ? 0u : snap_off)));
#pragma unroll
            for (int d = 0; d < D; ++d) xe[u][d] = src[d];
            re[u] = src[D];
            // a partner inside the window gets its second evaluation at its proposal: requested now,
            // with the record, instead of one round trip later (flush reads it from LDS)
            win[u] = e[u] >= jprev && e[u] < jk;
            const double *prow = (const double *)((const char *)props + __umul24((uint32_t)ee, (uint32_t)(PW * sizeof(double))));
#pragma unroll
            for (int d = 0; d < D; ++d) xp[u][d] = win[u] ? prow[d] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < NCH; ++u) {
            if (q0 + 64 * u >= nd.total) break;             // wave-uniform
            double contrib = 0.0;
            if (e[u] >= 0) {
                const bool in_dir = (kind[u] == 0 || kind[u] == 2);
                const double wsp = kind[u] < 2 ? 1.0 : (kind[u] == 2 ? nd.adj_in : nd.adj_out);
                contrib = cc_term_delta_fast<D>(xe[u], xk0, xk1, e[u] == jk, in_dir, kind[u] < 2, wsp,
                                                bin, bout, irj, re[u], c.squared, etab);
                acc += contrib;
            }
            const unsigned long long mw = __ballot(win[u]);   // its acceptance changes this term
            const int nw = __popcll(mw);
            if (wcnt + nw > CcWin<D>::WCAP) flush();
            if (win[u]) {
                const int pos = wcnt + __popcll(mw & below);
                sw.e[pos] = e[u] | (kind[u] << 28);
                sw.contrib[pos] = contrib;
                sw.re[pos] = re[u];
#pragma unroll
                for (int d = 0; d < D; ++d) sw.xp[pos][d] = xp[u][d];
            }
            wcnt += nw;
        }
    }
    flush();
    const double total = wave_sum_all(acc);
    if (lane == 0) {
        const size_t slot = ((size_t)bb * c.T + t) * CP_B + k;
        pb.tot[slot] = total;
        pb.cnt[slot * 2] = bx; pb.cnt[slot * 2 + 1] = bo;
    }
}

// The cross sums of a batch (its nodes' corrections for the previous batch's acceptances, final by now): the
// first CP_XCH entries of every node requested in one go, whatever the counts will turn out to be; a list
// longer than that comes sixteen entries per trip.  In list order, by the node's thread alone: the same sum
// whoever computes it - the slice's helper workgroup or, without helpers, the upper half of its resolver.
// (macros: the loads stay in flight across the barrier that publishes sPrev between the two)
#define cc_cross_loads \
        int xm[CP_XCH];                                                                                 \
        double xh[CP_XCH];                                                                              \
_Pragma("unroll")                                                                                    \
        for (int u = 0; u < CP_XCH; ++u) {                                                              \
            const size_t p = lbase + (size_t)min(u, pb.cap - 1) * CP_B;                              \
            xm[u] = pb.xidx[p];                                                                      \
            xh[u] = pb.xval[p];                                                                      \
        }
#define cc_cross_sums \
        double xs = 0.0;                                                                             \
_Pragma("unroll")                                                                                    \
        for (int u = 0; u < CP_XCH; ++u)                                                                \
            if (u < ncx && ((sPrev[xm[u] >> 6] >> (xm[u] & 63)) & 1ull)) xs += xh[u];                \
        for (int e0 = CP_XCH; e0 < ncx; e0 += 16) {                                                     \
            int m[16];                                                                               \
            double h[16];                                                                            \
_Pragma("unroll")                                                                                    \
            for (int u = 0; u < 16; ++u) {                                                           \
                const size_t p = lbase + (size_t)min(e0 + u, ncx - 1) * CP_B;                        \
                m[u] = pb.xidx[p];                                                                   \
                h[u] = pb.xval[p];                                                                   \
            }                                                                                        \
_Pragma("unroll")                                                                                    \
            for (int u = 0; u < 16; ++u)                                                             \
                if (e0 + u < ncx && ((sPrev[m[u] >> 6] >> (m[u] & 63)) & 1ull)) xs += h[u];          \
        }

// Resolve batch b of slice t: thread k owns node k of the batch.
#ifdef DLSM_PIPE_TIMING
// resolver phase stamps and evaluator entry / exit of the last sweep (profiles/ccpipe_timing.py)
__device__ unsigned long long g_cc_res_t[32][16][8];
__device__ unsigned long long g_cc_item_t[32][4096][2];
#define DLSM_CC_STAMP(I_, DEP_) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(DEP_)); cts[I_] = t_; }
#else
#define DLSM_CC_STAMP(I_, DEP_)
#endif
// One wavefront's arrival at its workgroup's barrier, as the instruction (see ccpipe_resolve): the
// "memory" clobber keeps the compiler from moving LDS / global accesses across it; loads already
// requested stay in flight (gfx950 needs no drained counters at s_barrier).
__device__ __forceinline__ void cc_barrier_arrive() { asm volatile("s_barrier" ::: "memory"); }
// ... with this wavefront's LDS stores completed first (what the other side reads behind the barrier)
__device__ __forceinline__ void cc_barrier_arrive_after_lds_stores() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int D>
__device__ __forceinline__ void ccpipe_resolve(const ChainView &c, const CcPipeBuf &pb, int b, int t,
                                               unsigned long long (*sMask)[CP_WAVES],
                                               unsigned long long *sPrev, int *sChanged,
                                               double *sCross, double (*sOv)[CP_B], int (*sOi)[CP_B]
#ifdef DLSM_PIPE_TIMING
                                               , int tl
#endif
                                               ) {
    constexpr int PW = 2 * D + 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef DLSM_PIPE_TIMING
    unsigned long long cts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    DLSM_CC_STAMP(0, (double)tid)
    const int N = c.N;
    const int j0 = b * CP_B;
    const int nb = min(CP_B, N - j0);
    const int bb = b & 1;
    // Two threads per node, half a workgroup apart: thread k carries the node's state and its
    // own entries, thread CP_B + k its cross entries - so that everything a node needs from
    // memory leaves in ONE go without either thread holding both lists in registers
    // (profiles/ccpipe_timing.py: the resolver is the launch's critical path, and was four
    // memory round trips of 3 - 4 us in sequence - state, cross entries, own entries, each behind
    // a count).  The first CP_XCH cross / CP_OWN_REGS own entries are requested whatever the counts
    // will turn out to be (the lists are entry-major with room for `cap` entries; what lies
    // beyond a count is ignored).
    static_assert(CP_THREADS == 2 * CP_B, "two threads per node");
    const bool upper = tid >= CP_B;
    const int k = upper ? tid - CP_B : tid;
    const bool valid = !upper && k < nb;
    const int kc = min(k, nb - 1);
    unsigned long long *accg = pb.accmask + (size_t)t * CP_WAVES;
    if (tid < CP_WAVES) sPrev[tid] = b > 0 ? accg[tid] : 0ull;
    const size_t slot = ((size_t)bb * c.T + t) * CP_B + kc;
    const size_t lbase = ((size_t)bb * c.T + t) * pb.cap * CP_B + kc;
    // Cross entries up front: 24 (a node of config 4 has 12 on average, more than 16 in one of
    // eight cases, and a second trip to memory for the few costs every wavefront 3.6 us); the two
    // halves are separate code paths, so neither holds the other's registers.
    const int nlist = pb.cnt[slot * 2 + (upper ? 0 : 1)];
    const int ncx = upper ? nlist : 0, nown = upper ? 0 : nlist;
    int oi[CP_OWN_REGS];
    double ov[CP_OWN_REGS];
    double r = 0.0, lu = 0.0, st = 0.0, x1[D];
    int32_t na = 0, ns = 0, un = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) x1[d] = 0.0;
#pragma unroll
    for (int e = 0; e < CP_OWN_REGS; ++e) { oi[e] = 0; ov[e] = 0.0; }
    if (upper && pb.helpers) {
        // the slice's helper workgroup (another CU) has summed the cross entries: wait for its announcement,
        // take the sums past the L1
        cc_barrier_arrive();                           // (the lower half's barrier: it waits for sPrev's store)
        // (the sums announce themselves: a slot holds CC_XS_EMPTY until its sum is stored - one trip to the L2 less
        // than a flag and then the sums, 0.7 us on the launch's critical path)
        const unsigned long long *slotp = (const unsigned long long *)&pb.xsum[(size_t)t * CP_B + k];
        unsigned long long got = CC_XS_EMPTY;
        for (int n = 0; n < pb.budget; ++n) {
            got = __hip_atomic_load(slotp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__ballot(got == CC_XS_EMPTY) == 0ull) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (__ballot(got == CC_XS_EMPTY) != 0ull && lane == 0)
            __hip_atomic_fetch_or(pb.err, CC_ERR_HELPER, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        sCross[k] = __longlong_as_double((long long)got);
        // taken: empty again for the next launch (a plain store: the kernel boundary publishes it; the helper
        // stores each slot once per launch, before this)
        pb.xsum[(size_t)t * CP_B + k] = __longlong_as_double((long long)CC_XS_EMPTY);
    } else if (upper) {
        cc_cross_loads
        // Each half of the workgroup meets the other at its OWN s_barrier: the halves are whole
        // wavefronts - `upper` is wave-uniform - and the hardware counts a workgroup's ARRIVALS,
        // whatever the instruction address (gfx950 ISA: s_barrier).  That is a contract of the
        // instruction set, not of HIP's __syncthreads() (undefined in divergent code), so the two
        // arrivals are written as what they are: the instruction itself behind exactly the wait
        // its side needs (cc_barrier_*: a compiler barrier too).  This side has stored nothing
        // the other reads and keeps its 48 loads in flight across the barrier.
        // (One barrier behind the if / else was built: both halves' operands are then live at one
        // program point, 119 -> 128 VGPRs and 14 spilled, on the kernel's critical path.)
        cc_barrier_arrive();                           // sPrev visible
        cc_cross_sums
        sCross[k] = xs;
    } else {
#pragma unroll
        for (int e = 0; e < CP_OWN_REGS; ++e) {
            const size_t p = lbase + (size_t)min(e, pb.cap - 1) * CP_B;
            oi[e] = pb.oidx[p];
            ov[e] = pb.oval[p];
        }
        int oi2[CP_OWN_LDS];
        double ov2[CP_OWN_LDS];
#pragma unroll
        for (int e = 0; e < CP_OWN_LDS; ++e) {
            const size_t p = lbase + (size_t)min(CP_OWN_REGS + e, pb.cap - 1) * CP_B;
            oi2[e] = pb.oidx[p];
            ov2[e] = pb.oval[p];
        }
        r = pb.tot[slot];
        const double *pr = pb.prop + ((size_t)t * N + j0 + kc) * PW;
        double x0[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { x1[d] = pr[d]; x0[d] = pr[D + 2 + d]; }
        // prior terms of the step's logp closure, with the neighbouring slices as they are now
        // (the odd slices run one batch behind: kernels_spec_pipe.hpp)
        r += node_log_prior<D>(c, t, j0 + kc, x1) - node_log_prior<D>(c, t, j0 + kc, x0);
        lu = log(pr[D]);
        const size_t tjc = (size_t)t * N + j0 + kc;
        st = c.step[tjc];
        na = c.nacc[tjc]; ns = c.nsteps[tjc]; un = c.until[tjc];
#pragma unroll
        for (int e = 0; e < CP_OWN_LDS; ++e) {          // (this thread's own column: read back by itself only)
            sOi[e][k] = CP_OWN_REGS + e < nown ? oi2[e] : 0;
            sOv[e][k] = ov2[e];
        }
        cc_barrier_arrive_after_lds_stores();          // (the upper half's barrier; sPrev is this half's store)
    }
    DLSM_CC_STAMP(2, r)
    // the node's first own entries stay in registers through the passes
#pragma unroll
    for (int e = 0; e < CP_OWN_REGS; ++e) oi[e] = e < nown ? oi[e] : 0;
    __syncthreads();                                   // sCross visible
    if (!upper) r += sCross[k];
    // Fixed point of a -> F(a) WITHOUT workgroup barriers (round 5).  The eight wavefronts that own
    // the batch's nodes iterate on their own: a pass reads the other wavefronts' mask words as they
    // are at that moment (whatever a neighbour has already corrected is used at once, Gauss-Seidel
    // fashion, instead of one barrier later), re-ballots, and publishes its word only when it
    // changed - word first, then the version counter sCtl[0] (release).  A wavefront whose last pass
    // changed nothing and was based on the version it still reads stands at that version: it files the
    // version in sCtl[1 + w] and polls (one LDS read) instead of recomputing.  The system is solved
    // when all eight have filed the CURRENT version v: a wavefront that has filed v recomputes only
    // after the version has moved, a wavefront that changes its word at v never files v, so eight
    // filings of v mean no word changed while the version was v, and every pass behind a filing read
    // exactly those words (a word's store precedes the increment that ends the version it was
    // computed at) - they are a fixed point, and the fixed point of the triangular system is the
    // sequential scan's result.  (The barrier form cost ~10 passes x 0.45 us for every wavefront.)
    unsigned long long *sW = sMask[0];
    uint32_t *sW32 = (uint32_t *)sMask[0];              // (bit b of word w = bit b & 31 of half-word 2 w + (b >> 5))
    int *sCtl = sChanged;                               // [0] version, [1 + w] version filed by wavefront w
    constexpr int NRW = CP_B / 64;                      // wavefronts that own nodes
    // (relaxed workgroup-scope atomics, not `volatile`: a volatile access is followed by a full wait in
    // this compiler, which made a pass fourteen LDS round trips one behind the other - 0.9 us; the
    // atomics are re-read every pass like volatile ones and their waits are batched)
#define CC_LD64(P_) __hip_atomic_load((P_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define CC_LD32(P_) __hip_atomic_load((P_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
    unsigned long long mine = __ballot(valid && !(lu >= r));
    if (!upper && lane == 0) { sW[wave] = mine; sCtl[1 + wave] = -1; }
    if (tid == 0) sCtl[0] = 0;
    __syncthreads();
    DLSM_CC_STAMP(3, ov[0])
    if (!upper) {
        int computed_v = -2, filed_v = -2;                  // version my last pass was based on / I have filed
        bool quiet = false;                                 // ... and that pass changed nothing
        bool settled = false;
        for (int spin = 0; spin < (1 << 22); ++spin) {
            // (acquire: the version word orders the mask words read behind it against the writers' release
            // fetch_add - on LDS it costs nothing measurable, and the termination argument no longer leans on
            // in-order LDS issue and the compiler keeping monotonic loads in place: round-5 advice)
            const int v = __hip_atomic_load(&sCtl[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (quiet && v == computed_v) {
                // nothing has been published since the version read in front of my last pass, and that pass
                // changed nothing: my word stands at version v.  File it (once) and poll - a poll is one LDS
                // round trip and leaves the SIMD's issue slots to the wavefronts that still compute.
                // (Requesting the pass's words with every poll - one round trip less per link of a chain of
                // flips - measured 0.4 % slower.)
                if (filed_v != v) {
                    if (lane == 0) __hip_atomic_store(&sCtl[1 + wave], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    filed_v = v;
                }
                const int filed = CC_LD32(&sCtl[1 + (lane & (NRW - 1))]);   // lane w: the version wavefront w has filed
                if (__ballot(filed != v) == 0ull) { settled = true; break; }    // every wavefront stands at the current version
                continue;
            }
            computed_v = v;                                 // (read BEFORE the words of this pass)
            uint32_t wd[CP_OWN_REGS];                       // every word of the pass requested before the first use
#pragma unroll
            for (int e = 0; e < CP_OWN_REGS; ++e) wd[e] = CC_LD32(&sW32[oi[e] >> 5]);
            double tm[CP_OWN_REGS];
#pragma unroll
            for (int e = 0; e < CP_OWN_REGS; ++e)
                tm[e] = (e < nown && ((wd[e] >> (oi[e] & 31)) & 1u)) ? ov[e] : 0.0;
            static_assert(CP_OWN_REGS == 12, "the sum below is a tree over twelve terms");
            double s_own = (((tm[0] + tm[1]) + (tm[2] + tm[3])) + ((tm[4] + tm[5]) + (tm[6] + tm[7]))) +
                           ((tm[8] + tm[9]) + (tm[10] + tm[11]));
            if (nown > CP_OWN_REGS) {                           // the next entries: from this thread's LDS column
                int m2[CP_OWN_LDS];
                uint32_t w2[CP_OWN_LDS];
                double h2[CP_OWN_LDS];
#pragma unroll
                for (int e = 0; e < CP_OWN_LDS; ++e) { m2[e] = sOi[e][k]; h2[e] = sOv[e][k]; }
#pragma unroll
                for (int e = 0; e < CP_OWN_LDS; ++e) w2[e] = CC_LD32(&sW32[m2[e] >> 5]);
#pragma unroll
                for (int e = 0; e < CP_OWN_LDS; ++e)
                    s_own += (CP_OWN_REGS + e < nown && ((w2[e] >> (m2[e] & 31)) & 1u)) ? h2[e] : 0.0;
            }
            for (int e0 = CP_OWN_REGS + CP_OWN_LDS; e0 < nown; e0 += 8) {   // the long list: from memory, eight
                int m[8];                                       // loads in flight per trip
                double h[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const size_t p = lbase + (size_t)min(e0 + u, nown - 1) * CP_B;
                    m[u] = pb.oidx[p];
                    h[u] = pb.oval[p];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + u < nown && ((CC_LD64(&sW[m[u] >> 6]) >> (m[u] & 63)) & 1ull)) s_own += h[u];
            }
            const unsigned long long g = __ballot(valid && !(lu >= r + s_own));
            quiet = g == mine;
            if (!quiet) {
                mine = g;
                if (lane == 0) {
                    __hip_atomic_store(&sW[wave], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    // the word before the version (release)
                    __hip_atomic_fetch_add(&sCtl[0], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
#ifdef DLSM_PIPE_TIMING
            cts[6] += 1;
#endif
        }
        // the spin bound ran out: the mask a wavefront holds is NOT the fixed point - say so.  (Through a pointer
        // held in device memory, read here only: the launch's own error word, pb.err, kept alive across this loop
        // cost the kernel its last scalar registers - 36 bytes of scratch in every instantiation.)
        if (!settled && lane == 0) {
            int32_t *w = g_cc_fixpoint_err;
            if (w) __hip_atomic_fetch_or(w, CC_ERR_FIXPOINT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
#undef CC_LD64
#undef CC_LD32
    DLSM_CC_STAMP(4, (double)lane)
    const int accepted = (int)((mine >> lane) & 1ull);
    if (valid) {
        const size_t tj = (size_t)t * N + j0 + k;
        if (accepted) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                c.X[tj * D + d] = x1[d];
                pb.cur[tj * cp_record_width(D) + d] = x1[d];
            }
        }
        metropolis_bookkeeping(st, na, ns, un, c.tune, c.tune_interval, accepted);
        c.step[tj] = st; c.nacc[tj] = na; c.nsteps[tj] = ns; c.until[tj] = un;
    }
    if (!upper && lane == 0) accg[wave] = mine;
#ifdef DLSM_PIPE_TIMING
    DLSM_CC_STAMP(5, (double)lane)
    if (tid == 0 && tl >= 0 && tl < 32 && t < 16) for (int i = 0; i < 7; ++i) if (i != 1) g_cc_res_t[tl][t][i] = cts[i];   // ([1], [7]: the helper's)
#endif
}

// The helper of slice t's resolver: the cross sums of batch b (cc_cross_loads / _sums above) on a CU of its own.
// The resolver's first trip to memory was 250 KB through ONE CU's vector L1 - 147 KB of them the cross entries -
// and lasted 4.7 us, with the cross sums' barrier at 7 us (profiles/r05_ccpipe_timing.json); a build without
// the cross entries ran the launch in 10.7 instead of 12.55 us.  The helper takes those 147 KB through another
// CU's L1 and stores the 512 sums past its L1 into slots that hold CC_XS_EMPTY; the resolver's upper half polls
// its nodes' slots past ITS L1 until none is empty, and empties them again behind its read (device_common.hpp:
// the sc1 / sc1 hand-off, with the datum as its own flag).  A launch whose resolver runs has its helper run; the
// sweep's first launch, which resolves nothing, empties every slot.
template <int D>
__device__ __forceinline__ void ccpipe_cross_helper(const ChainView &c, const CcPipeBuf &pb, int b, int t,
                                                    unsigned long long *sPrev
#ifdef DLSM_PIPE_TIMING
                                                    , int tl
#endif
                                                    ) {
    const int tid = threadIdx.x;
#ifdef DLSM_PIPE_TIMING
    unsigned long long cts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    DLSM_CC_STAMP(1, (double)tid)
#endif
    const int nb = min(CP_B, c.N - b * CP_B);
    const int bb = b & 1;
    if (tid < CP_WAVES) sPrev[tid] = b > 0 ? pb.accmask[(size_t)t * CP_WAVES + tid] : 0ull;
    if (tid < CP_B) {                                   // (whole wavefronts: see the resolver's note on s_barrier)
        const int k = tid, kc = min(k, nb - 1);
        const size_t slot = ((size_t)bb * c.T + t) * CP_B + kc;
        const size_t lbase = ((size_t)bb * c.T + t) * pb.cap * CP_B + kc;
        const int ncx = pb.cnt[slot * 2];
        cc_cross_loads
        cc_barrier_arrive_after_lds_stores();           // sPrev (wavefront 0's store) visible; the 48 loads stay in flight
        cc_cross_sums
        coh_store<true>(&pb.xsum[(size_t)t * CP_B + k], xs);       // (an 8-byte store past the L1: the announcement)
#ifdef DLSM_PIPE_TIMING
        DLSM_CC_STAMP(7, xs)
        if (tid == 0 && tl >= 0 && tl < 32 && t < 16) { g_cc_res_t[tl][t][1] = cts[1]; g_cc_res_t[tl][t][7] = cts[7]; }
#endif
    } else {                                            // (the workgroup's other eight wavefronts: an arrival only)
        cc_barrier_arrive();
    }
}

// Launch l: even slices resolve batch l and evaluate batch l + 1; odd slices resolve batch
// l - 1 and evaluate batch l.  Workgroups [0, T) resolve, [T, 2 T) are their helpers (pb.helpers), the rest evaluate.
template <int D>
__global__ __launch_bounds__(CP_THREADS) void k_ccpipe_step(ChainView c, CcPipeBuf pb, int l) {
    __shared__ unsigned long long sMask[2][CP_WAVES];
    __shared__ unsigned long long sPrev[CP_WAVES];
    __shared__ int sChanged[1 + CP_B / 64];
    __shared__ double sCross[CP_B];
    __shared__ double sOv[CP_OWN_LDS][CP_B];
    __shared__ int sOi[CP_OWN_LDS][CP_B];
    __shared__ CcWin<D> sWin[CP_WAVES];
    const int T = c.T;
    if ((int)blockIdx.x < T) {
        const int t = blockIdx.x;
        const int b = l - (t & 1);
        // (the sweep's first launch resolves nothing: it empties the helpers' hand-over slots)
        if (l < 0 && pb.helpers && (int)threadIdx.x < CP_B)
            pb.xsum[(size_t)t * CP_B + threadIdx.x] = __longlong_as_double((long long)CC_XS_EMPTY);
        if (b >= 0 && b < pb.nbat) ccpipe_resolve<D>(c, pb, b, t, sMask, sPrev, sChanged, sCross, sOv, sOi
#ifdef DLSM_PIPE_TIMING
                                                     , l + 1
#endif
                                                     );
        return;
    }
    const int nres = pb.helpers ? 2 * T : T;            // workgroups in front of the evaluators
    if ((int)blockIdx.x < nres) {
        const int t = (int)blockIdx.x - T;
        const int b = l - (t & 1);
        if (b >= 0 && b < pb.nbat) ccpipe_cross_helper<D>(c, pb, b, t, sPrev
#ifdef DLSM_PIPE_TIMING
                                                           , l + 1
#endif
                                                           );
        return;
    }
    __shared__ double sTab[EXPTAB_N];                   // 2^(j / 256): the evaluators' exponential
    const int lane = threadIdx.x & 63;
    const int nE = (T + 1) / 2, nO = T / 2;
    const int beE = l + 1, beO = l;
    const int nbE = (beE >= 0 && beE < pb.nbat) ? min(CP_B, c.N - beE * CP_B) : 0;
    const int nbO = (beO >= 0 && beO < pb.nbat) ? min(CP_B, c.N - beO * CP_B) : 0;
    // items rank-major: q -> (rank q / slices, slice q mod slices) - ascending q is descending term count, so the
    // longest items take the wavefront slots 0 .. 3 of the workgroups (one per SIMD, first to start) and the
    // shortest the slots that give a SIMD its third item
    const int nslE = nbE > 0 ? nE : 0, nslO = nbO > 0 ? nO : 0, nsl = max(nslE + nslO, 1);
    const int nodes = nsl * max(nbE, nbO);
    const float inv_nsl = __builtin_amdgcn_rcpf((float)nsl);
    const int n_wg = (int)gridDim.x - nres, wg = (int)blockIdx.x - nres;
    const int nwaves = n_wg * CP_WAVES;
    // item q -> wavefront (q / workgroups) of workgroup (q % workgroups): a launch's items are spread over
    // every evaluator CU (2560 items on 246 CUs: 10 or 11 wavefronts each, 2 - 3 per SIMD) instead of filling
    // 160 CUs with four wavefronts per SIMD - the item is ~900 float64 vector instructions and four of them
    // on one SIMD take turns issuing (round 5: profiles/r05_ccpipe_timing.json)
    const int gw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) * n_wg + wg);
    // (Built and dropped, round 5: the workgroup's wavefronts without an item - five or six of sixteen - requesting
    // the term rows of the NEXT launch's items into this XCD's L2: 2475 it/s against 2533 without; the first item's
    // row head requested before the table's barrier: 2483 against 2519, three spilled registers.)
    exp_table_fill(sTab, threadIdx.x);
    __syncthreads();
    for (int q = gw; q < nodes; q += nwaves) {
        const int k = (int)(((float)q + 0.5f) * inv_nsl);          // q / nsl (q < 2^20): the rank inside the batch
        const int si = q - k * nsl;
        const bool odd = si >= nslE;
        if (k >= (odd ? nbO : nbE)) continue;                        // (a ragged last batch beside a full one)
        const int t = odd ? 2 * (si - nslE) + 1 : 2 * si;
#ifdef DLSM_PIPE_TIMING
        unsigned long long cts[2];
        DLSM_CC_STAMP(0, (double)lane)
#endif
        ccpipe_eval_item<D>(c, pb, odd ? beO : beE, t, k, lane, sWin[threadIdx.x >> 6], sTab);
#ifdef DLSM_PIPE_TIMING
        DLSM_CC_STAMP(1, (double)lane)
        if (lane == 0 && l + 1 >= 0 && l + 1 < 32 && gw < 4096) { g_cc_item_t[l + 1][gw][0] = cts[0]; g_cc_item_t[l + 1][gw][1] = cts[1]; }
#endif
    }
}

}  // namespace dlsm
