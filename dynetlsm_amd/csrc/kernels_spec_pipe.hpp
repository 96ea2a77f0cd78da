// Pipelined speculative sweep (algo 4): the speculative-batch scan of
// kernels_spec_sweep.hpp with the evaluation of batch b + 1 moved off the
// critical path.
//
// In the two-kernel form eval(b + 1) reads the positions resolve(b) has just
// written, so the launches alternate strictly.  Two facts remove that edge:
//   * a node moves at most once per parity, so "the position of node i before its
//     own step" is its position at the start of the parity: a snapshot taken by the
//     propose kernel is valid for every node that is not resolved yet, however far
//     the resolver has got;
//   * the effect of batch b's acceptances on node k of batch b + 1 is the same
//     H[k][m] correction that already couples the nodes inside a batch.
// So eval(b + 1) uses X for nodes of batches < b (final), the snapshot for all
// others, and emits the cross block Hx[k][m], m in batch b, next to its own diagonal
// block; resolve(b + 1) adds the Hx rows of the nodes batch b accepted.  eval(b + 1)
// then depends on resolve(b - 1) only, and ONE launch carries both roles:
//
//     launch b:  workgroups [0, nsl)   resolve batch b of their slice
//                workgroups [nsl, ..)  evaluate batch b + 1 (one item per wavefront)
//
// No workgroup of a launch reads what another one of the same launch writes; the
// kernel boundary is the only synchronisation.  Algebraically this is still the
// sequential Gauss-Seidel scan (differences are rounding only).
//
// Both parities run in the same launches.  Slices of different parity interact only
// through the prior term of node (t, j), which looks at X[t +- 1][j]; if the odd
// slices run ONE BATCH BEHIND the even ones, an even node of batch l still sees the
// odd slices' node j untouched, and an odd node of batch l - 1 sees the even slices'
// node j final - exactly the even-then-odd order of the reference.  Launch l
// therefore resolves batch l of the even slices and batch l - 1 of the odd ones and
// evaluates batches l + 1 / l; the prior delta is computed by the resolver.
//
// The accept test runs in the multiplicative domain, u_k < exp(r_k) * prod_m H[k][m]:
// the evaluators hand over products of (1 + E e^{-d}) factors as they accumulate
// them, so a wavefront spends one division where the additive form needs a
// float64 log per node part and per H entry (each executed by all 64 lanes); the
// resolver pays one exp per node instead.  (A node whose log-ratio is beyond +-700 - exp would
// saturate - is resolved in the log domain, with the logs of its H factors: pipe_resolve.)
#pragma once
#include "kernels_spec_sweep.hpp"

namespace dlsm {

// 1: the last wavefront of a SIMD hands its first-round H entries to the first one (pipe_item_finish).  Under
// oldest-first issue this measured slower (10.77 against 10.54 us); with the items advancing together
// (DLSM_TRIP_PRIO) the first wavefront of a SIMD still leaves ~1.8 us before the last one, whose entry is the
// launch's tail: 9.82 -> 9.58 us per launch, C2 4565 -> 4660 it/s
#ifndef DLSM_H_SHIFT
#define DLSM_H_SHIFT 1
#endif
// 1: a wavefront's issue priority falls as its item advances (s_setprio 3 for the first quarter of the
// prefetched trips .. 0 for the last).  The arbiter serves a SIMD's ready wavefronts oldest first: its four
// items ran nearly one after the other (first trip done after 0.7 / 1.4 / 3.0 / 4.5 us, exit after 3.8 / 5.3 /
// 7.1 / 8.6: profiles/r04_h_entry_ablation.md), and the youngest ran its tail alone, with nobody to fill the
// slots its dependent float64 instructions leave.  With the priority tied to progress the four advance
// together and end together: k_pipe_step 10.53 -> 10.08 us per launch, C2 4300 -> 4470 it/s (boundaries at
// the quarters or one trip later: the same; two levels instead of four: half the gain); with priority 3 from
// the item's FIRST instruction (pipe_item_prologue: a wavefront that enters late gets its loads out at once)
// 9.90 us, 4570 it/s.  (The table's fill + barrier moved behind the item's operand requests, so that the two
// round trips overlap: 10.65 us - the barrier then holds all sixteen wavefronts until the last one's operands
// have arrived; dropped.)
#ifndef DLSM_TRIP_PRIO
#define DLSM_TRIP_PRIO 1
#endif
#if DLSM_TRIP_PRIO
#define DLSM_TRIP_PRIO_STEP(U_)                                                                \
        if ((U_) == 0) __builtin_amdgcn_s_setprio(3);                                          \
        else if ((U_) == (PP_NPRE * 1) / 4) __builtin_amdgcn_s_setprio(2);                     \
        else if ((U_) == (PP_NPRE * 2) / 4) __builtin_amdgcn_s_setprio(1);                     \
        else if ((U_) == (PP_NPRE * 3) / 4) __builtin_amdgcn_s_setprio(0);
#else
#define DLSM_TRIP_PRIO_STEP(U_)
#endif
constexpr int PP_THREADS = 1024;
constexpr int PP_WAVES = PP_THREADS / 64;
constexpr int PP_B = 128;               // nodes per batch (two 64-lane halves)
constexpr int PP_MAXPARTS = 8;          // (PipeLds::plan holds 8 parts)
// per (slice, batch slot) of PipeBuf::acc: [0] count, [1 .. PP_B] the accepted nodes ascending (the [m][k]
// resolver's gather list), [PP_ACC_MASK ..] the same set as two 64-bit masks (the row resolver's)
constexpr int PP_ACC = PP_B + 8;
constexpr int PP_ACC_MASK = PP_B + 4;

// What an evaluator of kernels_pipe_lds.hpp reads of the launch's arguments, in ONE piece at the head of PipeBuf's
// tail: its first 128 bytes are requested together by the wavefront's first instructions (the compiler fetches a
// kernel argument where it is first used - a dozen scalar loads, each with a wait of its own, in a prologue that
// the four wavefronts of a SIMD issue one after the other).
struct PipeLds {
    const double *X; const uint32_t *ybits; double *prop; double *full0; double *Hd; int32_t *acc; double *consts;
    double *xprod;
    int T, N, W, squared, parts, nbat, lds_cap, xserve;
    // per launch: the batch evaluated / its nodes / the active slices (even and odd slices); evaluator workgroup e
    // (blockIdx.x - T) holds 16 consecutive nodes (e mod 8) of active slice si and part p, e / 8 = p nsl + si with
    // the quotient as a multiplication: r / nsl = r nsl_magic >> 16 (nsl_magic = ceil(2^16 / nsl), r < 2^16 / nsl)
    int beE, beO, nbE, nbO, nslE, nslO, xstride; uint32_t nsl_magic;
    uint32_t plan[4][8];        // the parts' trip lists (pipe_plan_entry), by (window trips - 1, part)
};
static_assert(sizeof(PipeLds) == 256, "PipeLds: two 64-byte halves of scalars and the plan table");

// G batches are resolved (and G evaluated) per launch.  Batch b is evaluated while the batches
// from ws(b) = G (b / G - 1) on are still unresolved - its WINDOW: G = 1: the previous batch;
// G = 2: two or three batches - and everything produced per batch lives in slot b mod 2G.
struct PipeBuf {
    double *prop;    // [T][N][2D + 2] : x1[D], u, (unused), x0[D] (snapshot)
    double *full0;   // [2G][T][PP_B][parts][2] : (sum of linear terms, ratio of products)
    double *Hd;      // [2G][T][PP_B][PP_B] : Hd[m][k], k > m, both in the batch: the FACTOR
                     //                      exp(H[k][m]) of node m's acceptance
    double *Hx;      // [2G][T][xr][PP_B] : Hx[m][k], m the m-th node of the window's earlier batches
    int32_t *acc;    // [T][2G][PP_ACC] : count, accepted nodes of the batch in that slot, their masks
    double *consts;  // [2] : E = exp(sum of intercepts), flush interval
    const int32_t *nctrl;   // case-control: valid controls per (t, i, direction)
    int parts, per, nbat;
    int G, xr;       // batches per launch; rows of an Hx block = (2G - 1) PP_B
    // (ProposeBuf's flag words of the persistent form removed in round 5: always null / 0 here)
    int32_t *sync; int nsync, queue0;
    LsmDeviceState *lsm_draw;   // not NULL: the proposal pass also draws the intercept proposal
    PipeLds lds;     // kernels_pipe_lds.hpp: the evaluators' arguments (filled by the host per launch)
    // the cross products by the evaluators (kernels_pipe_lds.hpp, pipe_xserve): slot [t][k] holds PP_XP_EMPTY
    // until the wavefront that serves row k of slice t has stored prod_{m accepted} Hx[k][m]; the row's owner in
    // the resolver workgroup polls it past its L1 and empties it again
    double *xprod; int32_t *err; int xserve, budget;
    int lds_eval;    // undirected model: the evaluators of kernels_pipe_lds.hpp (rows staged in LDS, interleaved
                     // parts, H factors inside the trips); 0: pipe_eval_item
};
constexpr unsigned long long PP_XP_EMPTY = 0x7FF8C0DE5EED0002ull;     // a NaN payload no arithmetic produces
constexpr int PP_ERR_XSERVE = 1 << 29;          // sticky error word: a resolver ran out of its poll budget
// first batch of the window of batch b
__host__ __device__ __forceinline__ int pipe_window_start(int b, int G) {
    const int ws = G * (b / G - 1);
    return ws > 0 ? ws : 0;
}

// The proposal pass of a sweep, as pieces: it runs as its own launch (k_pipe_propose) or, inside
// the device-resident loops, as extra workgroups of the previous iteration's last launch
// (kernels_tail_propose.hpp) - everything it reads is final by then.
__device__ __forceinline__ void pipe_propose_consts(const ChainView &c, double *consts,
                                                    const double *intercept) {
    const double E = c.model == DLSM_UNDIRECTED ? exp(intercept[0]) : exp(intercept[0] + intercept[1]);
    consts[0] = E;
    consts[1] = (double)flush_interval(E);
}

// The undirected loop's intercept proposal and the log-uniform of its accept test
// (sample_coefficients.py:76-86; Philox stream INTERCEPT, counters (0, 0 | 1, iter)) - what the
// centring pass draws (k_post_apply), from the same counters and the same intercept and step size,
// which are settled when the previous iteration ends: drawn with the sweep's proposals, the
// likelihood pass can run before the centring pass (kernels_tail_propose.hpp).
__device__ __forceinline__ void pipe_propose_intercept(const ChainView &c, LsmDeviceState *lsm,
                                                       const double *intercept, uint32_t iter) {
    double u0, u1, z0, z1;
    philox_uniform2(c.seed, 0, 0, iter, stream_word(c.chain, STREAM_INTERCEPT), u0, u1);
    box_muller(u0, u1, z0, z1);
    const double b0 = intercept[0];
    lsm->cand[0] = b0;
    lsm->cand[1] = b0 + lsm->i_step[0] * z0;
    philox_uniform2(c.seed, 0, 1, iter, stream_word(c.chain, STREAM_INTERCEPT), u0, u1);
    lsm->logu = log(u0);
}

// the proposal of node (t, j) from its position x0
template <int D>
__device__ __forceinline__ void pipe_propose_row_from(const ChainView &c, const ProposeBuf &pb, uint32_t iter,
                                                      int t, int j, const double (&x0)[D]) {
    const int N = c.N;
    double x1[D], logu;
    make_proposal<D>(c, iter, t, j, x0, c.step[(size_t)t * N + j], x1, logu);
    double *pr = pb.prop + ((size_t)t * N + j) * (2 * D + 2);
#pragma unroll
    for (int d = 0; d < D; ++d) { pr[d] = x1[d]; pr[D + 2 + d] = x0[d]; }
    {   // the uniform itself (same draw as make_proposal's log u)
        double u0, u1;
        philox_uniform2(c.seed, (uint32_t)j, (uint32_t)t, iter,
                        stream_word(c.chain, STREAM_SWEEP_UNIFORM), u0, u1);
        pr[D] = u0;
    }
    pr[D + 1] = 0.0;
}

// workgroup `fb` of ceil(N / 256) T, 256 threads
template <int D>
__device__ __forceinline__ void pipe_propose_rows(const ChainView &c, const ProposeBuf &pb, uint32_t iter,
                                                  int fb, int tid) {
    const int N = c.N, nbx = (N + 255) / 256;
    const int t = fb / nbx;
    const int j = (fb - t * nbx) * 256 + tid;
    if (pb.sync) {
        const int flat = fb * 256 + tid;
        if (flat < pb.nsync) pb.sync[(size_t)flat * 16] = flat == 0 ? pb.queue0 : 0;
    }
    if (j >= N) return;
    // valid for the whole sweep: X[t, j] and its step size change only at step (t, j)
    double x0[D], x1[D], logu;
#pragma unroll
    for (int d = 0; d < D; ++d) x0[d] = c.X[((size_t)t * N + j) * D + d];
    make_proposal<D>(c, iter, t, j, x0, c.step[(size_t)t * N + j], x1, logu);
    double *pr = pb.prop + ((size_t)t * N + j) * (2 * D + 2);
#pragma unroll
    for (int d = 0; d < D; ++d) { pr[d] = x1[d]; pr[D + 2 + d] = x0[d]; }
    {   // the uniform itself (same draw as make_proposal's log u)
        double u0, u1;
        philox_uniform2(c.seed, (uint32_t)j, (uint32_t)t, iter,
                        stream_word(c.chain, STREAM_SWEEP_UNIFORM), u0, u1);
        pr[D] = u0;
    }
    pr[D + 1] = 0.0;
}

template <int D>
__global__ __launch_bounds__(256) void k_pipe_propose(ChainView c, PipeBuf pb, IterRef ir) {
    const int fb = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
    if (fb == 0 && threadIdx.x == 0) {
        pipe_propose_consts(c, pb.consts, c.intercept);
        if (pb.lsm_draw) pipe_propose_intercept(c, pb.lsm_draw, c.intercept, ir.get());
    }
    const ProposeBuf nb{pb.prop, pb.consts, pb.sync, pb.nsync, pb.queue0, pb.lsm_draw};
    pipe_propose_rows<D>(c, nb, ir.get(), fb, (int)threadIdx.x);
}

#ifdef DLSM_PIPE_TIMING
// phase stamps (100 MHz constant clock) of every evaluator wavefront and every resolver
// workgroup of the last sweep: profiles/pipe_timing.py reads them
__device__ unsigned long long g_pipe_item_t[24][4096][6];
__device__ unsigned long long g_pipe_res_t[24][32][5];
__device__ __forceinline__ unsigned long long pipe_clock(double dep) {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep));
    return t;
}
#define DLSM_STAMP(I_, DEP_) ts[I_] = pipe_clock(DEP_);
#else
#define DLSM_STAMP(I_, DEP_)
#endif

// ---- the item's wavefront reductions -------------------------------------------------------
// An evaluator launch is bound by float64 VALU issue (profiles/pipe_timing.py: the four
// wavefronts of a SIMD run one after the other, the SIMD is busy from the first operand's
// arrival to the launch's end), so what is not a neighbour term is overhead to be counted in
// instructions.  Only lane 0 stores the record: the sum and the two products are reduced
// towards it, the two products share one tree from the second step on (even lanes carry P0, odd lanes
// P1: quad_perm xor 2 and the row rotations keep the parity), and the rows are combined
// through the LDS crossbar (ds_bpermute: no VALU cycles) instead of v_readlane.
__device__ __forceinline__ double lane_get(double v, int byte_addr) {      // 4 * source lane
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v)),
                            __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v)));
}
// lane 0: sum = sum over the wavefront of s, ratio = prod a / prod b
__device__ __forceinline__ void pipe_reduce(double s, double a, double b, int lane, double &sum,
                                            double &ratio) {
    const int x16 = (lane ^ 16) << 2, x32 = (lane ^ 32) << 2;
    s += dpp_move<0xB1>(s);         // quad_perm [1, 0, 3, 2]
    a *= dpp_move<0xB1>(a);
    b *= dpp_move<0xB1>(b);
    double q = (lane & 1) ? b : a;
    s += dpp_move<0x4E>(s);         // quad_perm [2, 3, 0, 1]
    q *= dpp_move<0x4E>(q);
    s += dpp_move<0x124>(s);        // row_ror:4
    q *= dpp_move<0x124>(q);
    s += dpp_move<0x128>(s);        // row_ror:8
    q *= dpp_move<0x128>(q);
    s += lane_get(s, x16);
    q *= lane_get(q, x16);
    s += lane_get(s, x32);
    q *= lane_get(q, x32);
    sum = s;
    // lane 0: prod a / prod b (products of factors >= 1 that the caller keeps inside the double
    // range: the reciprocal's Newton form, within 2 ulp of the division at a fifth of it)
    ratio = q * fast_rcp(dpp_move<0xB1>(q));
}

#include "kernels_pipe_item.hpp"
#include "kernels_pipe_ccdense.hpp"

// ---- the resolver for H blocks stored BY ROWS (every launch of the exact-likelihood models) ----------------------
// The evaluators file the factors of later node kk as ONE row [window (PP_B) | own batch (PP_B)]: a wavefront's
// 64 entries - consecutive entries of one row in the flat list of pipe_h_decode - are 512 contiguous bytes
// instead of 64 stores a kilobyte apart (the [m][k] blocks: 2.15 MB written and 7.9 MB fetched per launch, each
// 128-byte line assembled from 16 wavefronts on eight XCDs).
constexpr int PR_LD = PP_B + 1;          // LDS row stride of the diagonal block (doubles)

// the product (or, for a node resolved in the log domain, the sum) over the 8 adjacent lanes
// that share a node: quad_perm xor 1, xor 2, then the half-row mirror
__device__ __forceinline__ double group8_prod(double v) {
    v *= dpp_move<0xB1>(v);
    v *= dpp_move<0x4E>(v);
    v *= dpp_move<0x141>(v);
    return v;
}
__device__ __forceinline__ double group8_sum(double v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    return v;
}

template <int D>
__device__ __forceinline__ void row_resolve(const ChainView &c, const PipeBuf &pb, int b, int t,
                                                double *sD, double *sPart,
                                                unsigned long long (*sMask)[2],
                                                double *sCross,
                                                unsigned long long *sSatMask, double *sTab, bool served
#ifdef DLSM_PIPE_TIMING
                                                , int tl
#endif
                                                ) {
    constexpr int PW = 2 * D + 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef DLSM_PIPE_TIMING
    unsigned long long ts[5] = {0, 0, 0, 0, 0};
#endif
    DLSM_STAMP(0, (double)tid)
    const int N = c.N;
    const int j0 = b * PP_B;
    const int nb = min(PP_B, N - j0);
    const int bb = b & 1;
    const int half = wave & 1, part = wave >> 1;
    const int k = 64 * half + lane;
    const bool owner = wave < 2;
    const bool valid = k < nb;
    const double *H = pb.Hd + ((size_t)bb * c.T + t) * PP_B * (2 * PP_B);
    // ---- every load of the batch, at once ------------------------------------------------------
    // cross factors: node kx = tid / 8; trip u: window nodes 16 u + 2 jx, + 1 - the 8 lanes of a
    // node read one 128-byte line per trip
    const int kx = tid >> 3, jx = tid & 7;
    double cr[16];
    const uint32_t cr_off = (uint32_t)((kx * (2 * PP_B) + 2 * jx) * sizeof(double));
    // (served: the launch's evaluators multiply the accepted factors of every row and hand over one number per
    // node - the 128 KB of the cross block were 1.8 us of this workgroup's 4.1 us until its blocks were in LDS)
    if (b > 0 && !served) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#if defined(DLSM_X_RES) && (DLSM_X_RES & 1)      // measurement only: no cross block
            const double2 v = make_double2(1.0, 1.0);
#else
            const double2 v = coh_load2<false>(H, cr_off, (uint32_t)(16 * u * sizeof(double)));
#endif
            cr[2 * u] = v.x; cr[2 * u + 1] = v.y;
        }
    }
    // diagonal block: 16 bytes per thread and trip, a wavefront per row (rows >= nb and columns
    // >= the row's node were never written: never read either)
    // (trip u: row 16 u + wave, columns 2 lane, 2 lane + 1)
    // (d = 3, 4: the block in two halves - the owners' positions, proposals and prior operands are 8 d
    // registers more, and all of it at once spilled 3 - 35 registers: the second half is requested once
    // the first has left for LDS, one round trip later, on instantiations no benchmark configuration runs)
    constexpr int BLK_H = D <= 2 ? 8 : 4;
    double2 blk[BLK_H];
    const uint32_t blk_off = (uint32_t)((wave * (2 * PP_B) + PP_B + 2 * lane) * sizeof(double));
#pragma unroll
    for (int u = 0; u < BLK_H; ++u)
#if defined(DLSM_X_RES) && (DLSM_X_RES & 2)      // measurement only: no diagonal block
        blk[u] = make_double2(1.0, 1.0);
#else
        blk[u] = coh_load2<false>(H, blk_off, (uint32_t)(u * PP_WAVES * (2 * PP_B) * sizeof(double)));
#endif
    // the owners' inputs (requested now, used once the blocks above have left their registers)
    double2 tv[4];
    double x1[D], x0[D], uk = 1.0, st = 0.0;
    int32_t na = 0, ns = 0, un = 0;
    NodePriorPre<D> npre;
    const int kc = min(k, nb - 1);
    const int p1 = pb.parts;
    const double2 *frec = (const double2 *)pb.full0 + ((size_t)bb * c.T + t) * PP_B * pb.parts;
#pragma unroll
    for (int d = 0; d < D; ++d) { x1[d] = 0.0; x0[d] = 0.0; }
#pragma unroll
    for (int u = 0; u < 4; ++u) tv[u] = make_double2(0.0, 1.0);
    auto owner_loads = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            tv[u] = coh_load2<false>(frec, (uint32_t)((kc * p1 + min(u, p1 - 1)) * sizeof(double2)));
        const double *pr = pb.prop + ((size_t)t * N + j0 + kc) * PW;
#pragma unroll
        for (int d = 0; d < D; ++d) { x1[d] = pr[d]; x0[d] = pr[D + 2 + d]; }
        uk = pr[D];
        const size_t tjc = (size_t)t * N + j0 + kc;
        st = c.step[tjc]; na = c.nacc[tjc]; ns = c.nsteps[tjc]; un = c.until[tjc];
        npre.request1(c, t, j0 + kc);        // the prior terms' operands, with everything else
    };
    if (owner) owner_loads();
    // the owners' exponential table (tab_exp), requested with everything else: it is read behind the barrier
    // below (round 6: its own fill + barrier in front of this function kept every request of the batch
    // a memory round trip late - the resolver is the launch's critical path once the evaluators are shorter)
    double tabv = 0.0;
    if (tid < EXPTAB_N) tabv = c_exp2_tab[tid];
#pragma unroll
    for (int u = 0; u < BLK_H; ++u) {
        double *dst = sD + (u * PP_WAVES + wave) * PR_LD + 2 * lane;
        dst[0] = blk[u].x; dst[1] = blk[u].y;
    }
    if (BLK_H < 8) {
        __builtin_amdgcn_sched_barrier(0);              // (not hoisted among the loads above)
#pragma unroll
        for (int u = 0; u < 8 - BLK_H; ++u)
            blk[u] = coh_load2<false>(H, blk_off, (uint32_t)((BLK_H + u) * PP_WAVES * (2 * PP_B) * sizeof(double)));
#pragma unroll
        for (int u = 0; u < 8 - BLK_H; ++u) {
            double *dst = sD + ((BLK_H + u) * PP_WAVES + wave) * PR_LD + 2 * lane;
            dst[0] = blk[u].x; dst[1] = blk[u].y;
        }
    }
    if (owner) npre.request2(c);               // (the labels are in: the components' means and variances)
    // the window's accepted nodes among this thread's 16 factors: cr[2 u + i] belongs to node
    // 16 u + 2 jx + i -> bit 2 u + i of mp.  Multiplicative domain; a node that turns out to be
    // resolved in the log domain (below) redoes its part from memory.
    unsigned int mp = 0u;
    if (b > 0) {
        // the previous batch's acceptances (the launch before left them)
        const unsigned long long *pmg = (const unsigned long long *)(pb.acc + ((size_t)t * 2 + ((b - 1) & 1)) * PP_ACC +
                                                                     PP_ACC_MASK);
        const unsigned long long pm0 = pmg[0], pm1 = pmg[1];
        const unsigned long long mlo = pm0 >> (2 * jx), mhi = pm1 >> (2 * jx);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            mp |= ((unsigned int)((u < 4 ? mlo : mhi) >> (16 * (u & 3))) & 3u) << (2 * u);
        if (!served) {
            double prod = 1.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) prod *= (mp >> i) & 1u ? cr[i] : 1.0;
            const double v = group8_prod(prod);
            if (jx == 0) sCross[kx] = v;
        }
    }
    // multiplicative domain: r = exp(log-ratio of node k), lu = its uniform draw.  A node whose
    // log-ratio is beyond +-700 (exp would saturate) is resolved in the log domain instead
    // (pipe_resolve)
    double r = 0.0, lu = 0.0, lr = 0.0;
    bool sat = false;
    if (owner) {
        double tot = tv[0].x, pr_ = tv[0].y;
#pragma unroll
        for (int u = 1; u < 4; ++u)
            if (u < p1) { tot += tv[u].x; pr_ *= tv[u].y; }
        for (int u = 4; u < p1; ++u) {                 // more than four parts (short slices, many CUs)
            const double2 w = coh_load2<false>(frec, (uint32_t)((kc * p1 + u) * sizeof(double2)));
            tot += w.x; pr_ *= w.y;
        }
        // prior terms of the step's logp closure, the neighbouring slices as they are now (the
        // odd slices wait for the even ones: header)
        const double prior = npre.value(x1) - npre.value(x0);
        const double ek = tot + prior;
        sat = !(fabs(ek) <= 700.0);
        // (r = exp(ek) pr_ behind the barrier: the table is in LDS by then)
        r = sat ? 1.0 : pr_;
        lr = ek;
        lu = uk;
        if (sat) { lr = ek + log(pr_); lu = log(lu); }
        const unsigned long long sm = __ballot(sat);
        if (lane == 0) sSatMask[half] = sm;
    }
    if (tid < EXPTAB_N) sTab[tid] = tabv;
    __syncthreads();                                   // sD, sCross, sSatMask, sTab visible
    // (the table exponential: the compiler's exp() keeps a dozen float64 constants alive through
    // the whole batch loop, in registers the blocks above need - it spilled them)
    if (owner && !sat) r *= tab_exp(lr, sTab);
    DLSM_STAMP(1, (double)tid)
    const unsigned long long sat0 = sSatMask[0], sat1 = sSatMask[1];
    const bool anysat = (sat0 | sat1) != 0ull;         // workgroup-uniform; practically never
    const bool satk = (((half ? sat1 : sat0) >> lane) & 1ull) != 0ull;
    if (anysat && b > 0) {
        // slow path: the log-domain nodes' cross terms as sums of logs
        const bool satx = ((((kx >> 6) ? sat1 : sat0) >> (kx & 63)) & 1ull) != 0ull;
        if (__ballot(satx) != 0ull) {
            double lsum = 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double2 v = coh_load2<false>(H, cr_off, (uint32_t)(16 * u * sizeof(double)));
                if (satx && ((mp >> (2 * u)) & 1u)) lsum += log(v.x);
                if (satx && ((mp >> (2 * u + 1)) & 1u)) lsum += log(v.y);
            }
            const double vs = group8_sum(lsum);
            if (satx && jx == 0) sCross[kx] = vs;
        }
        __syncthreads();
    }
    if (owner) {
        if (b > 0) {
            double vk;
            if (served && !sat) {
                // the product announces itself: the slot is PP_XP_EMPTY until the serving wavefront has stored it
                vk = 1.0;
                if (valid) {
                    const unsigned long long *slotp = (const unsigned long long *)&pb.xprod[(size_t)t * PP_B + k];
                    unsigned long long got = PP_XP_EMPTY;
                    for (int n = 0; n < pb.budget; ++n) {
                        got = __hip_atomic_load(slotp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (__ballot(got == PP_XP_EMPTY) == 0ull) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (got == PP_XP_EMPTY)
                        __hip_atomic_fetch_or(pb.err, PP_ERR_XSERVE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    vk = __longlong_as_double((long long)got);
                    // taken: empty again for the next launch (the kernel boundary publishes the plain store)
                    pb.xprod[(size_t)t * PP_B + k] = __longlong_as_double((long long)PP_XP_EMPTY);
                }
            } else {
                if (served && valid)    // (a node resolved in the log domain: its slot is emptied, its sum is sCross's)
                    pb.xprod[(size_t)t * PP_B + k] = __longlong_as_double((long long)PP_XP_EMPTY);
                vk = sCross[k];
            }
            if (sat) lr += vk; else r *= vk;
        }
        const unsigned long long g = __ballot(valid && !(sat ? lu >= lr : lu >= r));
        if (lane == 0) sMask[0][half] = g;
    }
    __syncthreads();
    DLSM_STAMP(2, (double)tid)
    int cur = 0;
    for (int pass = 0; pass < 2 * PP_B + 2; ++pass) {
        const unsigned long long gm = sMask[cur][part >> 2];
        unsigned int bits = (unsigned int)(gm >> (16 * (part & 3))) & 0xFFFFu;
        const int mbase = 16 * part;
        double sum = 1.0, lsum = 0.0;
        if (half == 1 || part < 4) {                   // rows >= 64 never touch half 0
            const double *rowk = sD + k * PR_LD;
            while (bits) {
                int f[4];
                double h[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    f[u] = bits ? mbase + __builtin_ctz(bits) : 1 << 20;
                    bits &= bits - 1u;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) h[u] = rowk[f[u] < (1 << 20) ? f[u] : 0];
                if (!anysat) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) sum *= k > f[u] ? h[u] : 1.0;
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (k > f[u]) { if (satk) lsum += log(h[u]); else sum *= h[u]; }
                }
            }
        }
        sPart[wave * 64 + lane] = satk ? lsum : sum;
        __syncthreads();
        if (owner) {
            double q = sat ? lr : r;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const double v = sPart[(2 * p + half) * 64 + lane];
                if (sat) q += v; else q *= v;
            }
            const unsigned long long g = __ballot(valid && !(lu >= q));
            if (lane == 0) sMask[cur ^ 1][half] = g;
        }
        __syncthreads();
        const bool same = sMask[cur ^ 1][0] == sMask[cur][0] &&
                          sMask[cur ^ 1][1] == sMask[cur][1];
        cur ^= 1;
        if (same) break;
    }
    DLSM_STAMP(3, (double)cur)
    if (owner) {
        const unsigned long long m0 = sMask[cur][0], m1 = sMask[cur][1];
        const unsigned long long mine = half == 0 ? m0 : m1;
        const int accepted = (int)((mine >> lane) & 1ull);
        if (valid) {
            const size_t tj = (size_t)t * N + j0 + k;
            if (accepted) {
#pragma unroll
                for (int d = 0; d < D; ++d) coh_store<false>(&c.X[tj * D + d], x1[d]);
            }
            metropolis_bookkeeping(st, na, ns, un, c.tune, c.tune_interval, accepted);
            c.step[tj] = st; c.nacc[tj] = na; c.nsteps[tj] = ns; c.until[tj] = un;
        }
        // the next batch's window: this batch's acceptances, as a mask
        if (tid == 0) {
            unsigned long long *pmw = (unsigned long long *)(pb.acc + ((size_t)t * 2 + (b & 1)) * PP_ACC + PP_ACC_MASK);
            pmw[0] = m0; pmw[1] = m1;
        }
    }
#ifdef DLSM_PIPE_TIMING
    DLSM_STAMP(4, (double)cur)
    if (tid == 0 && tl >= 0 && tl < 24 && t < 32)
        for (int i = 0; i < 5; ++i) g_pipe_res_t[tl][t][i] = ts[i];
#endif
}

}  // namespace dlsm
#include "kernels_pipe_lds.hpp"
namespace dlsm {

// Launch l: even slices resolve batches G l .. G l + G - 1 and evaluate batches G (l + 1) ..;
// odd slices run one launch behind (batches outside [0, nbat) do nothing).
// Workgroups [0, T) are the resolvers, the rest evaluate one item per wavefront and round.
// MODEL = PIPE_UNDIRECTED_LONG: the undirected model with parts longer than the prefetch
constexpr int PIPE_UNDIRECTED_LONG = 3;

template <int D, int MODEL_, int G = 1>
__global__ __launch_bounds__(PP_THREADS) void k_pipe_step(ChainView c, PipeBuf pb, int l) {
    constexpr int MODEL = MODEL_ == PIPE_UNDIRECTED_LONG ? DLSM_UNDIRECTED : MODEL_;
    constexpr bool TP = MODEL_ != DLSM_UNDIRECTED;
    extern __shared__ __attribute__((aligned(16))) double pp_sH[];      // 128 x PR_LD
    __shared__ double sPart[PP_WAVES * 64];
    __shared__ unsigned long long sMask[2][2];
    __shared__ int sPrev[3 * PP_B];
    __shared__ int sOwn[PP_B + 1];
    __shared__ unsigned char sSat[PP_B];
    const int T = c.T;
    // one batch per launch, exact likelihoods: the H blocks by rows and their resolver (row_resolve)
    constexpr bool ROWS = G == 1 && MODEL != DLSM_DIRECTED_CASE_CONTROL;
    const int bx = (int)blockIdx.x;
    const bool grid3 = MODEL_ == DLSM_UNDIRECTED && G == 1 && pb.lds_eval && (int)gridDim.x > T;    // kernels_pipe_lds.hpp's evaluators
    if (grid3) {
        if (bx >= T) {
#ifdef DLSM_PIPE_TIMING
            const unsigned long long t_kernel = pipe_clock((double)threadIdx.x);    // the wavefront's first instruction
#endif
            pipe_eval_lds<D>(c, pb, l, pp_sH
#ifdef DLSM_PIPE_TIMING
                             , t_kernel
#endif
                             );
            return;
        }
    }
    if (bx < T) {
        const int t = bx;
        if (ROWS) {
            __shared__ double sCross[PP_B];
            __shared__ unsigned long long sSatMask[2];
            __shared__ double sTab[EXPTAB_N];
            const int b = l - (t & 1);
            const bool served = pb.xserve != 0 && grid3;                  // (both sides decide by the same rule)
            if (l == -1 && threadIdx.x < PP_B)                            // the sweep's first launch: every slot empty
                pb.xprod[(size_t)t * PP_B + threadIdx.x] = __longlong_as_double((long long)PP_XP_EMPTY);
            if (b < 0 || b >= pb.nbat) return;
            row_resolve<D>(c, pb, b, t, pp_sH, sPart, sMask, sCross, sSatMask, sTab, served
#ifdef DLSM_PIPE_TIMING
                                  , l + 1
#endif
                                  );
            return;
        }
        bool own_prev = false;
        for (int g = 0; g < G; ++g) {
            const int b = G * (l - (t & 1)) + g;
            if (b < 0 || b >= pb.nbat) continue;
            if (own_prev) __syncthreads();          // the LDS of the batch before is free again
            pipe_resolve<D, G>(c, pb, b, t, pp_sH, sPart, sMask, sPrev, sSat, sOwn, own_prev
#ifdef DLSM_PIPE_TIMING
                            , l + 1
#endif
                            );
            own_prev = true;
        }
        return;
    }
    // (the undirected model's evaluators are kernels_pipe_lds.hpp's above; when their rows do not fit the LDS or
    // the items need a second round the host launches the PIPE_UNDIRECTED_LONG instantiation, whose evaluators are
    // pipe_eval_item's software-pipelined trips below)
    if (MODEL_ == DLSM_UNDIRECTED) return;
#ifdef DLSM_PIPE_TIMING
    const unsigned long long t_kernel = pipe_clock((double)threadIdx.x);    // the wavefront's first instruction
#endif
    const int lane = threadIdx.x & 63;
    const int nE = (T + 1) / 2, nO = T / 2;
    const int beE = G * (l + 1), beO = G * l;        // first batch evaluated (even / odd slices)
    const int nbE = (beE >= 0 && beE < pb.nbat) ? min(PP_B, c.N - beE * PP_B) : 0;
    const int nbO = (beO >= 0 && beO < pb.nbat) ? min(PP_B, c.N - beO * PP_B) : 0;
    if (MODEL == DLSM_DIRECTED_CASE_CONTROL) {
        // four nodes per workgroup round, CC_PARTS wavefronts each (pb.parts == CC_PARTS)
        static_assert(PP_B == 128 && CC_PARTS * 64 == 2 * PP_B, "a group's wavefronts cover its LDS column");
        const int wave = threadIdx.x >> 6, sub = wave & (CC_PARTS - 1), grp = wave / CC_PARTS;
        constexpr int GROUPS = PP_WAVES / CC_PARTS;
        double *col = pp_sH + grp * (2 * PP_B);
        const int nodesE = nE * nbE, nodes = nodesE + nO * nbO;
        for (int base = ((int)blockIdx.x - T) * GROUPS; base < nodes;
             base += ((int)gridDim.x - T) * GROUPS) {
            const int q = base + grp;
            const bool valid = q < nodes;
            const bool odd = q >= nodesE;
            const int qq = odd ? q - nodesE : q;
            const int nb = max(odd ? nbO : nbE, 1);
            const int k = qq % nb;
            const int t = 2 * (qq / nb) + (odd ? 1 : 0);
            const int be = odd ? beO : beE;
            col[64 * sub + lane] = 0.0;
            __syncthreads();
            if (valid) pipe_cc_accumulate<D>(c, pb, be, t, k, sub, lane, col);
            __syncthreads();
            if (valid) pipe_cc_writeout<D>(c, pb, be, t, k, sub, lane, col);
            __syncthreads();
        }
        return;
    }
    // items ordered (part, active slice, k) with 128 k-slots per slice, so that an item id
    // decodes with a shift, a mask and one small quotient (no integer divisions)
    // (a launch's first batch exists whenever any of its G batches does)
    constexpr int gsh = G == 2 ? 8 : 7;                // k-slots per slice: G * 128
    const int nslE = nbE > 0 ? nE : 0, nslO = nbO > 0 ? nO : 0, nsl = nslE + nslO;
    const int nitems = (pb.parts * nsl) << gsh;
    const float inv_nsl = 1.0f / (float)max(nsl, 1);
    // the evaluators' table of 2^(j / 2048) (tab_exp11) in the dynamic LDS the resolvers use for H
    // (a barrier-free fill - every wavefront copying the table itself with global_load_lds_dwordx4 -
    // measured no better: 3614 against 3638 it/s; nor did the barrier behind the item's loads: 3615)
    if (MODEL == DLSM_UNDIRECTED) {
        exp_table11_fill<PP_THREADS>(pp_sH, threadIdx.x);
        __syncthreads();
    }
    const int nwaves = ((int)gridDim.x - T) * PP_WAVES;
    const int gw = __builtin_amdgcn_readfirstlane(
        ((int)blockIdx.x - T) * PP_WAVES + (int)(threadIdx.x >> 6));
    for (int q = gw; q < nitems; q += nwaves) {
        const int k = q & (PP_B - 1);
        const int g = (q >> 7) & (G - 1);
        const int r = q >> gsh;
        const int p = (int)(((float)r + 0.5f) * inv_nsl);        // r / nsl (r < 2^20)
        const int si = r - p * nsl;
        const bool odd = si >= nslE;
        const int be = (odd ? beO : beE) + g;
        const int nb = be < pb.nbat ? min(PP_B, c.N - be * PP_B) : 0;
        if (k >= nb) continue;
        const int t = odd ? 2 * (si - nslE) + 1 : 2 * si;
        constexpr int IM = MODEL == DLSM_DIRECTED_CASE_CONTROL ? DLSM_DIRECTED : MODEL;
        PipeItemPre<D> pre;
        pipe_item_prologue<D, IM>(c, pb, be, t, k, p, lane, pre);
        pipe_eval_item<D, IM, TP, G>(c, pb, be, nb, t, k, p, lane, pp_sH, pre
#ifdef DLSM_PIPE_TIMING
            , l + 1, gw
#endif
            );
#ifdef DLSM_PIPE_TIMING
        // (slot 4, "H operands here", gives way to the wavefront's first stamp in the kernel: both evaluators
        // on one time axis)
        if (lane == 0 && l + 1 >= 0 && l + 1 < 24 && q < 4096) g_pipe_item_t[l + 1][q][4] = t_kernel;
#endif
    }
}

// The sweep's LAST launch only resolves (the last batch of the odd slices; of the even ones when
// T = 1): a handful of workgroups busy for 9 us and the chip idle - while the centring sums, the next
// launch, need nothing but the positions that are final already.  Here they ride in that launch as
// workgroups T ..: every row except the nodes i >= jl of the slices that are still being resolved
// (and the difference terms that touch them: post_row_own_left / post_row_diff_left), which the
// resolver workgroups sum themselves once they have settled them (records nwg .. nwg + T - 1).
// The resolver workgroups run exactly what k_pipe_step runs for them before that.
struct PipePostRide { const double *xref; IterRef ir; double *rec; int nwg, jl, par; };
// workgroup `bx` of T + nwg (the LDS of k_pipe_step's resolvers is the caller's)
template <int D>
__device__ __forceinline__ void pipe_last_ride_wg(const ChainView &c, const PipeBuf &pb, int l,
                                                  const PipePostRide &pr, int bx, double *pp_sH,
                                                  double *sPart, unsigned long long (*sMask)[2],
                                                  int *sPrev, int *sOwn, unsigned char *sSat) {
    const int T = c.T;
    if (bx < T) {
        // The rows this launch is still moving (nodes i >= jl of this slice) and the difference terms
        // that touch them are summed HERE, by the workgroup that has just settled them, into record
        // nwg + t: nobody else may read those rows while the launch runs, and the centring pass may
        // not read any position at all once its workgroups have begun to rewrite them in place.
        constexpr int W = PostRec<D>::W;
        __shared__ double sRedL[W][PP_WAVES];
        const int t = bx, tid = threadIdx.x;
        const int b = l - (t & 1);
        const bool mine = b >= 0 && b < pb.nbat;
        if (mine) {
            __shared__ double sCross[PP_B];
            __shared__ unsigned long long sSatMask[2];
            __shared__ double sTab[EXPTAB_N];
            row_resolve<D>(c, pb, b, t, pp_sH, sPart, sMask, sCross, sSatMask, sTab, false
#ifdef DLSM_PIPE_TIMING
                                  , l + 1
#endif
                                  );
        }
        __syncthreads();                    // the accepted positions of this workgroup are in memory
        double acc[W];
#pragma unroll
        for (int q = 0; q < W; ++q) acc[q] = 0.0;
        if (mine && (t & 1) == pr.par) {
            const int nl = c.N - pr.jl;
            for (int q = tid; q < 2 * nl; q += PP_THREADS) {
                const bool nxt = q >= nl;                   // the difference term of the slice behind
                const int i = pr.jl + (nxt ? q - nl : q), tt = nxt ? t + 1 : t;
                if (tt >= T) continue;
                post_row_terms<D>(c, pr.xref, (long)tt * c.N + i, !nxt, tt >= 1, acc);
            }
        }
#pragma unroll
        for (int q = 0; q < W; ++q) {
            const double v = wave_sum_all(acc[q]);
            if ((tid & 63) == 0) sRedL[q][tid >> 6] = v;
        }
        __syncthreads();
        if (tid < W) {
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < PP_WAVES; ++w) v += sRedL[tid][w];
            pr.rec[(size_t)(pr.nwg + t) * W + tid] = v;
        }
        return;
    }
    post_reduce_wg<D, PP_THREADS>(c, pr.xref, -1, pr.ir, pr.rec, bx - T, pr.nwg, pr.jl, pr.par);
}

template <int D>
__global__ __launch_bounds__(PP_THREADS) void k_pipe_last_ride(ChainView c, PipeBuf pb, int l, PipePostRide pr) {
    extern __shared__ __attribute__((aligned(16))) double pp_sH[];      // 128 x 128
    __shared__ double sPart[PP_WAVES * 64];
    __shared__ unsigned long long sMask[2][2];
    __shared__ int sPrev[3 * PP_B];
    __shared__ int sOwn[PP_B + 1];
    __shared__ unsigned char sSat[PP_B];
    pipe_last_ride_wg<D>(c, pb, l, pr, (int)blockIdx.x, pp_sH, sPart, sMask, sPrev, sOwn, sSat);
}

}  // namespace dlsm
