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

// One neighbour of the directed model (both directions of the pair share the distance):
// delta = log[(1 + E e^{-d0 a})(1 + E e^{-d0 g}) / ((1 + E e^{-d1 a})(1 + E e^{-d1 g}))]
//         + (d0 - d1)(y_ji a + y_ij g),   E = e^{b_in + b_out}
// (delta_directed) as factors of the running products of `ra` - no log, no division and the
// lean exp per neighbour.  An exponent above 40 (negative a at a large distance, the
// degenerate corner of the parameter space) goes through the exact term-by-term form into
// `exact` instead; the products are flushed before they could leave the double range.
__device__ __forceinline__ void pipe_directed_term(RatioAcc &ra, double &exact, double d0, double d1,
                                                   int y_ji, int y_ij, double a, double g, double E,
                                                   double lE) {
    const double x0a = -d0 * a, x0g = -d0 * g, x1a = -d1 * a, x1g = -d1 * g;
    if (fmax(fmax(x0a, x0g), fmax(x1a, x1g)) + lE > 40.0 || !(E < 1e17)) {
        exact += delta_directed(d0, d1, y_ji, y_ij, a, g, E);
        return;
    }
    if (y_ji) ra.lin += (d0 - d1) * a;
    if (y_ij) ra.lin += (d0 - d1) * g;
    if (fmax(ra.P0, ra.P1) > 1e200) ra.flush();
    ra.P0 *= fma(E, fast_exp(x0a), 1.0) * fma(E, fast_exp(x0g), 1.0);
    ra.P1 *= fma(E, fast_exp(x1a), 1.0) * fma(E, fast_exp(x1g), 1.0);
}

// The H entries of a (slice, batch) - node kk has ncross + kk of them: the window's earlier batches
// (cross block), then the earlier nodes of its own batch - as ONE flat list dealt out over the lanes
// working on the (slice, batch).  Rows are listed in PAIRS (r, nb - 1 - r), r < ceil(nb / 2): a pair
// holds L = 2 ncross + nb - 1 entries whatever r, so f -> (pair, offset) is one quotient by a launch
// constant and the pair's two rows are told apart by a compare - 15 vector instructions where the
// triangular prefix of round 3 (prefix(kk) = ncross kk + kk (kk - 1) / 2, inverted by a float root
// and a two-branch fix-up) took 28 and was evaluated twice per entry.  Which lane computes an entry
// does not change its value: the blocks are bit for bit the same.  (nb odd: the middle row is a
// pair of its own whose second member is empty - its slots beyond ncross + (nb - 1) / 2 are idle.)
struct PipeHList { int L, npairs, nslots; float invL; };
__device__ __forceinline__ PipeHList pipe_h_list(int ncross, int nb) {
    PipeHList h;
    h.L = max(2 * ncross + nb - 1, 1);
    h.npairs = (nb + 1) >> 1;
    h.nslots = h.npairs * h.L;
    h.invL = 1.0f / (float)h.L;
    return h;
}
// slot f < nslots -> (kk, e); false: the slot is idle (second half of an odd batch's middle pair)
__device__ __forceinline__ bool pipe_h_decode(int f, const PipeHList &hl, int ncross, int nb, int &kk, int &e) {
    int r = (int)(((float)f + 0.5f) * hl.invL);            // f / L (f < 2^17: off by one at most)
    int rem = f - r * hl.L;
    if (rem < 0) { --r; rem += hl.L; } else if (rem >= hl.L) { ++r; rem -= hl.L; }
    r = min(r, hl.npairs - 1);                              // only for a clamped prefetch index
    const int len0 = ncross + r;                            // entries of row r
    const bool second = rem >= len0;
    kk = second ? nb - 1 - r : r;
    e = second ? rem - len0 : rem;
    return !(second && kk == r);                            // the middle row has no partner
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

// Operands of a lane's FIRST H entry (proposal / snapshot rows of the two nodes, the edge's bit,
// the directed model's radii).
template <int D>
struct PipeHPre {
    double xm0[D], xm1[D], xa0[D], xa1[D], rm, rkk;
    uint32_t yw, yw2;
};
template <int D, int MODEL>
__device__ __forceinline__ void pipe_h_operands(const ChainView &c, const double *props,
                                                const char *yrows, const char *ytrows, int jm_,
                                                int jkk, PipeHPre<D> &o) {
    constexpr int PW = 2 * D + 2;
    const double *rowm = (const double *)((const char *)props + __umul24((uint32_t)jm_, (uint32_t)(PW * sizeof(double))));
    const double *rowk = (const double *)((const char *)props + __umul24((uint32_t)jkk, (uint32_t)(PW * sizeof(double))));
#pragma unroll
    for (int d = 0; d < D; ++d) {
        o.xm0[d] = rowm[D + 2 + d];
        o.xm1[d] = rowm[d];
        o.xa0[d] = rowk[D + 2 + d];
        o.xa1[d] = rowk[d];
    }
    const uint32_t woff = ((uint32_t)jkk * (uint32_t)c.W + ((uint32_t)jm_ >> 5)) * 4u;
    o.yw = *(const uint32_t *)(yrows + woff);
    o.yw2 = 0u; o.rm = 1.0; o.rkk = 1.0;
    if (MODEL == DLSM_DIRECTED) {
        o.yw2 = *(const uint32_t *)(ytrows + woff);
        o.rm = c.radii[jm_]; o.rkk = c.radii[jkk];
    }
}
// One H entry: flat index f of (slice t, batch be), operands o.  Rows of the H blocks are addressed as
// 32-bit offsets from a scalar base.
template <int D, int MODEL, int G, bool SQ>
__device__ __forceinline__ void pipe_h_entry(const ChainView &c, const PipeBuf &pb, int be, int nb, int t,
                                             const double *etab, int kk, int e, const PipeHPre<D> &o, bool stamp
#ifdef DLSM_PIPE_TIMING
                                             , unsigned long long *ts
#endif
                                             ) {
    const int j0 = be * PP_B;
    const int jprev = pipe_window_start(be, G) * PP_B;
    const int ncross = j0 - jprev;
    const int bb = be & (2 * G - 1);
    const double E = pb.consts[0];
    // one batch per launch: the blocks by ROWS of the later node (row_resolve); the [m][k] blocks of pipe_resolve
    // are the dense case-control form's
    constexpr bool HROWS = G == 1;
    char *hbase = (char *)(pb.Hd + ((size_t)bb * c.T + t) * PP_B * (HROWS ? 2 * PP_B : PP_B));
    const uint32_t hx_off = (uint32_t)((const char *)(pb.Hx + ((size_t)bb * c.T + t) * ((2 * G - 1) * PP_B) * PP_B) -
                                       (const char *)hbase);                          // one allocation
    const int jm_ = jprev + e;                 // jprev + ncross == j0
    double xm0[D], xm1[D], xa0[D], xa1[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { xm0[d] = o.xm0[d]; xm1[d] = o.xm1[d]; xa0[d] = o.xa0[d]; xa1[d] = o.xa1[d]; }
    const int y1 = (int)((o.yw >> (jm_ & 31)) & 1u);
    const bool cross = e < ncross;
    const int m = cross ? e : e - ncross;
    // (SQ: the squared-distance model, a template flag here as in the trips: as a run-time flag it
    // cost two selects per distance and a clamp per exponential on every entry)
    const double a0 = dist_fast<D>(xm0, xa0, SQ ? 1 : 0);
    const double a1 = dist_fast<D>(xm0, xa1, SQ ? 1 : 0);
    const double b0 = dist_fast<D>(xm1, xa0, SQ ? 1 : 0);
    const double b1 = dist_fast<D>(xm1, xa1, SQ ? 1 : 0);
#ifdef DLSM_PIPE_TIMING
    if (stamp) { DLSM_STAMP(4, a0 + b1) }
#endif
    double h;
    if (MODEL == DLSM_UNDIRECTED) {
        const double eb0 = SQ ? tab_exp11_clamped(-b0, etab) : tab_exp11(-b0, etab);
        const double ea1 = SQ ? tab_exp11_clamped(-a1, etab) : tab_exp11(-a1, etab);
        const double eb1 = SQ ? tab_exp11_clamped(-b1, etab) : tab_exp11(-b1, etab);
        const double ea0 = SQ ? tab_exp11_clamped(-a0, etab) : tab_exp11(-a0, etab);
        double num = fma(E, eb0, 1.0) * fma(E, ea1, 1.0);
        double den = fma(E, eb1, 1.0) * fma(E, ea0, 1.0);
        // the edge's factor e^{(b0 - b1) - (a0 - a1)} from the four exponentials at hand;
        // a fifth one only when their product left the normal range (distances > 300)
        const double fn = eb1 * ea0, fd = eb0 * ea1;
        const bool tiny = y1 && !(fd > 1e-290 && fn > 1e-290);      // both products normal
        if (y1 && !tiny) { num *= fn; den *= fd; }
        // (den is a product of factors >= 1 and, with an edge, of fd > 1e-290: normal, so the
        // reciprocal's Newton form applies - within 2 ulp of the division at a fifth of it)
        h = num * fast_rcp(den);
        if (__builtin_amdgcn_ballot_w64(tiny)) { if (tiny) h *= fast_exp((b0 - b1) - (a0 - a1)); }
    } else {
        double bin = c.intercept[0], bout = c.intercept[1];
        const double lE = bin + bout;
        const int y2 = (int)((o.yw2 >> (jm_ & 31)) & 1u);
        const double irm = 1.0 / o.rm, irkk = 1.0 / o.rkk;
        const double aa = bin * irm + bout * irkk, cc = bin * irkk + bout * irm;
        RatioAcc rb, rq;
        double eb = 0.0, eq = 0.0;
        pipe_directed_term(rb, eb, b0, b1, y1, y2, aa, cc, E, lE);
        pipe_directed_term(rq, eq, a0, a1, y1, y2, aa, cc, E, lE);
        // exp(delta(b) - delta(a)) without the logs: the products divide out
        h = ((rb.P0 * rq.P1) / (rb.P1 * rq.P0)) * exp((rb.lin - rq.lin) + (eb - eq));
    }
    if (HROWS)      // one row of 2 PP_B factors per later node kk - its window's nodes, then its own
                    // batch's - so that a wavefront's entries are contiguous
        coh_store<false>((double *)(hbase + (uint32_t)(kk * (2 * PP_B) + (cross ? e : PP_B + m)) * 8u), h);
    else
        coh_store<false>((double *)(hbase + ((cross ? hx_off : 0u) + (uint32_t)(m * PP_B + kk) * 8u)), h);
}

// The item's tail: wavefront reductions, the (sum, ratio) record, and this lane's share of the
// batch's H entries.
template <int D, int MODEL, int G>
__device__ __forceinline__ void pipe_item_finish(const ChainView &c, const PipeBuf &pb, int be, int nb,
                                                 int t, int k, int p, int lane, const double *etab,
                                                 double acc, RatioAcc &ra, bool noflush
#ifdef DLSM_PIPE_TIMING
                                                 , unsigned long long *ts
#endif
                                                 ) {
    constexpr int PW = 2 * D + 2;
    const int N = c.N, W = c.W;
    const int j0 = be * PP_B;
    const int jprev = pipe_window_start(be, G) * PP_B;
    const int ncross = j0 - jprev;
    const int bb = be & (2 * G - 1);
    const double *props = pb.prop + (size_t)t * N * PW;
    const int hround = nb * pb.parts * 64;
    const PipeHList hl = pipe_h_list(ncross, nb);
    const int hf0 = (k * pb.parts + p) * 64 + lane;
    double tot_l, tot_r;
    if (noflush) {
        // the products of the whole wave stay in range: multiply across lanes
        pipe_reduce(ra.lin + ra.lg, ra.P0, ra.P1, lane, tot_l, tot_r);
    } else {
        acc += ra.value();                       // directed: lin / products and the rare exact terms
        tot_l = wave_sum_all(acc); tot_r = 1.0;
    }
    if (lane == 0) {
        double2 *f = (double2 *)pb.full0 + (((size_t)bb * c.T + t) * PP_B + k) * pb.parts + p;
        coh_store2<false>(f, 0u, make_double2(tot_l, tot_r));
    }
    DLSM_STAMP(3, tot_r)
    // this lane's H entries (pipe_h_decode).  Rows of `props` and the bits are addressed as
    // 32-bit offsets from scalar bases.
    const char *yrows = (const char *)(c.ybits + (size_t)t * N * W);
    const char *ytrows = MODEL == DLSM_DIRECTED ? (const char *)(c.ytbits + (size_t)t * N * W) : nullptr;
    int f = hf0;
#define DLSM_H_CALL(SQ_, KK_, E_, O_, STAMP_)                                                         \
    pipe_h_entry<D, MODEL, G, SQ_>(c, pb, be, nb, t, etab, KK_, E_, O_, STAMP_ DLSM_H_TS)
#ifdef DLSM_PIPE_TIMING
#define DLSM_H_TS , ts
#else
#define DLSM_H_TS
#endif
    // HSHIFT (the launch-per-batch evaluators: a workgroup's 16 wavefronts hold items k0 .. k0 + 15 of one
    // part, wavefronts w, w + 4, w + 8, w + 12 share a SIMD): the LAST wavefront of a SIMD hands its
    // first-round entries to the FIRST one - the same arithmetic on the same SIMD, but no longer the launch's tail
    constexpr bool HSHIFT = DLSM_H_SHIFT != 0;
    const int wig = (int)(threadIdx.x >> 6);
    bool first = true;
    if (HSHIFT && wig >= 12 && f < hround) f += hround;           // handed over (its later rounds stay)
    for (; f < hl.nslots; f += hround) {
        int kk, e;
        if (pipe_h_decode(f, hl, ncross, nb, kk, e)) {
            PipeHPre<D> o;
            pipe_h_operands<D, MODEL>(c, props, yrows, ytrows, jprev + e, j0 + kk, o);
            if (c.squared) DLSM_H_CALL(true, kk, e, o, f == hf0);
            else DLSM_H_CALL(false, kk, e, o, f == hf0);
        }
        if (HSHIFT && first && wig < 4 && k + 12 < nb) {
            // ... and the first-round entries of item k + 12 (same part, same slice: the SIMD's last wavefront)
            const int f2 = hf0 + 12 * pb.parts * 64;
            if (f2 < hl.nslots && pipe_h_decode(f2, hl, ncross, nb, kk, e)) {
                PipeHPre<D> o;
                pipe_h_operands<D, MODEL>(c, props, yrows, ytrows, jprev + e, j0 + kk, o);
                if (c.squared) DLSM_H_CALL(true, kk, e, o, false); else DLSM_H_CALL(false, kk, e, o, false);
            }
        }
        first = false;
    }
#undef DLSM_H_CALL
#undef DLSM_H_TS
    DLSM_STAMP(5, acc)
}

// What an item reads about its OWN node before the first neighbour: static for the whole sweep
// (proposal and snapshot from the propose kernel, the node's row of the network).
template <int D>
struct PipeItemPre {
    double xk0[D], xk1[D], E, bin, bout, irk;
    int nflush;
    uint32_t yseg, ycseg;
};
template <int D, int MODEL>
__device__ __forceinline__ void pipe_item_prologue(const ChainView &c, const PipeBuf &pb, int be, int t,
                                                   int k, int p, int lane, PipeItemPre<D> &q) {
    constexpr int PW = 2 * D + 2;
    const int N = c.N, W = c.W;
#if DLSM_TRIP_PRIO
    __builtin_amdgcn_s_setprio(3);          // from the item's first instruction: its loads leave at once
#endif
    const int jk = be * PP_B + k;
    const double *props = pb.prop + (size_t)t * N * PW;
    const uint32_t *yr = c.ybits + ((size_t)t * N + jk) * W;
    const uint32_t *yc = MODEL == DLSM_DIRECTED ? c.ytbits + ((size_t)t * N + jk) * W : nullptr;
    q.E = pb.consts[0];
    q.nflush = (int)pb.consts[1];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        q.xk0[d] = props[(size_t)jk * PW + D + 2 + d];
        q.xk1[d] = props[(size_t)jk * PW + d];
    }
    q.bin = 0.0; q.bout = 0.0; q.irk = 0.0;
    if (MODEL == DLSM_DIRECTED) {
        q.bin = c.intercept[0]; q.bout = c.intercept[1];
        q.irk = 1.0 / c.radii[jk];
    }
    // the bits of row k for the part's neighbours: lane w holds word (lo >> 5) + w of the row
    const int w0 = (p * pb.per) >> 5;
    q.yseg = yr[min(w0 + lane, W - 1)];
    q.ycseg = MODEL == DLSM_DIRECTED ? yc[min(w0 + lane, W - 1)] : 0u;
}

// trips of 64 neighbours whose operands an undirected item loads up front
__host__ __device__ constexpr int pipe_prefetch_trips(int D) {
    // (d = 3, 4: one trip fewer than would fit on paper - seven / five spilled; d = 5 .. 8: what is left beside the
    // item's own 4 d registers of positions)
    return D == 1 ? 16 : D == 2 ? 11 : D == 3 ? 6 : D == 4 ? 4 : D <= 6 ? 3 : 2;
}

// One wavefront: part p of node k of batch `be` in slice t - the evaluator of the directed model and of the
// undirected model where kernels_pipe_lds.hpp's does not apply (PIPE_UNDIRECTED_LONG).  TP: the trips beyond the
// prefetched ones are software-pipelined.  (Round 6 removed the forms measured and dropped in rounds 3 - 5: rows
// staged in LDS by a persistent launch, the lane's H entry at the head of the item or requested mid-way, and
// their measurement switches - profiles/r04_h_entry_ablation.md holds their numbers.)
template <int D, int MODEL, bool TP, int G>
__device__ __forceinline__ void pipe_eval_item(const ChainView &c, const PipeBuf &pb, int be,
                                               int nb, int t, int k, int p, int lane,
                                               const double *etab, const PipeItemPre<D> &pre
#ifdef DLSM_PIPE_TIMING
                                               , int tl, int tgw
#endif
                                               ) {
    constexpr int PW = 2 * D + 2;
    const int N = c.N, W = c.W;
#ifdef DLSM_PIPE_TIMING
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
#endif
    DLSM_STAMP(0, (double)lane)
    const int j0 = be * PP_B, jk = j0 + k;
    const int jprev = pipe_window_start(be, G) * PP_B;      // nodes >= jprev: snapshot positions
    const int ncross = j0 - jprev;
    const int bb = be & (2 * G - 1);
    const double *Xt = c.X + (size_t)t * N * D;
    const double *props = pb.prop + (size_t)t * N * PW;
    const uint32_t *yr = c.ybits + ((size_t)t * N + jk) * W;
    const uint32_t *yc = MODEL == DLSM_DIRECTED ? c.ytbits + ((size_t)t * N + jk) * W : nullptr;
    const double E = pre.E;
    const int nflush = pre.nflush;
    double xk0[D], xk1[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { xk0[d] = pre.xk0[d]; xk1[d] = pre.xk1[d]; }
    const double bin = pre.bin, bout = pre.bout, irk = pre.irk;
    const double lE = bin + bout;              // log E (directed model)
    const int lo = p * pb.per, hi = min(N, lo + pb.per);
    // neighbours per lane loaded up front (the directed model carries more per neighbour)
    // (TP with the undirected model = parts longer than the prefetch: the software-pipelined trips behind
    // it hold a row in flight - one prefetched trip fewer, or four registers spill)
    constexpr int PP_NPRE = MODEL == DLSM_UNDIRECTED ? pipe_prefetch_trips(D) - (TP ? 1 : 0) : 1;
    // The item is a chain of dependent latencies, so the neighbours' loads are issued before
    // the first use: PP_NPRE per lane (clamped addresses, no predication) - all 11 trips of a
    // part at C2.
    // The bits of row k for the part's neighbours: lane w holds word (lo >> 5) + w of the row
    // (64 words = 2048 neighbours; a longer part reads the rest from memory), handed to the
    // trips as scalar lane masks instead of a load and a register per neighbour.
    double xpre[PP_NPRE][D], rpre[MODEL == DLSM_DIRECTED ? PP_NPRE : 1];
    const uint32_t yseg = pre.yseg, ycseg = pre.ycseg;
    // lo and jprev are multiples of 64, so a trip's 64 neighbours are all on one side of jprev
    // (the clamped ones included: N - 1 >= jprev): the array and its row stride are scalar
    // choices and a lane's address is a 32-bit offset from a scalar base
    auto x_source = [&](int first, int ic, uint32_t &off) -> const char * {
        const bool snap = first >= jprev;
        off = __umul24((uint32_t)ic, (uint32_t)((snap ? PW : D) * sizeof(double)));    // N < 2^24
        return snap ? (const char *)(props + D + 2) : (const char *)Xt;
    };
#pragma unroll
    for (int u = 0; u < PP_NPRE; ++u) {
        const int ic = min(lo + lane + 64 * u, N - 1);
        uint32_t off;
        const char *base = x_source(lo + 64 * u, ic, off);
        coh_load_row<D, false>(base, off, xpre[u]);
        if (MODEL == DLSM_DIRECTED) rpre[u] = c.radii[ic];
    }
    // H entries of the batch: (node kk, entry e), e < ncross + kk: the previous batch (cross
    // block) then the earlier nodes of kk's own batch.  They are dealt out evenly over ALL the
    // lanes working on this (slice, batch) - not to the wavefronts of "their" node, whose
    // entry counts differ by 2x - through the flat list of pipe_h_decode; one entry per lane when
    // there are >= 3 parts.  Their operands are loaded where they are used, after the neighbour
    // loop: the registers a prefetch would hold are worth more as prefetched neighbours
    // (measured: +6 % at C2).
    double acc = 0.0;
    RatioAcc ra;
#define DLSM_PIPE_TERM(XI_, YB_, YCB_, RI_, FLUSH_, SQ_)                                      \
    {                                                                                         \
        if (MODEL == DLSM_UNDIRECTED) {                                                       \
            const double d0_ = dist_fast<D>(XI_, xk0, SQ_);                                   \
            const double d1_ = dist_fast<D>(XI_, xk1, SQ_);                                   \
            ra.lin = fma((YB_) ? 1.0 : 0.0, d0_ - d1_, ra.lin);                               \
            ra.P0 *= fma(E, (SQ_) ? tab_exp11_clamped(-d0_, etab) : tab_exp11(-d0_, etab), 1.0);  \
            ra.P1 *= fma(E, (SQ_) ? tab_exp11_clamped(-d1_, etab) : tab_exp11(-d1_, etab), 1.0);  \
            if (FLUSH_) if (++ra.cnt >= nflush) ra.flush();                                   \
        } else {                                                                              \
            const double d0_ = dist_fast<D>(XI_, xk0, c.squared);                             \
            const double d1_ = dist_fast<D>(XI_, xk1, c.squared);                             \
            const double iri_ = fast_rcp(RI_);                                                \
            pipe_directed_term(ra, acc, d0_, d1_, (int)(YB_), (int)(YCB_),                    \
                               bin * iri_ + bout * irk, bin * irk + bout * iri_, E, lE);      \
        }                                                                                     \
    }
    // the running products of the whole part stay in range without a flush when it has no
    // more than nflush neighbours (the usual case): that loop carries no flush counter
    const bool noflush = MODEL == DLSM_UNDIRECTED && nflush >= hi - lo && !c.squared;
    // Trip u covers neighbours lo + 64 u + lane (lo is a multiple of 64): their bits of row k
    // are words 2u, 2u + 1 of the segment held across the lanes, read into a scalar pair that
    // serves as the lane mask of "y = 1" directly; "i < hi and i != k" is a scalar mask too.
#define DLSM_PIPE_MASKS(U_)                                                                   \
        const int base_ = lo + 64 * (U_);                                                     \
        const int rem_ = hi - base_, self_ = jk - base_;                                      \
        unsigned long long vm_ = rem_ >= 64 ? ~0ull : (rem_ > 0 ? (1ull << rem_) - 1ull : 0ull); \
        if (self_ >= 0 && self_ < 64) vm_ &= ~(1ull << self_);                                \
        const bool in_seg_ = 2 * (U_) + 1 < 64;                                               \
        const int w_ = in_seg_ ? 2 * (U_) : 0;                                                \
        const unsigned long long ym_ =                                                        \
            ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)yseg, w_ + 1) << 32) | \
            (uint32_t)__builtin_amdgcn_readlane((int)yseg, w_);                               \
        const unsigned long long ycm_ = MODEL != DLSM_DIRECTED ? 0ull :                       \
            ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)ycseg, w_ + 1) << 32) | \
            (uint32_t)__builtin_amdgcn_readlane((int)ycseg, w_);                              \
        const int i_ = base_ + lane;                                                          \
        const bool yb_ = in_seg_ ? __builtin_amdgcn_inverse_ballot_w64(ym_)                   \
                                 : (bool)bit_of(yr, min(i_, N - 1));                          \
        const bool ycb_ = MODEL != DLSM_DIRECTED ? false :                                    \
            (in_seg_ ? __builtin_amdgcn_inverse_ballot_w64(ycm_) : (bool)bit_of(yc, min(i_, N - 1)));
#define DLSM_PIPE_REQUEST(U_)                                                                 \
    {                                                                                         \
        const int in_ = min(lo + 64 * (U_) + lane, N - 1);                                    \
        uint32_t off_;                                                                        \
        const char *base_src_ = x_source(lo + 64 * (U_), in_, off_);                          \
        coh_load_row<D, false>(base_src_, off_, xn);                                            \
        if (MODEL == DLSM_DIRECTED) rn = c.radii[in_];                                        \
    }
#define DLSM_PIPE_LOOPS(FLUSH_, SQ_)                                                          \
    _Pragma("unroll")                                                                         \
    for (int u = 0; u < PP_NPRE; ++u) {                                                       \
        DLSM_TRIP_PRIO_STEP(u)                                                                \
        DLSM_PIPE_MASKS(u)                                                                    \
        if (__builtin_amdgcn_inverse_ballot_w64(vm_))                                         \
            DLSM_PIPE_TERM(xpre[u], yb_, ycb_, rpre[MODEL == DLSM_DIRECTED ? u : 0], FLUSH_, SQ_) \
        if (u == 0) { DLSM_STAMP(1, ra.P0) }                                                  \
        if (u == PP_NPRE - 1) { DLSM_STAMP(2, ra.P0) }                                        \
    }                                                                                         \
    /* the trips beyond the prefetched ones.  TP: each trip requests the next one's operands   \
       (clamped address, no predication) before it computes. */                               \
    double xn[D], rn = 1.0;                                                                   \
    if (TP && lo + 64 * PP_NPRE < hi) DLSM_PIPE_REQUEST(PP_NPRE)                                    \
    for (int u = PP_NPRE; lo + 64 * u < hi; ++u) {                                            \
        double xi[D];                                                                         \
        double ri = 1.0;                                                                      \
        if (TP) {                                                                             \
            _Pragma("unroll")                                                                 \
            for (int d = 0; d < D; ++d) xi[d] = xn[d];                                        \
            ri = rn;                                                                          \
            DLSM_PIPE_REQUEST(u + 1)                                                          \
        }                                                                                     \
        DLSM_PIPE_MASKS(u)                                                                    \
        if (__builtin_amdgcn_inverse_ballot_w64(vm_)) {                                       \
            if (!TP) {                                                                        \
                uint32_t off_;                                                                \
                const char *base_src_ = x_source(base_, i_, off_);                            \
                coh_load_row<D, false>(base_src_, off_, xi);                                    \
                if (MODEL == DLSM_DIRECTED) ri = c.radii[min(i_, N - 1)];                     \
            }                                                                                 \
            DLSM_PIPE_TERM(xi, yb_, ycb_, ri, FLUSH_, SQ_)                                    \
        }                                                                                     \
    }
    if (noflush) { DLSM_PIPE_LOOPS(false, 0) }
    else if (c.squared) { DLSM_PIPE_LOOPS(true, 1) }
    else { DLSM_PIPE_LOOPS(true, 0) }
#undef DLSM_PIPE_LOOPS
#undef DLSM_PIPE_REQUEST
#undef DLSM_PIPE_MASKS
#undef DLSM_PIPE_TERM
    pipe_item_finish<D, MODEL, G>(c, pb, be, nb, t, k, p, lane, etab, acc, ra, noflush
#ifdef DLSM_PIPE_TIMING
                                       , ts
#endif
                                       );
#ifdef DLSM_PIPE_TIMING
    if (lane == 0 && tl >= 0 && tl < 24 && tgw < 4096)
        for (int i = 0; i < 6; ++i) g_pipe_item_t[tl][tgw][i] = ts[i];
#endif
}


// Case-control likelihood (a3 inside a9 / a10): the O(deg + 2C) gathered terms of a node
// as in k_spec_eval_cc, with the snapshot rule for the neighbours' positions, split over
// the CC_PARTS wavefronts of a group: wavefront `sub` takes the 64-term chunks sub,
// sub + CC_PARTS, ...  H is non-zero only for the nodes of the window [jprev, jk) that sit in
// node k's edge / control lists (at most once per direction): their corrections are added into
// the group's LDS column (two addends at most per entry: order independent) and
// exponentiated on the way out; every other entry of the column is the factor 1.
//   term kinds: 0 in-edge, 1 out-edge (eta - softplus(eta)),
//               2 in-control, 3 out-control (- adj * softplus(eta))
constexpr int CC_PARTS = 4;

template <int D>
__device__ __forceinline__ void pipe_cc_accumulate(const ChainView &c, const PipeBuf &pb, int be,
                                                   int t, int k, int sub, int lane, double *col) {
    constexpr int PW = 2 * D + 2;
    const int N = c.N;
    const int j0 = be * PP_B, jk = j0 + k;
    const int jprev = max(0, j0 - PP_B);
    const int bb = be & 1;
    const size_t node = (size_t)t * N + jk;
    const double *Xt = c.X + (size_t)t * N * D;
    const double *props = pb.prop + (size_t)t * N * PW;
    double xk0[D], xk1[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        xk0[d] = props[(size_t)jk * PW + D + 2 + d];
        xk1[d] = props[(size_t)jk * PW + d];
    }
    const double bin = c.intercept[0], bout = c.intercept[1];
    const double rj = c.radii[jk];
    const int in_deg = c.degree[node * 2], out_deg = c.degree[node * 2 + 1];
    const int nci = pb.nctrl[node * 2], nco = pb.nctrl[node * 2 + 1];
    const double adj_in = (double)(N - in_deg - 1) / (double)nci;
    const double adj_out = (double)(N - out_deg - 1) / (double)nco;
    const int total_terms = in_deg + out_deg + nci + nco;
    double acc = 0.0;
    for (int q0 = 64 * sub; q0 < total_terms; q0 += 64 * CC_PARTS) {
        const int q = q0 + lane;
        int e = -1, kind = 0;
        if (q < total_terms) {
            int r = q;
            if (r < in_deg) { e = c.in_edges[node * c.Din + r]; kind = 0; }
            else if ((r -= in_deg) < out_deg) { e = c.out_edges[node * c.Dout + r]; kind = 1; }
            else if ((r -= out_deg) < nci) { e = c.ctrl_in[node * c.C + r]; kind = 2; }
            else { r -= nci; e = c.ctrl_out[node * c.C + r]; kind = 3; }
        }
        if (e >= 0) {
            const double *src = e < jprev ? Xt + (size_t)e * D : props + (size_t)e * PW + D + 2;
            double xe[D];
#pragma unroll
            for (int d = 0; d < D; ++d) xe[d] = src[d];
            const double re = c.radii[e];
            const bool in_dir = (kind == 0 || kind == 2);
            const double wsp = kind < 2 ? 1.0 : (kind == 2 ? adj_in : adj_out);
            // delta of this term when k moves, the neighbour at position xn
#define DLSM_CC_DELTA(OUT_, XN_, SELF_)                                                        \
            {                                                                                   \
                const double d0_ = (SELF_) ? 0.0 : dist_of<D>(XN_, xk0, c.squared);             \
                const double d1_ = (SELF_) ? 0.0 : dist_of<D>(XN_, xk1, c.squared);             \
                const double e0_ = in_dir ? bin * (1 - d0_ / rj) + bout * (1 - d0_ / re)        \
                                          : bin * (1 - d0_ / re) + bout * (1 - d0_ / rj);       \
                const double e1_ = in_dir ? bin * (1 - d1_ / rj) + bout * (1 - d1_ / re)        \
                                          : bin * (1 - d1_ / re) + bout * (1 - d1_ / rj);       \
                const double sp_ = log((1.0 + exp(e1_)) / (1.0 + exp(e0_)));                    \
                OUT_ = (kind < 2 ? (e1_ - e0_) : 0.0) - wsp * sp_;                              \
            }
            double contrib;
            DLSM_CC_DELTA(contrib, xe, e == jk)
            acc += contrib;
            if (e >= jprev && e < jk) {         // a node of the window: its acceptance matters
                double xe1[D], moved;
#pragma unroll
                for (int d = 0; d < D; ++d) xe1[d] = props[(size_t)e * PW + d];
                DLSM_CC_DELTA(moved, xe1, false)
                atomicAdd(&col[e - jprev], moved - contrib);
            }
#undef DLSM_CC_DELTA
        }
    }
    const double total = wave_sum_all(acc);
    if (lane == 0) {
        double2 *f = (double2 *)pb.full0 + (((size_t)bb * c.T + t) * PP_B + k) * pb.parts + sub;
        *f = make_double2(total, 1.0);
    }
}

// the group's column -> factors; wavefront `sub` writes entries [64 sub, 64 sub + 64)
template <int D>
__device__ __forceinline__ void pipe_cc_writeout(const ChainView &c, const PipeBuf &pb, int be,
                                                 int t, int k, int sub, int lane,
                                                 const double *col) {
    const int j0 = be * PP_B;
    const int ncross = j0 - max(0, j0 - PP_B);
    const int bb = be & 1;
    const int m = 64 * sub + lane;              // index into [cross block | own batch]
    if (m >= ncross + k) return;
    const double v = col[m];
    const double f = v == 0.0 ? 1.0 : exp(v);
    if (m < ncross) pb.Hx[(((size_t)bb * c.T + t) * PP_B + m) * PP_B + k] = f;    // G = 1
    else pb.Hd[(((size_t)bb * c.T + t) * PP_B + (m - ncross)) * PP_B + k] = f;
}

// Resolve batch b of slice t: the fixed-point solve of k_spec_resolve for one batch, the
// acceptances of its window's earlier batches (pipe_window_start) entering through gathered rows
// of the cross block.  With G > 1 a workgroup resolves G batches one after the other; the list
// of the batch it has just resolved is in sOwn (own_prev), the others come from memory.
template <int D, int G>
__device__ __forceinline__ void pipe_resolve(const ChainView &c, const PipeBuf &pb, int b, int t,
                                             double *sH, double *sPart,
                                             unsigned long long (*sMask)[2], int *sPrev,
                                             unsigned char *sSat, int *sOwn, bool own_prev
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
    constexpr int G2 = 2 * G;
    const int bb = b & (G2 - 1);
    const int half = wave & 1, part = wave >> 1;
    const int k = 64 * half + lane;
    const bool owner = wave < 2;
    const bool valid = k < nb;
    const double *Hd = pb.Hd + ((size_t)bb * c.T + t) * PP_B * PP_B;
    const double *Hx = pb.Hx + ((size_t)bb * c.T + t) * ((G2 - 1) * PP_B) * PP_B;
    int32_t *acct = pb.acc + (size_t)t * G2 * PP_ACC;
    int32_t *accg = acct + (size_t)bb * PP_ACC;                // this batch's list
    const int ws = pipe_window_start(b, G);
    const int nwin = b - ws;                                    // earlier batches of the window: <= 3
    // diagonal block -> LDS (unconditional clamped loads, see k_spec_resolve)
    double2 blk[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int q = min(u * PP_THREADS + tid, nb * (PP_B / 2) - 1);
        blk[u] = coh_load2<false>(Hd, (uint32_t)(((q >> 6) * PP_B + 2 * (q & 63)) * sizeof(double)));
    }
    // their accepted nodes as rows of the cross block: row = 128 (batch - ws) + node
    int cntw[3] = {0, 0, 0};
#pragma unroll
    for (int w = 0; w < G2 - 1; ++w)
        if (w < nwin) {
            const bool own = own_prev && ws + w == b - 1;
            cntw[w] = own ? sOwn[0] : acct[(size_t)((ws + w) & (G2 - 1)) * PP_ACC];
        }
    const int nprev = cntw[0] + cntw[1] + cntw[2];
#pragma unroll
    for (int w = 0; w < G2 - 1; ++w)
        if (w < nwin) {
            const bool own = own_prev && ws + w == b - 1;
            const int32_t *lst = own ? sOwn : acct + (size_t)((ws + w) & (G2 - 1)) * PP_ACC;
            const int off = w == 0 ? 0 : (w == 1 ? cntw[0] : cntw[0] + cntw[1]);
            for (int a = tid; a < cntw[w]; a += PP_THREADS) sPrev[off + a] = w * PP_B + lst[1 + a];
        }
    // multiplicative domain: r = exp(log-ratio of node k), lu = its uniform draw.  A node whose
    // log-ratio is beyond +-700 (exp would saturate) is resolved in the log domain instead -
    // log u against lr + the LOGS of its H factors - so that the decision is the sequential
    // scan's for every chain, not only for those whose single-node moves stay below 700 nats.
    double r = 0.0, lu = 0.0, st = 0.0, x1[D], lr = 0.0;
    bool sat = false;
    int32_t na = 0, ns = 0, un = 0;
#pragma unroll
    for (int d = 0; d < D; ++d) x1[d] = 0.0;
    if (owner) {
        const int kc = min(k, nb - 1);
        const double2 *f = (const double2 *)pb.full0 + ((size_t)bb * c.T + t) * PP_B * pb.parts;
        const int p1 = pb.parts;
        double2 tv[PP_MAXPARTS];
#pragma unroll
        for (int u = 0; u < PP_MAXPARTS; ++u)
            tv[u] = coh_load2<false>(f, (uint32_t)((kc * p1 + min(u, p1 - 1)) * sizeof(double2)));
        double tot = tv[0].x, pr_ = tv[0].y;
#pragma unroll
        for (int u = 1; u < PP_MAXPARTS; ++u) {
            tot += u < p1 ? tv[u].x : 0.0;
            pr_ *= u < p1 ? tv[u].y : 1.0;
        }
        const double *pr = pb.prop + ((size_t)t * N + j0 + kc) * PW;
        double x0[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { x1[d] = pr[d]; x0[d] = pr[D + 2 + d]; }
        // prior terms of the step's logp closure, with the neighbouring slices as they
        // are now (see the header: the odd slices run one batch behind)
        const double prior = node_log_prior<D, false>(c, t, j0 + kc, x1) -
                             node_log_prior<D, false>(c, t, j0 + kc, x0);
        const double ek = tot + prior;
        sat = !(fabs(ek) <= 700.0);
        r = sat ? 1.0 : exp(ek) * pr_;
        lu = pr[D];
        if (sat) { lr = ek + log(pr_); lu = log(lu); }
        sSat[k] = sat ? 1 : 0;
        const size_t tjc = (size_t)t * N + j0 + kc;
        st = c.step[tjc]; na = c.nacc[tjc]; ns = c.nsteps[tjc]; un = c.until[tjc];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
        ((double2 *)sH)[min(u * PP_THREADS + tid, nb * (PP_B / 2) - 1)] = blk[u];
    __syncthreads();                                   // sPrev, sH visible
    DLSM_STAMP(1, (double)tid)
    const bool satk = sSat[k] != 0;                    // column k is resolved in the log domain
    const bool anysat = __ballot(satk) != 0ull;        // (practically never: wave-uniform slow path)
    if (nprev > 0) {
        const double *colp = Hx + min(k, PP_B - 1);
        double prod = 1.0;
        int a = part;
        if (!anysat) {
            // PP_B / 8 = 16 rows per thread and trip (one trip per 128 accepted nodes): all their
            // loads in flight together (clamped addresses, the factor of a row that is not there
            // replaced by 1)
            for (int base = 0; base < nprev; base += PP_B) {
                double hh[PP_B / 8];
#pragma unroll
                for (int u = 0; u < PP_B / 8; ++u)
                    hh[u] = coh_load<false>(colp + (size_t)sPrev[min(base + a + 8 * u, nprev - 1)] * PP_B);
#pragma unroll
                for (int u = 0; u < PP_B / 8; ++u) prod *= base + a + 8 * u < nprev ? hh[u] : 1.0;
            }
        } else {
            double lsum = 0.0;
            for (; a < nprev; a += 8) {
                const double h = coh_load<false>(colp + (size_t)sPrev[a] * PP_B);
                if (satk) lsum += log(h); else prod *= h;
            }
            if (satk) prod = lsum;
        }
        sPart[wave * 64 + lane] = prod;
        __syncthreads();
        if (owner) {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const double v = sPart[(2 * p + half) * 64 + lane];
                if (sat) lr += v; else r *= v;
            }
        }
        __syncthreads();
    }
    if (owner) {
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
                for (int u = 0; u < 4; ++u) h[u] = col[(f[u] < (1 << 20) ? f[u] : 0) * PP_B];
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
        // accepted nodes of this batch (ascending) for the cross terms of the batches that have it
        // in their window: in memory for the launches to come, in LDS for this workgroup's next batch
        if (accepted) {
            const int base = half == 0 ? 0 : __popcll(m0);
            const int at = 1 + base + __popcll(mine & ((1ull << lane) - 1ull));
            accg[at] = k;
            sOwn[at] = k;
        }
        if (tid == 0) {
            const int cnt = __popcll(m0) + __popcll(m1);
            accg[0] = cnt;
            sOwn[0] = cnt;
        }
    }
#ifdef DLSM_PIPE_TIMING
    DLSM_STAMP(4, (double)cur)
    if (tid == 0 && tl >= 0 && tl < 24 && t < 32)
        for (int i = 0; i < 5; ++i) g_pipe_res_t[tl][t][i] = ts[i];
#endif
}

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
