// Pipelined speculative sweep in ONE launch (algo 7): the roles of k_pipe_step
// (kernels_spec_pipe.hpp) kept alive for the whole sweep, the kernel boundary between two batches
// replaced by per-slice flags.
//
// The launch-per-batch form pays, 18 times per sweep at T = 10, N = 2000, a dispatch floor and a
// cold start (wavefront launch, scalar loads, first trip to memory behind the boundary's cache
// invalidation) - about a third of its 12 us period - and makes every slice wait for the slowest
// role of every batch.  Here
//
//     workgroup t < T          resolves batches 0, 1, .. of slice t, one after the other;
//     workgroups >= T          evaluate (slice, batch, 16 nodes, part) groups, one item per
//                              wavefront, drawn in batch order from a ticket counter;
//
// and the only waits are the true dependencies of the scheme (header of kernels_spec_pipe.hpp):
//
//     eval(t, b)      needs  resolve(t, b - 2)     the positions of batches <= b - 2 are final
//     resolve(t, b)   needs  eval(t, b) complete   and resolve(t, b - 1): program order
//     resolve(odd t, b) needs resolve(t +- 1, b)   even-then-odd order of the prior's neighbours
//
// Even slices never wait for odd ones, tickets are handed out in dependency order and only to
// workgroups that are running, so the lowest unfinished ticket always belongs to a running
// workgroup whose dependencies are complete: the launch makes progress with ANY number of
// resident evaluator workgroups (a shared device only slows it down).  Every wait is bounded all
// the same: a poll budget, then a sticky error word that makes every role leave, reported by the
// host as DLSM_E_HIP.
//
// Hand-offs (MI355X guide, inter-workgroup visibility; helpers in device_common.hpp): handed-off
// bytes are stored and loaded `sc1` on both sides - final positions (resolver -> evaluators and
// the neighbouring slices' resolvers), (sum, product) records and H factors (evaluators ->
// resolver).  A resolver announces batch b with one sc1 flag store behind a workgroup barrier that
// follows every storing wavefront's s_waitcnt vmcnt(0); an evaluator wavefront drains its stores,
// bumps a counter in LDS, and the wavefront whose bump completes the group adds the group's items
// to the slice's per-batch counter (agent-scope atomic).  Consumers poll with relaxed sc1 loads
// from ONE wavefront, the others start behind a workgroup barrier.
//
// Decisions are those of algo 4 bit for bit: the items and the fixed-point solve are the same
// code (pipe_eval_item / pipe_resolve with COH = true), only their scheduling differs.
#pragma once
#include "kernels_spec_pipe.hpp"

namespace dlsm {

constexpr int PS_STRIDE = 16;            // int32 words between two flags: one 64-byte line each
constexpr int PS_KGROUP = PP_WAVES;      // nodes of a ticket: one per wavefront
constexpr int PS_GROUPS = PP_B / PS_KGROUP;

struct PipeSync {
    int32_t *words;      // [1 + T + T nbat][PS_STRIDE] : ticket counter, resolved[t], done[t][b]
    int32_t *err;        // sticky: a wait ran out of its budget
    int spin_budget;     // polls per wait
    __device__ __forceinline__ int32_t *queue() const { return words; }
    __device__ __forceinline__ int32_t *resolved(int t) const { return words + (size_t)(1 + t) * PS_STRIDE; }
    __device__ __forceinline__ int32_t *done(int T, int nbat, int t, int b) const {
        return words + (size_t)(1 + T + t * nbat + b) * PS_STRIDE;
    }
};

// Called by a whole wavefront: lane i waits for *word >= want (word == nullptr: nothing to wait
// for).  Relaxed sc1 polls with a sleep in between, bounded.  false = the budget ran out here or
// elsewhere (the error word is set): the caller leaves the kernel.
__device__ __forceinline__ bool pipe_wait(const int32_t *word, int want, const PipeSync &ps) {
    bool ok = word == nullptr;
    for (int spins = 0;; ++spins) {
        if (!ok) ok = coh_load_i32(word) >= want;
        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) return true;
        if (spins >= ps.spin_budget) {
            if (!ok) coh_store_i32(ps.err, 1);
            return false;
        }
        if ((spins & 255) == 255 && coh_load_i32(ps.err) != 0) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

#ifdef DLSM_PIPE_TIMING
// phase stamps of the persistent launch: resolver t, batch b: {wait start, wait end, resolve end,
// published}; evaluator workgroup w, round r (its wavefront 0): {round start, poll matched +
// barrier, item done and drained, ticket, items of the group}
__device__ unsigned long long g_persist_res_t[32][24][4];
__device__ unsigned long long g_persist_ev_t[256][24][5];
__device__ __forceinline__ unsigned long long persist_clock() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}
#endif

// ---- the resolver of the persistent launch ------------------------------------------------------
// pipe_resolve's fixed-point solve (same system, same passes) with its memory side rebuilt for a
// workgroup that stays on its CU:
//   * the evaluators file the H factors of batch b as one row of 2 PP_B doubles per node k -
//     [window nodes | own batch's nodes] - so their stores are contiguous (write-through 8-byte
//     stores down a column were 245 000 partial-line writes per step) and every load here is a
//     coalesced 16-byte one;
//   * the WHOLE cross block comes in with the first burst of loads (16 factors per thread, thread
//     = (node k, eighth of the window)) and the previous batch's acceptances select among them as
//     a bit mask held in LDS: no list, and no gather that could only start once that list existed;
//   * the diagonal block sits in LDS by rows of k with an odd stride, so that the passes' reads
//     (lane = k, row f fixed) fall into different banks.
// Decisions are pipe_resolve's up to the order in which a node's cross factors are multiplied.
// (PR_LD, group8_prod / group8_sum and the resolver itself - row_resolve<D, COH = true> - live in
// kernels_spec_pipe.hpp: the launch-per-batch kernels use the same resolver with plain loads)

// ticket -> (slice t, evaluated batch be, its size nb, group of 16 nodes kg, part p); false when
// the ticket is past the end or its slice has no batch at that step.  Wave-uniform; quotients
// through float reciprocals and one fix-up (tickets < 2^20).
__device__ __forceinline__ bool persist_decode(int ticket, int total, int gps, float inv_gps, int T,
                                               float inv_T, int nE, int nbat, int N, int &t, int &be,
                                               int &nb, int &kg, int &p) {
    int s = (int)(((float)ticket + 0.5f) * inv_gps);
    s -= (s * gps > ticket); s += ((s + 1) * gps <= ticket);
    const int g = ticket - s * gps;
    const int r = g / PS_GROUPS;
    kg = g - r * PS_GROUPS;
    p = (int)(((float)r + 0.5f) * inv_T);
    p -= (p * T > r); p += ((p + 1) * T <= r);
    const int si = r - p * T;
    t = si < nE ? 2 * si : 2 * (si - nE) + 1;
    const int l = s - 1;
    be = (t & 1) ? l : l + 1;
    const bool live = ticket < total && be >= 0 && be < nbat;
    nb = live ? min(PP_B, N - be * PP_B) : 0;
    return live;
}

// ---- the evaluator's item with its neighbours staged in LDS ---------------------------------
// The 16 wavefronts of a round work on 16 nodes of ONE (slice, part): they all need the same
// `per` neighbour rows.  Read past the L1 by every wavefront (the final positions are handed over
// inside the launch) those rows cost 16 trips to the L2 each - 3.7 us before the first operands
// of an item arrived, against 1.3 us in the launch-per-batch kernel, whose wavefronts share them
// through the L1.  So the workgroup stages them once per round: thread r loads row lo + r - its
// final position (sc1) when the node's batch is resolved, else the snapshot the propose kernel
// took (plain) - and the items (pipe_eval_item with LDSX) read them from LDS, one ds_read_b128 per
// trip at d = 2.
template <int D>
__device__ __forceinline__ void persist_stage_rows(const ChainView &c, const PipeBuf &pb, int t,
                                                   int be, int p, double *sX, int tid) {
    constexpr int PW = 2 * D + 2;
    const int N = c.N;
    const int jprev = pipe_window_start(be, 1) * PP_B;       // nodes >= jprev: snapshot positions
    const int lo = p * pb.per;
    const int n = min(pb.per, N - lo);
    const double *Xt = c.X + (size_t)t * N * D;
    const double *props = pb.prop + (size_t)t * N * PW;
    // lo and jprev are multiples of 64: a wavefront's 64 rows lie on one side of jprev
    for (int r = tid; r < n; r += PP_THREADS) {
        const int i = lo + r;
        double x[D];
        if (i >= jprev) {
            const double *src = props + (size_t)i * PW + D + 2;
#pragma unroll
            for (int d = 0; d < D; ++d) x[d] = src[d];
        } else {
            coh_load_row<D, true>(Xt, (uint32_t)i * (uint32_t)(D * sizeof(double)), x);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) sX[r * D + d] = x[d];
    }
}

template <int D, int MODEL_>
__global__ __launch_bounds__(PP_THREADS) void k_pipe_persist(ChainView c, PipeBuf pb, PipeSync ps) {
    constexpr int MODEL = MODEL_ == PIPE_UNDIRECTED_LONG ? DLSM_UNDIRECTED : MODEL_;
    // resolvers: the diagonal block, PP_B rows of PR_LD; evaluators: exp table + the round's rows
    extern __shared__ __attribute__((aligned(16))) double pp_sH[];
    __shared__ double sPart[PP_WAVES * 64];
    __shared__ double sCross[PP_B];
    __shared__ unsigned long long sMask[2][2];
    __shared__ unsigned long long sMaskPrev[2];
    __shared__ unsigned long long sSatMask[2];
    __shared__ double sTab[EXPTAB_N];                  // the resolvers' exp table (the evaluators' is in pp_sH)
    __shared__ int sGo[2], sTicket[2], sCnt;
    const int T = c.T, N = c.N, nbat = pb.nbat;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int last = T > 1 ? nbat : nbat - 1;      // steps l = -1 .. last, as the launches of algo 4
    // an earlier sweep of this chain gave up: leave (one load per workgroup, so that its
    // wavefronts agree)
    if (tid == 0) sGo[1] = coh_load_i32(ps.err) == 0 ? 1 : 0;
    if (tid == 0) sCnt = 0;
    if ((int)blockIdx.x < T) exp_table_fill(sTab, tid);
    __syncthreads();
    if (!sGo[1]) return;
    if ((int)blockIdx.x < T) {
        // ---- resolver of slice t -------------------------------------------------------------
        const int t = blockIdx.x;
        for (int l = -1; l <= last; ++l) {
            const int b = l - (t & 1);
            if (b < 0 || b >= nbat) continue;
            const int nb = min(PP_B, N - b * PP_B);
#ifdef DLSM_PIPE_TIMING
            unsigned long long ts0 = persist_clock(), ts1 = 0, ts2 = 0;
#endif
            if (wave == 0) {
                const int32_t *word = nullptr;
                int want = 0;
                if (lane == 0) { word = ps.done(T, nbat, t, b); want = nb * pb.parts; }
                else if ((t & 1) && lane == 1) { word = ps.resolved(t - 1); want = b + 1; }
                else if ((t & 1) && lane == 2 && t + 1 < T) { word = ps.resolved(t + 1); want = b + 1; }
                const bool ok = pipe_wait(word, want, ps);
                if (lane == 0) sGo[0] = ok ? 1 : 0;
            }
            __syncthreads();
            if (!sGo[0]) return;
#ifdef DLSM_PIPE_TIMING
            ts1 = persist_clock();
#endif
            row_resolve<D, true>(c, pb, b, t, pp_sH, sPart, sMask, sMaskPrev, sCross, sSatMask, sTab
#ifdef DLSM_PIPE_TIMING
                               , b
#endif
                               );
            drain_vmem();                      // this wavefront's position stores have left
#ifdef DLSM_PIPE_TIMING
            ts2 = persist_clock();
#endif
            __syncthreads();                   // ... every wavefront's; the LDS of this batch is free
            if (tid == 0) {
                coh_store_i32(ps.resolved(t), b + 1);
#ifdef DLSM_PIPE_TIMING
                if (t < 32 && b < 24) {
                    g_persist_res_t[t][b][0] = ts0; g_persist_res_t[t][b][1] = ts1;
                    g_persist_res_t[t][b][2] = ts2; g_persist_res_t[t][b][3] = persist_clock();
                }
#endif
            }
        }
        return;
    }
    // ---- evaluators ------------------------------------------------------------------------------
    // ticket -> (step s = l + 1, part p, slice, group of 16 nodes); slices in the order evens, odds
    if (MODEL == DLSM_UNDIRECTED) exp_table11_fill<PP_THREADS>(pp_sH, tid);     // visible behind the first round's barrier
    const int nE = (T + 1) / 2;
    const int gps = pb.parts * T * PS_GROUPS;             // tickets per step
    const int total = (last + 2) * gps;
    const float inv_gps = 1.0f / (float)gps, inv_T = 1.0f / (float)T;
    // A round: wavefront 0 decodes its ticket, requests the next one (it travels while this one is
    // worked on), polls the slice's flag and posts (ticket, go) in LDS; behind the barrier the
    // workgroup stages the part's neighbour rows, behind a second one the items run.  Slots
    // alternate by round parity.
    // (Requesting everything that does not depend on the awaited batch - the item's prologue, the
    // operands of its first H entry, the rows of the other batches - BEFORE the poll matched was
    // built and measured: it puts those loads on top of the resolvers' bandwidth-bound block loads
    // and lost 10 %: profiles/r03_persist_notes.md.)
    double *sX = pp_sH + EXPTAB11_N;                       // the round's neighbour rows, behind the exp table
    constexpr int IM = MODEL == DLSM_DIRECTED_CASE_CONTROL ? DLSM_DIRECTED : MODEL;
    int tk0 = (int)blockIdx.x - T;                         // wavefront 0's copy; the counter starts behind these
    for (int round = 0;; ++round) {
        const int par = round & 1;
        int next = 0;
#ifdef DLSM_PIPE_TIMING
        const unsigned long long tsr = persist_clock();
#endif
        if (wave == 0) {
            int t_, be_, nb_, kg_, p_;
            const bool live = persist_decode(tk0, total, gps, inv_gps, T, inv_T, nE, nbat, N, t_, be_, nb_, kg_, p_);
            if (lane == 0 && tk0 < total)
                next = __hip_atomic_fetch_add(ps.queue(), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ok = true;
            if (live && nb_ > kg_ * PS_KGROUP && be_ >= 2)
                ok = pipe_wait(lane == 0 ? ps.resolved(t_) : nullptr, be_ - 1, ps);
            if (lane == 0) { sGo[par] = ok ? 1 : 0; sTicket[par] = tk0; }
        }
        __syncthreads();                       // the poll has matched: everyone may load
        const int ticket = sTicket[par];
        if (ticket >= total || !sGo[par]) return;
        int t, be, nb, kg, p;
        persist_decode(ticket, total, gps, inv_gps, T, inv_T, nE, nbat, N, t, be, nb, kg, p);
        const int k = kg * PS_KGROUP + wave;
        const int nvalid = min(max(nb - kg * PS_KGROUP, 0), PS_KGROUP);
#ifdef DLSM_PIPE_TIMING
        const unsigned long long ts0 = persist_clock();
#endif
        if (nvalid > 0) persist_stage_rows<D>(c, pb, t, be, p, sX, tid);
        __syncthreads();
        if (k < nb) {
            PipeItemPre<D> pre;
            PipeHPre<D> nohp;                  // (unused: HPF = false)
            pipe_item_prologue<D, IM>(c, pb, be, t, k, p, lane, pre);
            pipe_eval_item<D, IM, false, 1, true, true>(c, pb, be, nb, t, k, p, lane, pp_sH, sX, pre, nohp
#ifdef DLSM_PIPE_TIMING
                , round, ((int)blockIdx.x - T) * PP_WAVES + wave
#endif
                );
        }
        if (nvalid > 0) {
            drain_vmem();                      // records and H factors of this wavefront have left
            if (lane == 0) {
                const int old = atomicAdd(&sCnt, 1);
                if ((old & (PP_WAVES - 1)) == PP_WAVES - 1)        // the group's last wavefront
                    __hip_atomic_fetch_add(ps.done(T, nbat, t, be), nvalid, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#ifdef DLSM_PIPE_TIMING
        if (tid == 0 && round < 24) {
            const int w = (int)blockIdx.x - T;
            if (w < 256) {
                g_persist_ev_t[w][round][0] = tsr; g_persist_ev_t[w][round][1] = ts0;
                g_persist_ev_t[w][round][2] = persist_clock();
                g_persist_ev_t[w][round][3] = (unsigned long long)ticket;
                g_persist_ev_t[w][round][4] = (unsigned long long)nvalid;
            }
        }
#endif
        if (wave == 0) tk0 = __builtin_amdgcn_readfirstlane(next);
    }
}

}  // namespace dlsm
