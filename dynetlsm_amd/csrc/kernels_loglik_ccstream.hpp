// Case-control full log-likelihood (a6, directed_likelihoods_fast.pyx:208-270) as a STREAM of term rows per
// wavefront (round 6).  k_loglik_casecontrol_rows gives a wavefront two rows and ends: 25 000 wavefronts at
// config 4 executing 39.7 M / 16.6 M vector instructions per pass (four / one candidate: 81.5 / 42.1 us, 65 / 27 of
// them vector issue - profiles/r06_cc_pass_notes.md).  Here
//   * a launch is one wave of RESIDENT wavefronts (workgroups per CU from the occupancy query, the grid trimmed by
//     ccs_plan.hpp); a wavefront walks a contiguous share of its slice's ENTRIES (cc_rows.hpp, k_cc_order: rows by
//     descending out-degree, cut into entries of at most two 64-term trips) and keeps PD entries' records and NB
//     entries' indices in flight while it computes one:
//       step j :  request B[j + PD] (records, by A[j + PD]'s indices) -> request A[j + PD + NB] -> compute B[j]
//     - table fill, barrier and workgroup sum once per ~15 entries;
//   * equal out-degrees lie side by side in the order and have the same control weight adj_out, so ONE running
//     product of control factors lives across rows and becomes a logarithm per run of equal weights
//     (ccs_row_weight), not per row and candidate;
//   * the arithmetic per term is that of the rows kernel with the bookkeeping moved out of the term:
//       eta_m = b_in,m (1 - d / r_q) + b_out,m (1 - d / r_i): the two brackets once per term, not per candidate;
//       eta > 130 (log(1 + e^eta) = eta) is a wave-level cold path, not four selects per candidate;
//       edge / control lanes by exec mask (a trip behind the out-edges has no edge lane at all);
//       the running products are tested for overflow once per entry (an entry multiplies a product by at most
//       (1 + e^130)^2 = 1e113); no separate sum for "big" control terms: they go to L with the row's weight;
//   * NT = 1024: the reciprocal radii of all nodes in LDS, ONE 16-byte position gathered per term (below).
// One record of M sums per workgroup; entries, shares and order fixed by (grid, rows): the same bits every launch.
#pragma once
#include "kernels_loglik.hpp"

namespace dlsm {

constexpr int LLCS_THREADS = 256;       // four wavefronts: one per SIMD
constexpr int LLCS_NS = 2;              // 64-term trips of an entry of the walking order (ccs_plan.hpp: CC_ENT_TERMS)

__device__ __forceinline__ double uniform_d(double v) {     // a wave-uniform double into scalar registers
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// one 64-term trip of a row: partner records (xq, 1 / r_q) against the row's own (xi, 1 / r_i);
// od / nv: edge lanes and valid lanes of this trip (lane < od: out-edge, od <= lane < nv: out-control)
template <int D, int M, bool TWO>
__device__ __forceinline__ void ccs_trip(const double *xq, double irq0, double irq1, const double *xi,
                                         double iri0, double iri1, const double *bin, const double *bout,
                                         int od, int nv, int lane, int squared, double adj,
                                         const double *sTab, double *L, double *Pe, double *Pc,
                                         bool live = true) {
    const double dd = dist_fast<D>(xq, xi, squared);
    const double u0 = fma(-dd, irq0, 1.0), v0 = fma(-dd, iri0, 1.0);
    const double u1 = TWO ? fma(-dd, irq1, 1.0) : u0, v1 = TWO ? fma(-dd, iri1, 1.0) : v0;
    double eta[M];
#pragma unroll
    for (int m = 0; m < M; ++m) eta[m] = fma(bin[m], m == 0 ? u0 : u1, bout[m] * (m == 0 ? v0 : v1));
    double emax = eta[0];
#pragma unroll
    for (int m = 1; m < M; ++m) emax = fmax(emax, eta[m]);
    const bool is_e = live && lane < od, is_c = live && lane >= od && lane < nv;
    if (__builtin_amdgcn_ballot_w64((is_e || is_c) && emax > 130.0) == 0) {
        double f[M];
#pragma unroll
        for (int m = 0; m < M; ++m) f[m] = 1.0 + tab_exp(fmax(eta[m], -700.0), sTab);
        if (od > 0) {                                   // wave-uniform: most trips hold no edge
            if (is_e) {
#pragma unroll
                for (int m = 0; m < M; ++m) { L[m] += eta[m]; Pe[m] *= f[m]; }
            }
        }
        if (is_c) {
#pragma unroll
            for (int m = 0; m < M; ++m) Pc[m] *= f[m];
        }
    } else {
        // log(1 + e^eta) = eta beyond 130: such a term adds nothing to an edge's sum and eta (times the row's
        // weight) to a control's.  (The empty asm keeps this block where it is: without it the compiler computed
        // the block in front of the branch in every trip - 200 of a four-candidate trip's 300 vector instructions.)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const bool big = eta[m] > 130.0;
            const double f = big ? 1.0 : 1.0 + tab_exp(fmin(fmax(eta[m], -700.0), 130.0), sTab);
            if (is_e) { L[m] += big ? 0.0 : eta[m]; Pe[m] *= f; }
            if (is_c) { L[m] -= big ? adj * eta[m] : 0.0; Pc[m] *= f; }
        }
    }
}

template <int M>
__device__ __forceinline__ void ccs_flush(double *L, double *P, double w, double limit) {
#pragma unroll
    for (int m = 0; m < M; ++m)
        if (__builtin_amdgcn_ballot_w64(P[m] > limit))
            if (P[m] > limit) { L[m] -= w * fast_log(P[m]); P[m] = 1.0; }
}

// the running product of a wavefront's control factors: kept across rows while the rows' weight adj_out stays the
// same (the walking order - k_cc_order - puts equal out-degrees side by side), turned into a logarithm when it
// changes.  Before a row's first two trips the product is at most 1e100, behind them at most 1e213.
template <int M>
__device__ __forceinline__ void ccs_row_weight(double *L, double *Pc, double &adj_cur, double adj) {
    const bool same = __double2loint(adj) == __double2loint(adj_cur) && __double2hiint(adj) == __double2hiint(adj_cur);
    if (!same) {                                          // wave-uniform
#pragma unroll
        for (int m = 0; m < M; ++m) { L[m] -= adj_cur * fast_log(Pc[m]); Pc[m] = 1.0; }
        adj_cur = adj;
    } else {
        ccs_flush<M>(L, Pc, adj_cur, 1e100);
    }
}

// PD: entries whose records are in flight while one is computed (the indices of one more are on their way).
// NT = 1024 (IR): the RECIPROCAL RADII of all N nodes lie in LDS (radii belong to nodes, not to slices: 8 N bytes,
// one 16-wavefront workgroup per CU shares them), and a term gathers its partner's position alone - one 16-byte
// request per term at d = 2, straight from X (eight positions per 128-byte line where a record has four) - instead
// of a record's two requests.  What this is for (profiles/r06_cc_pass_notes.md): with the vector arithmetic cut to
// ~30 / ~13 us the pass stayed at 54 / 50 us whatever the wavefronts per CU (8 .. 20) and the pipeline's depth
// (1 .. 3) - the vector L1 takes about one lane request per clock, and a term was two of them.
template <int D, int M, bool TWO, int PD, int NT>
__global__ __launch_bounds__(NT) void k_loglik_casecontrol_stream(
    ChainView c, LoglikCand cand, const double *__restrict__ XR, const int32_t *__restrict__ terms, int tw,
    const int32_t *__restrict__ order, const int32_t *__restrict__ order_count, int emax,
    double *__restrict__ partials, int rslot) {
    static_assert(!TWO || M == 2, "two radii: the radii step's two candidates");
    static_assert(CC_ENT_TERMS == 64 * LLCS_NS, "an entry is the trips one step requests ahead");
    constexpr bool IR = NT == 1024;
    static_assert(NT == LLCS_THREADS || (IR && !TWO), "reciprocal radii in LDS: one radius per node");
    constexpr int NWV = NT / 64, NS = LLCS_NS, NB = PD + 1;
    constexpr int RW = llcc_record_width(D);
    extern __shared__ __attribute__((aligned(16))) double sDyn[];
    double *sTab = sDyn;                                   // [EXPTAB_N]
    double *sRed = sDyn + EXPTAB_N;                        // [NWV * M] (M <= 4)
    double *sInv = sDyn + EXPTAB_N + NWV * 4;              // [N] (IR)
    const int tid = threadIdx.x, lane = tid & 63;
    exp_table_fill(sTab, tid);                             // (the first 256 threads) barrier below
    const int N = c.N, sl = (int)blockIdx.y;
    const int G = (int)gridDim.x * NWV;
    const int g = __builtin_amdgcn_readfirstlane((int)blockIdx.x * NWV + (tid >> 6));
    const int E = order_count[sl];
    int E0, nseq;                                          // this wavefront's entries
    ccs_share(g, G, E, E0, nseq);
    const int32_t *trows = terms + (size_t)sl * N * tw;
    const int32_t *ord = order + (size_t)sl * N * emax + E0;
    const char *Rt = (const char *)(XR + (size_t)sl * N * RW);
    const char *Xt = (const char *)(c.X + (size_t)sl * N * D);
    const int squared = c.squared;
    double bin[M], bout[M];
#pragma unroll
    for (int m = 0; m < M; ++m) { bin[m] = cand.intercepts[2 * m]; bout[m] = cand.intercepts[2 * m + 1]; }
    double L[M], Pe[M], Pc[M];
#pragma unroll
    for (int m = 0; m < M; ++m) { L[m] = 0.0; Pe[m] = 1.0; Pc[m] = 1.0; }
    double adj_cur = 1.0;

    // Every per-entry request is a VECTOR load, also those of one address (the header: lane l takes slot l; the own
    // record: lane l takes double l) - in order with the gathers behind one counter, one register each; as scalar
    // loads they would share the counter of the table's LDS reads, which can only be waited to zero.
    struct RowA { int hv, e[NS]; int ent; };               // header slots by lane, the entry's indices; ent: scalar
    struct RowB { int od, nv, who; double adj; double ov; double xe[NS][D], re0[NS], re1[NS]; int e[NS]; };
    int ordv = 0;                                          // the block's entries, one per lane
    auto issue_a = [&](RowA &a, int kb) {                  // kb: the entry's lane in the block; < 0: none
        a.ent = kb >= 0 ? __builtin_amdgcn_readlane(ordv, kb & 63) : -1;
        const int r = max(a.ent, 0) & 0xFFFFFF, seg = max(a.ent, 0) >> 24;
        const int32_t *row = trows + (size_t)r * tw;
        a.hv = row[min(lane, CP_HDR - 1)];
#pragma unroll
        for (int s = 0; s < NS; ++s) a.e[s] = row[CP_HDR + min(CC_ENT_TERMS * seg + 64 * s + lane, tw - CP_HDR - 1)];
    };
    auto issue_b = [&](RowB &bq, const RowA &a) {
        const bool have = a.ent >= 0;
        const int seg = max(a.ent, 0) >> 24;
        const int outdeg = __builtin_amdgcn_readlane(a.hv, 1), nt = outdeg + __builtin_amdgcn_readlane(a.hv, 3);
        bq.od = outdeg - CC_ENT_TERMS * seg;               // edge lanes of the entry's first trip (<= 0: none)
        bq.nv = have ? min(CC_ENT_TERMS, nt - CC_ENT_TERMS * seg) : 0;     // its out-edges + out-controls
        bq.adj = __hiloint2double(__builtin_amdgcn_readlane(a.hv, 7), __builtin_amdgcn_readlane(a.hv, 6));
        bq.who = __builtin_amdgcn_readlane(a.hv, 8);
        if (IR) bq.ov = ((const double *)(Xt + (size_t)bq.who * (D * sizeof(double))))[min(lane, D - 1)];
        else bq.ov = ((const double *)(Rt + (size_t)bq.who * (RW * sizeof(double))))[min(lane, RW - 1)];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int es = 64 * s + lane < bq.nv ? a.e[s] : 0;
            if (IR) {
                const double *xq = (const double *)(Xt + __umul24((uint32_t)es, (uint32_t)(D * sizeof(double))));
#pragma unroll
                for (int d = 0; d < D; ++d) bq.xe[s][d] = xq[d];
                bq.e[s] = es;
            } else {
                const double *rec = (const double *)(Rt + __umul24((uint32_t)es, (uint32_t)(RW * sizeof(double))));
#pragma unroll
                for (int d = 0; d < D; ++d) bq.xe[s][d] = rec[d];
                bq.re0[s] = rec[D + (M == 1 ? rslot : 0)];
                bq.re1[s] = TWO ? rec[D + 1] : 0.0;
            }
        }
    };
    auto rdl_d = [&](double v, int l) -> double {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                                __builtin_amdgcn_readlane(__double2loint(v), l));
    };
    auto compute = [&](const RowB &bq) {
        const int nv = bq.nv, od = bq.od;
        if (nv <= 0) return;                              // wave-uniform
        double xi[D];
#pragma unroll
        for (int d = 0; d < D; ++d) xi[d] = rdl_d(bq.ov, d);
        const double iri0 = IR ? uniform_d(sInv[bq.who]) : rdl_d(bq.ov, D + (M == 1 ? rslot : 0));
        const double iri1 = TWO ? rdl_d(bq.ov, D + 1) : iri0;
        const double adj = bq.adj;
        ccs_flush<M>(L, Pe, 1.0, 1e100);                  // (then two factors of at most 1e56.5 are safe)
        ccs_row_weight<M>(L, Pc, adj_cur, adj);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (64 * s >= nv) continue;                   // wave-uniform
            const double rq0 = IR ? sInv[bq.e[s]] : bq.re0[s];
            ccs_trip<D, M, TWO>(bq.xe[s], rq0, bq.re1[s], xi, iri0, iri1, bin, bout, od - 64 * s,
                                nv - 64 * s, lane, squared, adj, sTab, L, Pe, Pc);
        }
    };

    ordv = lane < nseq ? ord[lane] : -1;                   // the first block's entries
    RowA A[NB];                                            // entry x: A[x % NB], requested NB steps before its records
    RowB B[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) issue_a(A[u], u < nseq ? u : -1);
    if (IR) {      // the candidate's radii (one array for M = 1 and M = 4: cand.radii[0]), reciprocal as the records' (k_pack_xr)
        const double *rr = cand.radii[0];
        for (int i = tid; i < N; i += NT) sInv[i] = 1.0 / rr[i];
    }
    __syncthreads();                                       // sTab, sInv visible (the first entry's header is on its way)
    for (int k0 = 0; k0 < nseq; k0 += 64) {
        const int nb = min(64, nseq - k0);
        if (k0 > 0) {
            ordv = k0 + lane < nseq ? ord[k0 + lane] : -1;
#pragma unroll
            for (int u = 0; u < NB; ++u) issue_a(A[u], u < nb ? u : -1);
        }
#pragma unroll
        for (int u = 0; u < PD; ++u) {                     // (a block's start: PD entries' records, one after the other)
            issue_b(B[u], A[u]);
            issue_a(A[u], u + NB < nb ? u + NB : -1);
        }
        for (int j = 0; j < nb; j += NB) {
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                if (j + u < nb) {
                    const int w = (u + PD) % NB;           // entry j + u + PD: its records, then the indices of the one NB on
                    issue_b(B[w], A[w]);
                    issue_a(A[w], j + u + PD + NB < nb ? j + u + PD + NB : -1);
                    __builtin_amdgcn_sched_barrier(0);
                    compute(B[u]);
                }
            }
        }
    }
    const int wave = tid >> 6;
#pragma unroll
    for (int m = 0; m < M; ++m) {
        L[m] -= adj_cur * fast_log(Pc[m]) + fast_log(Pe[m]);
        const double v = wave_sum_all(L[m]);
        if (lane == 0) sRed[wave * M + m] = v;
    }
    __syncthreads();
    if (tid < M) {
        double s = 0.0;
        for (int w = 0; w < NWV; ++w) s += sRed[w * M + tid];
        partials[((size_t)sl * gridDim.x + blockIdx.x) * M + tid] = s;
    }
}

}  // namespace dlsm
