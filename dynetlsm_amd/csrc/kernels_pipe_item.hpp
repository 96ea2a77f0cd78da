// The pipelined sweep's evaluator of the directed model, and of the undirected model where kernels_pipe_lds.hpp's
// evaluators do not apply (PIPE_UNDIRECTED_LONG: rows that do not fit the LDS, items that need a second round):
// one wavefront per (node, part of the neighbours) with the part's first trips prefetched into registers, the rest
// software-pipelined, and the lane's share of the batch's H entries at the item's tail.  (Included by
// kernels_spec_pipe.hpp inside namespace dlsm, behind the item's wavefront reductions.)
#pragma once

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
        // (the table exponential, as the LDS evaluators' item - kernels_pipe_lds.hpp: the polynomial form's thirteen
        // constants were hoisted into registers in front of the fallback's trip loop and cost it a spilled one)
        if (__builtin_amdgcn_ballot_w64(tiny)) { if (tiny) h *= tab_exp11_clamped((b0 - b1) - (a0 - a1), etab); }
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


