// The pipelined sweep's DENSE case-control form (k_pipe_step<D, DLSM_DIRECTED_CASE_CONTROL>: 512 <= N < 2048, below
// the sparse sweep of kernels_ccpipe.hpp): four wavefronts per node accumulate its gathered terms and the corrections
// of the window's nodes into an LDS column, and the [m][k] resolver that reads them.  (Included by
// kernels_spec_pipe.hpp inside namespace dlsm.)
#pragma once

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

