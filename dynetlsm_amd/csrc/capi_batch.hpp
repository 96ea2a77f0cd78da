// dlsm_batch_*: several chains of one network driven through shared launches (kernels_batch.hpp).
// Included at the end of capi.hip (unity build): uses its helpers.
#pragma once

struct dlsm_batch {
    std::vector<dlsm_chain *> ch;
    hipStream_t stream = nullptr;           // the batch's own stream: every member chain's calls order on it
    long merged_iterations = 0, single_iterations = 0;
    std::string err;
};

static void batch_forget(dlsm_chain *h) {
    dlsm_batch *b = (dlsm_batch *)h->batch;
    if (!b) return;
    for (auto &p : b->ch) if (p == h) p = nullptr;
}

namespace {

__global__ void k_words_differ(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, long n,
                               int *__restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && a[i] != b[i]) atomicOr(out, 1);
}

// one merged iteration of the undirected LSM loop for every chain of the batch: the fused path of
// enqueue_lsm_iteration (sweep with riding centring sums, likelihood pass on the swept positions,
// centring + accept / reject + trace row + next proposal pass), launch by launch
template <int DD>
int batch_enqueue_iteration(dlsm_batch *b, int it, int procrustes_ref) {
    const int nc = (int)b->ch.size();
    dlsm_chain *h0 = b->ch[0];
    const int T = h0->T, N = h0->N;
    const size_t row = (size_t)T * N * DD;
    const IterRef ir{(uint32_t)it, nullptr};
    PipeBatchArgs pa;
    RideBatchArgs ra;
    FinBatchArgs fa;
    memset(&pa, 0, sizeof(pa)); memset(&ra, 0, sizeof(ra)); memset(&fa, 0, sizeof(fa));
    pa.nc = ra.nc = fa.nc = nc;
    const int nbat = (N + PP_B - 1) / PP_B;
    const long rows = (long)T * N;
    const int nwg = (int)std::min<long>(PS_BLOCKS - T, (rows + PP_THREADS - 1) / PP_THREADS);
    for (int c = 0; c < nc; ++c) {
        dlsm_chain *h = b->ch[c];
        int rc = ensure_partials(h, (size_t)ll_blocks(h) * 4 + (size_t)PS_BLOCKS * 26); if (rc) { b->err = h->err; return rc; }
        const int nip = h->lsm_cfg.n_iter_procrustes;
        const int pref = it > nip ? procrustes_ref : -1;
        const double *xref = pref >= 0 ? h->trace_X + row * pref : nullptr;
        const double *ride_xref = (xref && (nip < 0 || it > nip)) ? xref : nullptr;
        h->loop_draws_intercept = true;
        PipeBuf pb;
        rc = launch_sweep_pipe<DD>(h, ir, false, 1, false, &pb);
        h->loop_draws_intercept = false;
        if (rc) { b->err = h->err; return rc; }
        ChainView v = h->view();
        v.ybits = h0->ybits; v.ycm = h0->ycm;        // one copy of the network serves the whole batch
        pa.c[c] = v; pa.pb[c] = pb;
        ra.c[c] = v; ra.pb[c] = pb;
        ra.pr[c] = PipePostRide{ride_xref, ir, h->partials + (size_t)ll_blocks(h) * 4, nwg, (nbat - 1) * PP_B,
                                T > 1 ? 1 : 0};
        fa.c[c] = v;
        FinBatchChain &f = fa.f[c];
        f.partials = h->partials; f.nrec = ll_blocks(h); f.lsm = h->lsm; f.intercept = h->intercept;
        f.trace_ic = h->trace_ic; f.trace_logp = h->trace_logp;
        f.nb = ProposeBuf{pb.prop, pb.consts, pb.sync, pb.nsync, pb.queue0, pb.lsm_draw};
        f.pa = PostFusedArgs{xref ? 1 : 0, nip, h->partials + (size_t)ll_blocks(h) * 4, nwg + T, (nbat - 1) * PP_B,
                             T > 1 ? 1 : 0, xref, h->trace_X};
    }
    const size_t lds = (size_t)PP_B * PR_LD * sizeof(double);
    const bool lng = pa.pb[0].per > 64 * pipe_prefetch_trips(DD);
    {
        static bool armed = false;          // per instantiation
        if (!armed) {
            HIPCHK(h0, hipFuncSetAttribute((const void *)k_pipe_step_batch<DD, DLSM_UNDIRECTED>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIPCHK(h0, hipFuncSetAttribute((const void *)k_pipe_step_batch<DD, PIPE_UNDIRECTED_LONG>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIPCHK(h0, hipFuncSetAttribute((const void *)k_pipe_last_ride_batch<DD>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            armed = true;
        }
    }
    // launch l: even slices resolve batch l / evaluate l + 1, odd slices one launch behind
    const int last = T > 1 ? nbat : nbat - 1;
    const int grid = std::max(h0->n_cu, nc * T + 1);
    for (int l = -1; l < last; ++l) {
        if (lng) hipLaunchKernelGGL((k_pipe_step_batch<DD, PIPE_UNDIRECTED_LONG>), dim3(grid), dim3(PP_THREADS), lds,
                                    b->stream, pa, l);
        else hipLaunchKernelGGL((k_pipe_step_batch<DD, DLSM_UNDIRECTED>), dim3(grid), dim3(PP_THREADS), lds,
                                b->stream, pa, l);
    }
    hipLaunchKernelGGL((k_pipe_last_ride_batch<DD>), dim3(T + nwg, nc), dim3(PP_THREADS), lds, b->stream, ra, last);
    HIPCHK(h0, hipGetLastError());
    // the likelihood pass of every chain on the positions as the sweep left them (each fills the chip)
    for (int c = 0; c < nc; ++c) {
        dlsm_chain *h = b->ch[c];
        int nrec = 0;
        uint32_t *own = h->ybits;
        unsigned long long *own_cm = h->ycm;
        h->ybits = h0->ybits; h->ycm = h0->ycm;
        int rc = loglik_records(h, 2, h->lsm->cand, nullptr, nullptr, &nrec);
        h->ybits = own; h->ycm = own_cm;
        if (rc) { b->err = h->err; return rc; }
    }
    hipLaunchKernelGGL((k_lsm_finalize_apply_propose_batch<DD>), dim3(1 + propose_blocks(T, N), nc), dim3(256), 0,
                       b->stream, fa, ir);
    HIPCHK(h0, hipGetLastError());
    for (int c = 0; c < nc; ++c) {
        dlsm_chain *h = b->ch[c];
        h->next_prop = fa.f[c].nb; h->next_prop_ok = true; h->pipe_touched = true;
        h->prop_drawn_for = (long)it + 1;
    }
    return DLSM_OK;
}

}  // namespace

extern "C" {

const char *dlsm_batch_last_error(const dlsm_batch *b) { return b ? b->err.c_str() : g_err.c_str(); }

int dlsm_batch_create(dlsm_chain *const *chains, int n, dlsm_batch **out) {
    dlsm_chain *nullh = nullptr;
    if (!out || !chains) FAIL(nullh, DLSM_E_ARG, "null argument");
    *out = nullptr;
    if (n < 1 || n > BATCH_MAXC) FAIL(nullh, DLSM_E_LIMIT, "a batch holds 1..%d chains (got %d)", BATCH_MAXC, n);
    dlsm_chain *h0 = chains[0];
    for (int c = 0; c < n; ++c) {
        dlsm_chain *h = chains[c];
        if (!h) FAIL(nullh, DLSM_E_ARG, "chain %d is NULL", c);
        if (h->batch) FAIL(nullh, DLSM_E_ARG, "chain %d already belongs to a batch", c);
        for (int d = 0; d < c; ++d) if (chains[d] == h) FAIL(nullh, DLSM_E_ARG, "chain %d is listed twice", c);
        if (h->device != h0->device || h->T != h0->T || h->N != h0->N || h->D != h0->D || h->model != h0->model ||
            h->squared != h0->squared)
            FAIL(nullh, DLSM_E_ARG, "chain %d differs from chain 0 in device, shape or model", c);
        if (h->model != DLSM_UNDIRECTED) FAIL(nullh, DLSM_E_ARG, "the batch form covers the undirected model");
        if (!h->have_network) FAIL(nullh, DLSM_E_ARG, "chain %d has no network", c);
    }
    HIPCHK(nullh, hipSetDevice(h0->device));
    for (int c = 0; c < n; ++c) HIPCHK(nullh, hipStreamSynchronize(chains[c]->stream));
    // the chains of a batch share ONE network: checked word for word here, chain 0's copy is the
    // one the shared launches read
    const long nw = (long)h0->T * h0->N * h0->W;
    for (int c = 1; c < n; ++c) {
        HIPCHK(nullh, hipMemsetAsync(h0->dsmall, 0, sizeof(int), h0->stream));
        hipLaunchKernelGGL(k_words_differ, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, h0->stream, h0->ybits,
                           chains[c]->ybits, nw, (int *)h0->dsmall);
        int differ = 0;
        HIPCHK(nullh, hipMemcpyAsync(&differ, h0->dsmall, sizeof(int), hipMemcpyDeviceToHost, h0->stream));
        HIPCHK(nullh, hipStreamSynchronize(h0->stream));
        if (differ) FAIL(nullh, DLSM_E_DATA, "chain %d holds a different network than chain 0", c);
    }
    dlsm_batch *b = new dlsm_batch();
    b->ch.assign(chains, chains + n);
    // (a stream of the batch's own: a member chain destroyed before the batch takes its own stream with it)
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) {
        delete b;
        FAIL(nullh, DLSM_E_HIP, "hipStreamCreate failed");
    }
    for (int c = 0; c < n; ++c) {
        dlsm_chain *h = chains[c];
        drop_graph(h);
        h->own_stream = h->stream;          // every call on a member chain now orders on the batch's stream
        h->stream = b->stream;
        h->batch = b;
    }
    *out = b;
    return DLSM_OK;
}

void dlsm_batch_destroy(dlsm_batch *b) {
    if (!b) return;
    if (b->stream) { hipStreamSynchronize(b->stream); }
    for (dlsm_chain *h : b->ch) {
        if (!h) continue;                   // destroyed before the batch
        h->stream = h->own_stream; h->own_stream = nullptr; h->batch = nullptr;
        h->prop_drawn_for = -1;
    }
    if (b->stream) hipStreamDestroy(b->stream);
    delete b;
}

int dlsm_batch_synchronize(dlsm_batch *b) {
    dlsm_chain *nullh = nullptr;
    if (!b) FAIL(nullh, DLSM_E_ARG, "null batch");
    HIPCHK(nullh, hipStreamSynchronize(b->stream));
    return DLSM_OK;
}

// iterations first .. first + count - 1 of dlsm_lsm_run for every chain of the batch (each with
// its own configuration, trace and Philox chain id).  An iteration whose launches can be shared -
// every chain configured alike, the pipelined sweep, the previous iteration's last launch having
// drawn the proposals - runs merged; the others (the first and the last of a call, as in
// dlsm_lsm_run) run chain by chain on the batch's stream.  Asynchronous.
int dlsm_batch_lsm_run(dlsm_batch *b, int first, int count, int procrustes_ref) {
    dlsm_chain *nullh = nullptr;
    if (!b) FAIL(nullh, DLSM_E_ARG, "null batch");
    b->err.clear();
    for (dlsm_chain *h : b->ch) {
        if (!h) { b->err = "a chain of the batch was destroyed"; return DLSM_E_ARG; }
#define BNEED(cond, msg) do { if (!(cond)) { b->err = msg; return DLSM_E_ARG; } } while (0)
        BNEED(h->lsm_configured && h->trace_X, "configure every chain and allocate its trace first");
        BNEED(h->prior_kind == DLSM_PRIOR_RANDOM_WALK, "LSM uses the random-walk prior");
        BNEED(first >= 1 && count >= 0 && first + count <= h->trace_n, "iteration range out of a chain's trace");
        BNEED(procrustes_ref < h->trace_n, "procrustes_ref out of the trace");
        BNEED(h->have_X && h->have_samplers && h->have_prior, "a chain's state is incomplete");
#undef BNEED
    }
    if (count == 0) return DLSM_OK;
    dlsm_chain *h0 = b->ch[0];
    HIPCHK(nullh, hipSetDevice(h0->device));
    bool alike = resolve_sweep_algo(h0, h0->lsm_cfg.sweep_algo) == 4 && !h0->profiling &&
                 !(getenv("DLSM_BATCH_MERGE") && atoi(getenv("DLSM_BATCH_MERGE")) == 0) &&
                 h0->T + 4 <= PS_BLOCKS;
    for (dlsm_chain *h : b->ch)
        alike = alike && h->lsm_cfg.sweep_algo == h0->lsm_cfg.sweep_algo && !h->profiling &&
                h->lsm_cfg.n_iter_procrustes == h0->lsm_cfg.n_iter_procrustes && h->tune == h0->tune;
    for (dlsm_chain *h : b->ch) h->prop_drawn_for = -1;
    int rc = DLSM_OK;
    for (int it = first; it < first + count; ++it) {
        const bool last = it + 1 == first + count;
        bool merged = alike && !last;
        for (dlsm_chain *h : b->ch)
            merged = merged && h->prop_drawn_for == (long)it && h->next_prop_ok && h->next_prop.lsm_draw;
        if (merged) {
            DISPATCH_D(h0, h0->D, rc = batch_enqueue_iteration<DD>(b, it, procrustes_ref));
            if (rc) { if (b->err.empty()) b->err = h0->err; return rc; }
            ++b->merged_iterations;
            continue;
        }
        for (dlsm_chain *h : b->ch) {
            rc = enqueue_lsm_iteration(h, it, false, it > h->lsm_cfg.n_iter_procrustes ? procrustes_ref : -1, false,
                                       !last);
            if (rc) { b->err = h->err; return rc; }
        }
        ++b->single_iterations;
    }
    for (dlsm_chain *h : b->ch) h->prop_drawn_for = -1;
    return DLSM_OK;
}

// how many iterations of the calls so far ran merged / chain by chain (tests, bench)
int dlsm_batch_stats(dlsm_batch *b, int64_t *merged, int64_t *single) {
    dlsm_chain *nullh = nullptr;
    if (!b || !merged || !single) FAIL(nullh, DLSM_E_ARG, "null argument");
    *merged = b->merged_iterations; *single = b->single_iterations;
    return DLSM_OK;
}

}  // extern "C"
