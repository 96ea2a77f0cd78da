// C-ABI of the MI355X engine (include/dynetlsm_hip.h).  gfx950 only; there is
// deliberately no CPU path in this library.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "chain.hpp"
#include "device_common.hpp"
#include "kernels_labels.hpp"
#include "kernels_hdp.hpp"
#include "cc_rows.hpp"
#include "kernels_loglik.hpp"
#include "kernels_loglik_ccstream.hpp"
#include "kernels_hdploop.hpp"
#include "kernels_sweep.hpp"
#include "kernels_dirloop.hpp"
#include "kernels_spec.hpp"
#include "kernels_spec_sweep.hpp"
#include "kernels_spec_pipe.hpp"
#include "kernels_tail_propose.hpp"
#include "kernels_ccpipe.hpp"
#include "kernels_init.hpp"
#include "kernels_post.hpp"
#include "kernels_forecast.hpp"
#include "host_draws.hpp"

using namespace dlsm;

static thread_local std::string g_err;
static int32_t *g_fix_err_host = nullptr;       // CC_ERR_FIXPOINT's word (kernels_ccpipe.hpp), one per process and device
static int g_fix_err_device = -1;
static std::atomic<int> g_live_chains{0};       // handles alive in this process (capi_hdp.hpp, hdp_fork_arm)

#define FAIL(h, code, ...)                                      \
    do {                                                        \
        char _b[512];                                           \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                  \
        if (h) (h)->err = _b; else g_err = _b;                  \
        return (code);                                          \
    } while (0)

#define HIPCHK(h, call)                                                          \
    do {                                                                         \
        hipError_t _e = (call);                                                  \
        if (_e != hipSuccess)                                                    \
            FAIL(h, DLSM_E_HIP, "%s failed: %s (%s:%d)", #call,                  \
                 hipGetErrorString(_e), __FILE__, __LINE__);                     \
    } while (0)

#define NEED(h, cond, ...)                               \
    do {                                                 \
        if (!(cond)) FAIL(h, DLSM_E_ARG, __VA_ARGS__);   \
    } while (0)

// dispatch on the compile-time latent dimension
// (DLSM_DEV_ONLY_D2: development builds that instantiate n_features = 2 only - an eighth of the compile time
// for A/B libraries under tmp_timing/; never the shipped library)
#ifdef DLSM_DEV_ONLY_D2
#define DISPATCH_D(h, D_, ...)                                              \
    switch (D_) {                                                           \
        case 2: { constexpr int DD = 2; __VA_ARGS__; } break;               \
        default: FAIL(h, DLSM_E_LIMIT, "n_features=%d: this development build holds n_features = 2 only", D_); \
    }
#else
#define DISPATCH_D(h, D_, ...)                                              \
    switch (D_) {                                                           \
        case 1: { constexpr int DD = 1; __VA_ARGS__; } break;               \
        case 2: { constexpr int DD = 2; __VA_ARGS__; } break;               \
        case 3: { constexpr int DD = 3; __VA_ARGS__; } break;               \
        case 4: { constexpr int DD = 4; __VA_ARGS__; } break;               \
        case 5: { constexpr int DD = 5; __VA_ARGS__; } break;               \
        case 6: { constexpr int DD = 6; __VA_ARGS__; } break;               \
        case 7: { constexpr int DD = 7; __VA_ARGS__; } break;               \
        case 8: { constexpr int DD = 8; __VA_ARGS__; } break;               \
        default: FAIL(h, DLSM_E_LIMIT, "n_features=%d unsupported (1..8)", D_); \
    }
#endif

// A captured iteration freezes pointers and scalars of the chain: anything that changes
// them drops the graph (it is rebuilt by the next dlsm_lsm_run).
static int check_pipe_err(dlsm_chain *h);
static int check_sweep_algo(dlsm_chain *h, int algo);
static int build_colmajor(dlsm_chain *h);

static void drop_graph(dlsm_chain *h) {
    if (h->graph_exec) hipGraphExecDestroy(h->graph_exec);
    if (h->graph) hipGraphDestroy(h->graph);
    h->graph_exec = nullptr; h->graph = nullptr; h->graph_ref = -2; h->graph_algo = -1;
}

namespace {

struct ProfScope {
    dlsm_chain *h; int k; hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(dlsm_chain *h_, int k_) : h(h_), k(k_) {
        if (h->profiling) {
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, h->stream);
        }
    }
    ~ProfScope() {
        if (h->profiling) {
            hipEventRecord(e1, h->stream);
            h->prof[k].pending.emplace_back(e0, e1);
        }
    }
};

template <typename T>
int dev_alloc(dlsm_chain *h, T **p, size_t n) {
    HIPCHK(h, hipMalloc((void **)p, std::max<size_t>(n, 1) * sizeof(T)));
    return DLSM_OK;
}

// Host <-> device copies of the C-ABI, synchronous at return.  The caller's arrays are
// pageable memory, for which the runtime's own staging is slow at these sizes
// (profiles/micro/copy_paths.cpp: 32 KB up 39 us against 10 us from pinned memory, a
// few bytes down 25 us against 12 us), so small copies go through the handle's pinned buffer.
constexpr size_t STAGE_BYTES = 256 * 1024;
constexpr size_t STAGE_D2H_MAX = 64 * 1024;    // beyond it the extra memcpy costs what it saves

template <typename T>
int h2d(dlsm_chain *h, T *dst, const T *src, size_t n) {
    const size_t bytes = n * sizeof(T);
    const void *from = src;
    if (h->stage && bytes <= STAGE_BYTES) { memcpy(h->stage, src, bytes); from = h->stage; }
    HIPCHK(h, hipMemcpyAsync(dst, from, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DLSM_OK;
}

template <typename T>
int d2h(dlsm_chain *h, T *dst, const T *src, size_t n) {
    const size_t bytes = n * sizeof(T);
    const bool staged = h->stage && bytes <= STAGE_D2H_MAX;
    HIPCHK(h, hipMemcpyAsync(staged ? h->stage : (void *)dst, src, bytes, hipMemcpyDeviceToHost,
                             h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (staged) memcpy(dst, h->stage, bytes);
    return DLSM_OK;
}

// Several small uploads ahead of a launch: each takes the next slice of the pinned buffer and is
// only enqueued (the synchronous d2h that ends the call, or the caller, waits for them).
// `*off` = bytes of the buffer in use; a copy that does not fit goes the runtime's way.
template <typename T>
int h2d_enqueue(dlsm_chain *h, T *dst, const T *src, size_t n, size_t *off) {
    const size_t bytes = n * sizeof(T);
    const void *from = src;
    if (h->stage && *off + bytes <= STAGE_BYTES) {
        from = (char *)h->stage + *off;
        memcpy((void *)from, src, bytes);
        *off += (bytes + 255) & ~(size_t)255;
    }
    HIPCHK(h, hipMemcpyAsync(dst, from, bytes, hipMemcpyHostToDevice, h->stream));
    return DLSM_OK;
}

// int64 host array -> int32 device array; every value must lie in [lo, hi) (node indices
// index device arrays: a bad one would be an out-of-bounds read in the kernels)
int upload_i64_as_i32(dlsm_chain *h, int32_t **dst, const int64_t *src, size_t n,
                      int64_t lo, int64_t hi, const char *what) {
    std::vector<int32_t> tmp(n);
    for (size_t i = 0; i < n; ++i) {
        if (src[i] < lo || src[i] >= hi)
            FAIL(h, DLSM_E_DATA, "%s[%zu] = %lld outside [%lld, %lld)", what, i,
                 (long long)src[i], (long long)lo, (long long)hi);
        tmp[i] = (int32_t)src[i];
    }
    if (*dst) { hipFree(*dst); *dst = nullptr; }
    int rc = dev_alloc(h, dst, n);
    if (rc) return rc;
    return h2d(h, *dst, tmp.data(), n);
}

int ensure_partials(dlsm_chain *h, size_t n) {
    if (h->partials_cap >= n) return DLSM_OK;
    if (h->partials) hipFree(h->partials);
    h->partials = nullptr; h->partials_cap = 0;
    int rc = dev_alloc(h, &h->partials, n);
    if (rc) return rc;
    h->partials_cap = n;
    return DLSM_OK;
}

// counts and transition matrices of the label update, sized for the current n_components
int ensure_label_bufs(dlsm_chain *h) {
    const size_t nn = (size_t)h->T * h->K * h->K, nnk = (size_t)h->T * h->K;
    if (h->lab_cap >= nn) return DLSM_OK;   // n_components may grow on a live handle
    void *old[] = {h->lab_n, h->lab_nk, h->lab_w};
    for (void *p : old) if (p) hipFree(p);
    h->lab_n = h->lab_nk = nullptr; h->lab_w = nullptr; h->lab_cap = 0;
    int rc = dev_alloc(h, &h->lab_n, nn); if (rc) return rc;
    rc = dev_alloc(h, &h->lab_nk, nnk); if (rc) return rc;
    rc = dev_alloc(h, &h->lab_w, nn); if (rc) return rc;
    h->lab_cap = nn;
    return DLSM_OK;
}

int ll_blocks(const dlsm_chain *h) {
    if (h->model == DLSM_DIRECTED_CASE_CONTROL)     // (k_loglik_casecontrol_rows: whole workgroups per slice)
        return h->T * ((h->N + LLCC_NODES - 1) / LLCC_NODES);
    int nt = (h->N + LL_TILE - 1) / LL_TILE;
    // (undirected: two row halves per tile above the diagonal, one workgroup per diagonal tile: kernels_loglik.hpp)
    return h->model == DLSM_UNDIRECTED ? h->T * llu_blocks_per_slice(nt) : h->T * (nt * (nt + 1) / 2);
}

int check_ready_loglik(dlsm_chain *h) {
    NEED(h, h->have_X, "latent positions not set");
    if (h->model == DLSM_DIRECTED_CASE_CONTROL) {
        NEED(h, h->have_edges && h->have_controls, "edge lists / controls not set");
    } else {
        NEED(h, h->have_network, "network not uploaded");
    }
    if (h->model != DLSM_UNDIRECTED) NEED(h, h->have_radii, "radii not set");
    return DLSM_OK;
}

// Case-control model: the valid controls per node and direction (k_count_controls) and the nodes' term rows
// (cc_rows.hpp), rebuilt on the chain's stream when the edge tables or the controls have changed (upload /
// set / resample clear the two flags).  alloc_only: buffers only (configure calls).
static int ensure_cc_rows(dlsm_chain *h, bool alloc_only = false) {
    const size_t TN = (size_t)h->T * h->N;
    if (h->nctrl_cap < TN * 2) {
        if (h->nctrl) hipFree(h->nctrl);
        h->nctrl = nullptr; h->nctrl_cap = 0;
        HIPCHK(h, hipMalloc((void **)&h->nctrl, TN * 2 * sizeof(int32_t)));
        h->nctrl_cap = TN * 2;
        h->nctrl_valid = false;
    }
    const int cap = std::max(1, h->Din + h->Dout + 2 * h->C);
    const int tw = cp_terms_width(cap);
    const size_t n_terms = TN * tw;
    if (h->cc_terms_cap < n_terms || h->cc_tw != tw) {
        if (h->cc_terms_cap < n_terms) {
            if (h->cc_terms) hipFree(h->cc_terms);
            h->cc_terms = nullptr; h->cc_terms_cap = 0;
            HIPCHK(h, hipMalloc((void **)&h->cc_terms, n_terms * sizeof(int32_t)));
            h->cc_terms_cap = n_terms;
        }
        h->cc_tw = tw; h->cc_terms_valid = false;
    }
    {   // the likelihood pass's walking order: entries of at most CC_ENT_TERMS out-terms (k_cc_order) + a count per slice
        const int emax = cc_order_entries_max(h->Dout, h->C);
        if (h->cc_order_cap < TN * emax + (size_t)h->T || h->cc_emax != emax) {
            if (h->cc_order) hipFree(h->cc_order);
            h->cc_order = nullptr; h->cc_order_cap = 0;
            HIPCHK(h, hipMalloc((void **)&h->cc_order, (TN * emax + (size_t)h->T) * sizeof(int32_t)));
            h->cc_order_cap = TN * emax + (size_t)h->T; h->cc_emax = emax; h->cc_terms_valid = false;
            h->cc_order_cnt = h->cc_order + TN * emax;
        }
    }
    if (h->cc_pos_cap < TN) {
        if (h->cc_pos) hipFree(h->cc_pos);
        h->cc_pos = nullptr; h->cc_pos_cap = 0;
        HIPCHK(h, hipMalloc((void **)&h->cc_pos, TN * sizeof(int32_t)));
        h->cc_pos_cap = TN; h->cc_terms_valid = false;
    }
    if (alloc_only) return DLSM_OK;
    if (!h->nctrl_valid) {      // the control lists change only in set / resample
        hipLaunchKernelGGL(k_count_controls, dim3((unsigned)((TN + 255) / 256)), dim3(256),
                           0, h->stream, h->ctrl_in, h->ctrl_out, (long)TN, h->C, h->nctrl);
        h->nctrl_valid = true; h->cc_terms_valid = false;
    }
    if (!h->cc_terms_valid) {
        hipLaunchKernelGGL(k_cc_pos, dim3((unsigned)((h->N + CC_SORT_B - 1) / CC_SORT_B), (unsigned)h->T),
                           dim3(CC_SORT_B), 0, h->stream, h->view(), h->nctrl, h->cc_pos);
        hipLaunchKernelGGL(k_cc_rows, dim3((unsigned)((TN + 3) / 4)), dim3(256), 0, h->stream, h->view(),
                           h->nctrl, h->cc_pos, h->cc_terms, tw);
        hipLaunchKernelGGL(k_cc_order, dim3((unsigned)((h->N + 255) / 256), (unsigned)h->T), dim3(256), 0, h->stream,
                           h->view(), h->nctrl, h->cc_pos, h->cc_emax, h->cc_order, h->cc_order_cnt);
        h->cc_terms_valid = true;
    }
    return DLSM_OK;
}

// the case-control pass reads packed gather records (k_pack_xr) - any out-degree, any number of controls
static bool cc_prefetch_form(const dlsm_chain *h) {
    return h->model == DLSM_DIRECTED_CASE_CONTROL;
}
// its gather records (positions and both candidates' radii, one record per node)
template <int DD>
int ensure_xr(dlsm_chain *h) {
    const size_t need = (size_t)h->T * h->N * llcc_record_width(DD);
    if (h->xr_cap < need) {
        if (h->xr) hipFree(h->xr);
        h->xr = nullptr; h->xr_cap = 0;
        HIPCHK(h, hipMalloc((void **)&h->xr, need * sizeof(double)));
        h->xr_cap = need;
    }
    return DLSM_OK;
}

// The streaming case-control pass: as many workgroups per slice as stay resident together (occupancy query, once
// per instantiation), trimmed so that every wavefront walks the same number of row pairs.
template <int DD, int M, bool TWO, int PD, int NT>
int launch_ccs(dlsm_chain *h, const ChainView &v, const LoglikCand &cand, int rslot, int *nrec_out) {
    constexpr bool IR = NT == 1024;         // reciprocal radii in LDS: one workgroup per CU
    constexpr int NWV = NT / 64;
    const size_t lds = (size_t)(EXPTAB_N + NWV * 4 + (IR ? h->N : 0)) * sizeof(double);
    static int bpc = 0;                     // workgroups per CU
    if (bpc == 0) {
        int nblk = 0;
        if (IR) HIPCHK(h, hipFuncSetAttribute((const void *)k_loglik_casecontrol_stream<DD, M, TWO, PD, NT>,
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, k_loglik_casecontrol_stream<DD, M, TWO, PD, NT>,
                                                                    NT, lds);
        if (e != hipSuccess || nblk < 1) { (void)hipGetLastError(); nblk = IR ? 1 : 2; }
        const char *eo = getenv("DLSM_CC_PASS_WG_PER_CU");     // (experiments)
        if (eo && atoi(eo) > 0) nblk = atoi(eo);
        bpc = nblk;
    }
    const int N = h->N, T = h->T;
    // (a slice's entries - k_cc_order: N of them and a few per cent - are dealt to the wavefronts in equal
    // contiguous shares; the grid is trimmed so that the shares are whole numbers of entries, near enough)
    const int wps = ccs_workgroups_per_slice(N, T, h->n_cu, bpc, NWV, (N + LLCC_NODES - 1) / LLCC_NODES);   // (cap: ll_blocks)
    hipLaunchKernelGGL((k_loglik_casecontrol_stream<DD, M, TWO, PD, NT>), dim3((unsigned)wps, (unsigned)T),
                       dim3(NT), lds, h->stream, v, cand, h->xr, h->cc_terms, h->cc_tw, h->cc_order,
                       h->cc_order_cnt, h->cc_emax, h->partials, rslot);
    *nrec_out = wps * T;
    return DLSM_OK;
}
template <int DD>
int launch_loglik_ccstream(dlsm_chain *h, int M, bool two, const ChainView &v, const LoglikCand &cand, int rslot,
                           int *nrec_out) {
    // one radius per node (every pass but the two-radii form) and the reciprocals fit a CU's LDS: positions alone
    // are gathered (DLSM_CC_PASS=records keeps the gathered records)
    const bool ir = !two && (size_t)h->N * sizeof(double) <= 150 * 1024 &&
                    !(getenv("DLSM_CC_PASS") && strcmp(getenv("DLSM_CC_PASS"), "records") == 0);
    // (d <= 3: beyond it a position is as many requests as a record, and the 16-wavefront workgroup's 128 registers
    // no longer hold two entries of d doubles per term)
    if constexpr (DD <= 3) {
        if (ir) {
            if (M == 1) return launch_ccs<DD, 1, false, 2, 1024>(h, v, cand, rslot, nrec_out);
            if (M == 4) return launch_ccs<DD, 4, false, 1, 1024>(h, v, cand, 0, nrec_out);
            return launch_ccs<DD, 2, false, 1, 1024>(h, v, cand, 0, nrec_out);
        }
    }
    if (M == 1) return launch_ccs<DD, 1, false, 1, LLCS_THREADS>(h, v, cand, rslot, nrec_out);
    if (M == 4) return launch_ccs<DD, 4, false, 1, LLCS_THREADS>(h, v, cand, 0, nrec_out);
    if (two) return launch_ccs<DD, 2, true, 1, LLCS_THREADS>(h, v, cand, 0, nrec_out);
    return launch_ccs<DD, 2, false, 1, LLCS_THREADS>(h, v, cand, 0, nrec_out);
}

// Enqueue the log-likelihood record kernel for M candidates whose intercepts
// are at device address `d_ic` (and radii r0 / r1); returns the record count.
// reuse_pack: the records are those this pass needs already (the caller knows); rslot: which of
// a record's two radii a single candidate reads
template <int DD>
int launch_loglik_records(dlsm_chain *h, int M, const double *d_ic,
                          const double *r0, const double *r1, int *nrec_out,
                          bool reuse_pack = false, int rslot = 0) {
    const int nb = ll_blocks(h);
    int rc = ensure_partials(h, (size_t)nb * 4);
    if (rc) return rc;
    ChainView v = h->view();
    LoglikCand cand{d_ic, {r0, r1}};
    ProfScope ps(h, DLSM_K_LOGLIK);
    if (h->model == DLSM_UNDIRECTED) {
        const int prio = h->ll_beside_chain ? 0 : 1;      // (issue priority by progress, unless the chain's launches run beside)
        if (M == 1) hipLaunchKernelGGL((k_loglik_undirected<DD, 1>), dim3(nb), dim3(LLU_THREADS), 0, h->stream, v, cand, h->partials, prio);
        else hipLaunchKernelGGL((k_loglik_undirected<DD, 2>), dim3(nb), dim3(LLU_THREADS), 0, h->stream, v, cand, h->partials, prio);
    } else if (h->model == DLSM_DIRECTED) {
        if (M == 1) hipLaunchKernelGGL((k_loglik_directed<DD, 1>), dim3(nb), dim3(LL_THREADS), 0, h->stream, v, cand, h->partials);
        else hipLaunchKernelGGL((k_loglik_directed<DD, 2>), dim3(nb), dim3(LL_THREADS), 0, h->stream, v, cand, h->partials);
    } else {
        // (the pass gathers records as 32-bit lane offsets from a slice's base: umul24 of the node)
        if (h->N >= (1 << 24) || (double)h->N * llcc_record_width(DD) * sizeof(double) >= 4294967296.0)
            FAIL(h, DLSM_E_LIMIT, "case-control log-likelihood pass: N=%d is beyond its 32-bit gather offsets", h->N);
        const bool pf = cc_prefetch_form(h);
        if (pf) {       // positions and both candidates' radii as one record per node
            const size_t nodes = (size_t)h->T * h->N;
            int rc2 = ensure_xr<DD>(h); if (rc2) return rc2;
            if (!reuse_pack)
                hipLaunchKernelGGL((k_pack_xr<DD>), dim3((unsigned)((nodes + 255) / 256)), dim3(256), 0,
                                   h->stream, h->X, r0, M > 1 ? r1 : r0, (long)nodes, h->N, h->xr);
        }
        // out-edges and out-controls as dense 64-term trips from the node's row (cc_rows.hpp)
        { int rc3 = ensure_cc_rows(h); if (rc3) return rc3; }
        // (round 6: a resident wave of streaming wavefronts - kernels_loglik_ccstream.hpp; DLSM_CC_PASS=rows keeps
        // the two-rows-per-wavefront form)
        // (k_cc_order ranks by out_deg * 65536 + n_out_controls in 32 bits: lists beyond that keep the rows form)
        const bool stream_form = !(getenv("DLSM_CC_PASS") && strcmp(getenv("DLSM_CC_PASS"), "rows") == 0) &&
                                 cc_order_key_holds(h->Dout, h->C);
        if (stream_form) {
            const bool two = M == 2 && r1 != r0;
            int rc4 = launch_loglik_ccstream<DD>(h, M, two, v, cand, rslot, nrec_out); if (rc4) return rc4;
            HIPCHK(h, hipGetLastError());
            return DLSM_OK;
        }
        if (M == 1) hipLaunchKernelGGL((k_loglik_casecontrol_rows<DD, 1>), dim3(nb / h->T, h->T), dim3(LLCR_THREADS), 0, h->stream, v, cand, h->xr, h->cc_terms, h->cc_tw, h->partials, rslot);
        else if (M == 4) hipLaunchKernelGGL((k_loglik_casecontrol_rows<DD, 4>), dim3(nb / h->T, h->T), dim3(LLCR_THREADS), 0, h->stream, v, cand, h->xr, h->cc_terms, h->cc_tw, h->partials, 0);
        else hipLaunchKernelGGL((k_loglik_casecontrol_rows<DD, 2>), dim3(nb / h->T, h->T), dim3(LLCR_THREADS), 0, h->stream, v, cand, h->xr, h->cc_terms, h->cc_tw, h->partials, 0);
    }
    HIPCHK(h, hipGetLastError());
    *nrec_out = nb;
    return DLSM_OK;
}

int loglik_records(dlsm_chain *h, int M, const double *d_ic, const double *r0,
                   const double *r1, int *nrec, bool reuse_pack = false, int rslot = 0) {
    DISPATCH_D(h, h->D, return launch_loglik_records<DD>(h, M, d_ic, r0, r1, nrec, reuse_pack, rslot));
    return DLSM_OK;
}

int drain_profile(dlsm_chain *h) {
    for (int k = 0; k < DLSM_K_COUNT; ++k) {
        for (auto &pr : h->prof[k].pending) {
            float ms = 0.f;
            hipEventSynchronize(pr.second);
            hipEventElapsedTime(&ms, pr.first, pr.second);
            h->prof[k].ms += ms;
            h->prof[k].launches += 1;
            hipEventDestroy(pr.first);
            hipEventDestroy(pr.second);
        }
        h->prof[k].pending.clear();
    }
    return DLSM_OK;
}

}  // namespace

extern "C" {

int dlsm_abi_version(void) { return DLSM_ABI_VERSION; }

int dlsm_device_count(int *count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; g_err = hipGetErrorString(e); return DLSM_E_NODEV; }
    *count = n;
    return DLSM_OK;
}

const char *dlsm_last_error(const dlsm_chain *h) {
    return h ? h->err.c_str() : g_err.c_str();
}

int dlsm_create(int device, int T, int N, int D, int model, uint64_t seed,
                uint32_t chain_id, dlsm_chain **out) {
    dlsm_chain *nullh = nullptr;
    if (!out) FAIL(nullh, DLSM_E_ARG, "out is NULL");
    *out = nullptr;
    if (T < 1 || N < 2 || D < 1 || D > DLSM_D_MAX)
        FAIL(nullh, DLSM_E_ARG, "bad shape T=%d N=%d D=%d (need T>=1, N>=2, 1<=D<=%d)", T, N, D, DLSM_D_MAX);
    if (T > 65535) FAIL(nullh, DLSM_E_LIMIT, "T=%d > 65535", T);
    if (model < 0 || model > 2) FAIL(nullh, DLSM_E_ARG, "bad model %d", model);
    if (chain_id >= (1u << 24)) FAIL(nullh, DLSM_E_ARG, "chain_id must be < 2^24");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        FAIL(nullh, DLSM_E_NODEV, "no HIP device available (the engine has no CPU path)");
    if (device < 0 || device >= ndev)
        FAIL(nullh, DLSM_E_NODEV, "device %d out of range (%d devices)", device, ndev);
    hipDeviceProp_t prop;
    HIPCHK(nullh, hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        FAIL(nullh, DLSM_E_NODEV, "device %d is %s; this engine is built for gfx950 only",
             device, prop.gcnArchName);
    HIPCHK(nullh, hipSetDevice(device));
    dlsm_chain *h = new dlsm_chain();
    g_live_chains.fetch_add(1);
    h->device = device; h->T = T; h->N = N; h->D = D; h->model = model;
    h->seed = seed; h->chain = chain_id;
    h->W = ((N + 31) / 32 + 3) / 4 * 4;
    h->n_cu = prop.multiProcessorCount;
    auto bail = [&](int rc) { g_err = h->err; dlsm_destroy(h); return rc; };
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess)
        { h->err = "hipStreamCreate failed"; return bail(DLSM_E_HIP); }
    if (hipHostMalloc(&h->stage, STAGE_BYTES, hipHostMallocDefault) != hipSuccess)
        h->stage = nullptr;                     // the copies then take the runtime's path
    // the sticky error word of the in-kernel waits (the HDP-LPCM loop's two queues, the case-control sweep's
    // helper workgroups): host memory the device can store to, read by the host without a copy
    if (model == DLSM_DIRECTED_CASE_CONTROL && (!g_fix_err_host || g_fix_err_device != device)) {
        // (the first case-control chain of the process on this device; a second device re-points the word)
        int32_t *hostw = g_fix_err_host, *devw = nullptr;
        if (!hostw && hipHostMalloc((void **)&hostw, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) hostw = nullptr;
        if (hostw) {
            if (!g_fix_err_host) memset(hostw, 0, 64);
            if (hipHostGetDevicePointer((void **)&devw, hostw, 0) == hipSuccess &&
                hipMemcpyToSymbol(HIP_SYMBOL(g_cc_fixpoint_err), &devw, sizeof(devw)) == hipSuccess) {
                g_fix_err_host = hostw; g_fix_err_device = device;
            }
        }
        (void)hipGetLastError();
    }
    if (hipHostMalloc((void **)&h->fork_err_host, 64, hipHostMallocMapped) == hipSuccess) {
        memset(h->fork_err_host, 0, 64);
        if (hipHostGetDevicePointer((void **)&h->fork_err_dev, h->fork_err_host, 0) != hipSuccess) h->fork_err_dev = nullptr;
    } else { h->fork_err_host = nullptr; (void)hipGetLastError(); }
    const size_t TN = (size_t)T * N;
    int rc = 0;
    rc |= dev_alloc(h, &h->X, TN * D);
    rc |= dev_alloc(h, &h->intercept, 2);
    rc |= dev_alloc(h, &h->radii, N);
    rc |= dev_alloc(h, &h->radii_alt, N);
    rc |= dev_alloc(h, &h->step, TN);
    rc |= dev_alloc(h, &h->nacc, TN);
    rc |= dev_alloc(h, &h->nsteps, TN);
    rc |= dev_alloc(h, &h->until, TN);
    rc |= dev_alloc(h, &h->dsmall, 128);
    rc |= dev_alloc(h, &h->xref, TN * D);
    rc |= dev_alloc(h, &h->lsm, 1);
    rc |= dev_alloc(h, &h->z, TN);
    rc |= dev_alloc(h, &h->hdp, 1);
    if (rc) return bail(DLSM_E_HIP);
    if (hipHostMalloc((void **)&h->hsmall, 64 * sizeof(double)) != hipSuccess)
        { h->err = "hipHostMalloc failed"; return bail(DLSM_E_HIP); }
    hipMemsetAsync(h->intercept, 0, 2 * sizeof(double), h->stream);
    hipMemsetAsync(h->lsm, 0, sizeof(LsmDeviceState), h->stream);
    hipMemsetAsync(h->hdp, 0, sizeof(HdpDeviceState), h->stream);
    hipEventCreate(&h->timer0);
    hipEventCreate(&h->timer1);
    hipStreamSynchronize(h->stream);
    *out = h;
    return DLSM_OK;
}

void dlsm_destroy(dlsm_chain *h) {
    if (!h) return;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    drain_profile(h);
    void *ptrs[] = {h->ycm, h->ybits, h->ytbits, h->in_edges, h->out_edges, h->degree,
                    h->ctrl_in, h->ctrl_out, h->X, h->intercept, h->radii,
                    h->radii_alt, h->step, h->nacc, h->nsteps, h->until, h->mu,
                    h->sigma, h->z, h->partials, h->dsmall, h->xref, h->lab_n,
                    h->lab_nk, h->lab_w, h->spec, h->nctrl, h->stamps, h->lsm, h->trace_X, h->trace_ic,
                    h->trace_logp, h->hops, h->hops_max, h->pipe, h->post_zt, h->post_cooc,
                    h->trace_radii, h->hdp, h->hdp_buf, h->htr_mu, h->htr_sigma, h->htr_beta,
                    h->htr_w, h->htr_lambda, h->htr_hyper, h->htr_z, h->xr, h->cc_terms, h->cc_pos, h->cc_order};
    for (void *p : ptrs) if (p) hipFree(p);
    if (h->hsmall) hipHostFree(h->hsmall);
    if (h->timer0) hipEventDestroy(h->timer0);
    if (h->timer1) hipEventDestroy(h->timer1);
    if (h->ev_a) hipEventDestroy(h->ev_a);
    if (h->ev_b) hipEventDestroy(h->ev_b);
    if (h->stream2) { hipStreamSynchronize(h->stream2); hipStreamDestroy(h->stream2); }
    if (h->fork_stream) { hipStreamSynchronize(h->fork_stream); hipStreamDestroy(h->fork_stream); }
    if (h->fork_ev) hipEventDestroy(h->fork_ev);
    if (h->fork_flags) hipFree(h->fork_flags);
    if (h->fork_err_host) hipHostFree(h->fork_err_host);
    g_live_chains.fetch_sub(1);
    if (h->graph_exec) hipGraphExecDestroy(h->graph_exec);
    if (h->graph) hipGraphDestroy(h->graph);
    if (h->stream) hipStreamDestroy(h->stream);
    if (h->stage) hipHostFree(h->stage);
    delete h;
}

int dlsm_synchronize(dlsm_chain *h) {
    NEED(h, h != nullptr, "null handle");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return check_pipe_err(h);
}

// ---------------------------------------------------------------- network
int dlsm_upload_network(dlsm_chain *h, const double *Y) {
    NEED(h, h && Y, "null argument");
    drop_graph(h);
    NEED(h, h->model != DLSM_DIRECTED_CASE_CONTROL,
         "case-control chains take edge lists (dlsm_upload_edges)");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t TN = (size_t)h->T * h->N;
    const size_t words = TN * h->W;
    if (!h->ybits) { int rc = dev_alloc(h, &h->ybits, words); if (rc) return rc; }
    if (h->model == DLSM_DIRECTED && !h->ytbits) {
        int rc = dev_alloc(h, &h->ytbits, words); if (rc) return rc;
    }
    // stage the float64 network one time slice at a time (bounded staging memory)
    double *dY = nullptr;
    int *dflag = nullptr;
    const size_t slice = (size_t)h->N * h->N;
    HIPCHK(h, hipMalloc((void **)&dY, slice * sizeof(double)));
    HIPCHK(h, hipMalloc((void **)&dflag, sizeof(int)));
    HIPCHK(h, hipMemsetAsync(dflag, 0, sizeof(int), h->stream));
    for (int t = 0; t < h->T; ++t) {
        hipError_t e = hipMemcpyAsync(dY, Y + (size_t)t * slice, slice * sizeof(double),
                                      hipMemcpyHostToDevice, h->stream);
        if (e != hipSuccess) { hipFree(dY); hipFree(dflag); HIPCHK(h, e); }
        hipLaunchKernelGGL(k_pack_bits, dim3(h->N), dim3(256), 0, h->stream, dY, h->N,
                           h->W, 0, h->ybits + (size_t)t * h->N * h->W, dflag);
        if (h->model == DLSM_DIRECTED)
            hipLaunchKernelGGL(k_pack_bits, dim3(h->N), dim3(256), 0, h->stream, dY,
                               h->N, h->W, 1, h->ytbits + (size_t)t * h->N * h->W, dflag);
        hipStreamSynchronize(h->stream);
    }
    int flag = 0;
    hipMemcpy(&flag, dflag, sizeof(int), hipMemcpyDeviceToHost);
    hipFree(dY); hipFree(dflag);
    HIPCHK(h, hipGetLastError());
    if (flag) FAIL(h, DLSM_E_DATA, "network has entries other than 0.0 / 1.0 "
                                   "(missing-edge sampling is not supported)");
    { int rc = build_colmajor(h); if (rc) return rc; }
    h->have_network = true;
    h->have_hops = false;
    return DLSM_OK;
}

// the undirected model's column-block-major copy of the packed words (ChainView::ycm)
static int build_colmajor(dlsm_chain *h) {
    if (h->model != DLSM_UNDIRECTED) return DLSM_OK;
    const int Ncm = (h->N + 127) / 128 * 128;       // = ChainView::Ncm
    const size_t n = (size_t)h->T * (h->W / 2) * Ncm;
    if (!h->ycm) { int rc = dev_alloc(h, &h->ycm, n); if (rc) return rc; }
    hipLaunchKernelGGL(k_pack_colmajor, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->ybits,
                       h->T, h->N, h->W, Ncm, h->ycm);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DLSM_OK;
}

static int64_t packed_words(const dlsm_chain *h) {
    return (int64_t)h->T * h->N * h->W * (h->model == DLSM_DIRECTED ? 2 : 1);
}

int dlsm_network_packed_words(dlsm_chain *h, int64_t *n_words) {
    NEED(h, h && n_words, "null argument");
    NEED(h, h->model != DLSM_DIRECTED_CASE_CONTROL, "case-control chains hold edge lists");
    *n_words = packed_words(h);
    return DLSM_OK;
}

int dlsm_get_network_packed(dlsm_chain *h, uint32_t *buf, int64_t n_words) {
    NEED(h, h && buf, "null argument");
    NEED(h, h->model != DLSM_DIRECTED_CASE_CONTROL, "case-control chains hold edge lists");
    NEED(h, h->have_network, "network not uploaded");
    NEED(h, n_words == packed_words(h), "buffer must hold %lld words", (long long)packed_words(h));
    HIPCHK(h, hipSetDevice(h->device));
    const size_t one = (size_t)h->T * h->N * h->W;
    HIPCHK(h, hipMemcpyAsync(buf, h->ybits, one * sizeof(uint32_t), hipMemcpyDefault, h->stream));
    if (h->model == DLSM_DIRECTED)
        HIPCHK(h, hipMemcpyAsync(buf + one, h->ytbits, one * sizeof(uint32_t), hipMemcpyDefault,
                                 h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DLSM_OK;
}

int dlsm_set_network_packed(dlsm_chain *h, const uint32_t *buf, int64_t n_words) {
    NEED(h, h && buf, "null argument");
    drop_graph(h);
    NEED(h, h->model != DLSM_DIRECTED_CASE_CONTROL, "case-control chains hold edge lists");
    NEED(h, n_words == packed_words(h), "buffer must hold %lld words", (long long)packed_words(h));
    HIPCHK(h, hipSetDevice(h->device));
    const size_t one = (size_t)h->T * h->N * h->W;
    if (!h->ybits) { int rc = dev_alloc(h, &h->ybits, one); if (rc) return rc; }
    if (h->model == DLSM_DIRECTED && !h->ytbits) {
        int rc = dev_alloc(h, &h->ytbits, one); if (rc) return rc;
    }
    HIPCHK(h, hipMemcpyAsync(h->ybits, buf, one * sizeof(uint32_t), hipMemcpyDefault, h->stream));
    if (h->model == DLSM_DIRECTED)
        HIPCHK(h, hipMemcpyAsync(h->ytbits, buf + one, one * sizeof(uint32_t), hipMemcpyDefault,
                                 h->stream));
    // the layout's invariants, checked on the device: padding bits beyond N and the diagonal
    // are zero (the kernels rely on both), the undirected network is symmetric
    int *dflag = (int *)(h->dsmall + 48);
    HIPCHK(h, hipMemsetAsync(dflag, 0, sizeof(int), h->stream));
    hipLaunchKernelGGL(k_check_packed, dim3((unsigned)((size_t)h->T * h->N)), dim3(64), 0, h->stream,
                       h->ybits, h->model == DLSM_DIRECTED ? h->ytbits : h->ybits, h->T, h->N,
                       h->W, dflag);
    HIPCHK(h, hipGetLastError());
    int flag = 0;
    HIPCHK(h, hipMemcpyAsync(&flag, dflag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (flag) FAIL(h, DLSM_E_DATA, "packed network violates the layout (code %d: 1 padding bits, "
                                   "2 diagonal, 4 transpose mismatch)", flag);
    { int rc = build_colmajor(h); if (rc) return rc; }
    h->have_network = true;
    h->have_hops = false;
    return DLSM_OK;
}

int dlsm_upload_edges(dlsm_chain *h, const int64_t *in_edges, int Din,
                      const int64_t *out_edges, int Dout, const int64_t *degree) {
    NEED(h, h && in_edges && out_edges && degree, "null argument");
    drop_graph(h);
    NEED(h, h->model == DLSM_DIRECTED_CASE_CONTROL, "not a case-control chain");
    NEED(h, Din >= 0 && Dout >= 0, "negative padded degree");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t TN = (size_t)h->T * h->N;
    for (size_t i = 0; i < TN; ++i) {
        int64_t a = degree[i * 2], b = degree[i * 2 + 1];
        if (a < 0 || a > Din || b < 0 || b > Dout)
            FAIL(h, DLSM_E_DATA, "degree out of range at node %zu", i);
    }
    // a failed re-upload must not leave new lists behind old strides: the chain has no edges
    // until all three arrays are in
    h->have_edges = false;
    int rc = upload_i64_as_i32(h, &h->in_edges, in_edges, TN * Din, 0, h->N, "in_edges");
    if (rc) return rc;
    rc = upload_i64_as_i32(h, &h->out_edges, out_edges, TN * Dout, 0, h->N, "out_edges");
    if (rc) return rc;
    rc = upload_i64_as_i32(h, &h->degree, degree, TN * 2, 0, (int64_t)h->N, "degree");
    if (rc) return rc;
    h->Din = Din; h->Dout = Dout;
    h->have_edges = true;
    h->cc_terms_valid = false;
    return DLSM_OK;
}

int dlsm_set_controls(dlsm_chain *h, const int64_t *ctrl_in, const int64_t *ctrl_out,
                      int C) {
    NEED(h, h && ctrl_in && ctrl_out && C > 0, "bad argument");
    drop_graph(h);
    NEED(h, h->model == DLSM_DIRECTED_CASE_CONTROL, "not a case-control chain");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t TN = (size_t)h->T * h->N;
    h->have_controls = false;                       // (as dlsm_upload_edges)
    h->nctrl_valid = false; h->cc_terms_valid = false;
    int rc = upload_i64_as_i32(h, &h->ctrl_in, ctrl_in, TN * C, -1, h->N, "control_nodes_in");
    if (rc) return rc;                              // -1 = padding
    rc = upload_i64_as_i32(h, &h->ctrl_out, ctrl_out, TN * C, -1, h->N, "control_nodes_out");
    if (rc) return rc;
    h->C = C;
    h->have_controls = true;
    h->nctrl_valid = false;
    return DLSM_OK;
}

int dlsm_get_controls(dlsm_chain *h, int64_t *ctrl_in, int64_t *ctrl_out) {
    NEED(h, h && ctrl_in && ctrl_out, "null argument");
    NEED(h, h->have_controls, "controls not set");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t n = (size_t)h->T * h->N * h->C;
    std::vector<int32_t> tmp(n);
    int rc = d2h(h, tmp.data(), h->ctrl_in, n); if (rc) return rc;
    for (size_t i = 0; i < n; ++i) ctrl_in[i] = tmp[i];
    rc = d2h(h, tmp.data(), h->ctrl_out, n); if (rc) return rc;
    for (size_t i = 0; i < n; ++i) ctrl_out[i] = tmp[i];
    return DLSM_OK;
}

int dlsm_resample_controls(dlsm_chain *h, uint32_t iter, int n_control) {
    NEED(h, h != nullptr, "null handle");
    drop_graph(h);
    NEED(h, h->model == DLSM_DIRECTED_CASE_CONTROL && h->have_edges,
         "needs a case-control chain with edge lists");
    NEED(h, n_control > 0, "n_control must be positive");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t TN = (size_t)h->T * h->N;
    if (h->C != n_control || !h->ctrl_in) {
        if (h->ctrl_in) hipFree(h->ctrl_in);
        if (h->ctrl_out) hipFree(h->ctrl_out);
        h->ctrl_in = h->ctrl_out = nullptr;
        int rc = dev_alloc(h, &h->ctrl_in, TN * n_control); if (rc) return rc;
        rc = dev_alloc(h, &h->ctrl_out, TN * n_control); if (rc) return rc;
        h->C = n_control;
    }
    ChainView v = h->view();
    const int waves = 4;
    const size_t lds = (size_t)waves * ((h->N + 31) / 32) * sizeof(uint32_t);
    if (lds > 150 * 1024) FAIL(h, DLSM_E_LIMIT, "N too large for the control sampler");
    hipLaunchKernelGGL(k_resample_controls, dim3((unsigned)((TN + waves - 1) / waves)),
                       dim3(64 * waves), lds, h->stream, v, h->ctrl_in, h->ctrl_out,
                       iter);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_controls = true;
    h->nctrl_valid = false; h->cc_terms_valid = false;
    return DLSM_OK;
}

// ---------------------------------------------------------------- state
int dlsm_set_positions(dlsm_chain *h, const double *X) {
    NEED(h, h && X, "null argument");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = h2d(h, h->X, X, (size_t)h->T * h->N * h->D);
    if (!rc) h->have_X = true;
    return rc;
}

int dlsm_get_positions(dlsm_chain *h, double *X) {
    NEED(h, h && X, "null argument");
    NEED(h, h->have_X, "latent positions not set");
    HIPCHK(h, hipSetDevice(h->device));
    return d2h(h, X, h->X, (size_t)h->T * h->N * h->D);
}

int dlsm_set_intercepts(dlsm_chain *h, const double *b, int n) {
    NEED(h, h && b, "null argument");
    NEED(h, n == (h->model == DLSM_UNDIRECTED ? 1 : 2), "wrong number of intercepts");
    HIPCHK(h, hipSetDevice(h->device));
    return h2d(h, h->intercept, b, n);
}

int dlsm_get_intercepts(dlsm_chain *h, double *b, int n) {
    NEED(h, h && b && n >= 1 && n <= 2, "bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    return d2h(h, b, h->intercept, n);
}

int dlsm_set_radii(dlsm_chain *h, const double *radii) {
    NEED(h, h && radii, "null argument");
    NEED(h, h->model != DLSM_UNDIRECTED, "radii belong to directed models");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = h2d(h, h->radii, radii, h->N);
    if (!rc) h->have_radii = true;
    return rc;
}

int dlsm_get_radii(dlsm_chain *h, double *radii) {
    NEED(h, h && radii, "null argument");
    NEED(h, h->have_radii, "radii not set");
    HIPCHK(h, hipSetDevice(h->device));
    return d2h(h, radii, h->radii, h->N);
}

int dlsm_set_squared(dlsm_chain *h, int squared) {
    NEED(h, h != nullptr, "null handle");
    drop_graph(h);
    h->squared = squared ? 1 : 0;
    return DLSM_OK;
}

int dlsm_set_samplers(dlsm_chain *h, const double *step_size, const int32_t *n_accepted,
                      const int32_t *n_steps, const int32_t *steps_until_tune, int tune,
                      int tune_interval) {
    NEED(h, h && step_size && n_accepted && n_steps && steps_until_tune, "null argument");
    drop_graph(h);
    NEED(h, tune_interval > 0, "tune_interval must be positive");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t TN = (size_t)h->T * h->N;
    int rc = h2d(h, h->step, step_size, TN); if (rc) return rc;
    rc = h2d(h, h->nacc, n_accepted, TN); if (rc) return rc;
    rc = h2d(h, h->nsteps, n_steps, TN); if (rc) return rc;
    rc = h2d(h, h->until, steps_until_tune, TN); if (rc) return rc;
    h->tune = tune < 0 ? -1 : tune;
    h->tune_interval = tune_interval;
    h->have_samplers = true;
    return DLSM_OK;
}

int dlsm_get_samplers(dlsm_chain *h, double *step_size, int32_t *n_accepted,
                      int32_t *n_steps, int32_t *steps_until_tune) {
    NEED(h, h && step_size && n_accepted && n_steps && steps_until_tune, "null argument");
    NEED(h, h->have_samplers, "samplers not set");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t TN = (size_t)h->T * h->N;
    int rc = d2h(h, step_size, h->step, TN); if (rc) return rc;
    rc = d2h(h, n_accepted, h->nacc, TN); if (rc) return rc;
    rc = d2h(h, n_steps, h->nsteps, TN); if (rc) return rc;
    return d2h(h, steps_until_tune, h->until, TN);
}

int dlsm_set_prior_random_walk(dlsm_chain *h, double tau_sq, double sigma_sq) {
    NEED(h, h != nullptr, "null handle");
    drop_graph(h);
    NEED(h, tau_sq > 0 && sigma_sq > 0, "variances must be positive");
    h->prior_kind = DLSM_PRIOR_RANDOM_WALK;
    h->tau_sq = tau_sq; h->sigma_sq = sigma_sq;
    h->have_prior = true;
    return DLSM_OK;
}

int dlsm_set_prior_mixture(dlsm_chain *h, const double *mu, const double *sigma,
                           double lmbda, const int64_t *z, int K) {
    NEED(h, h && mu && sigma, "null argument");
    drop_graph(h);
    NEED(h, K >= 1 && K <= 64, "n_components must be in 1..64");
    NEED(h, z || (h->have_prior && h->prior_kind == DLSM_PRIOR_MIXTURE && h->K == K),
         "z = NULL keeps the device's labels: none are there for this n_components");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t TN = (size_t)h->T * h->N;
    for (size_t i = 0; z && i < TN; ++i)
        if (z[i] < 0 || z[i] >= K) FAIL(h, DLSM_E_DATA, "label out of range at %zu", i);
    if (h->K != K) {
        if (h->mu) hipFree(h->mu);
        if (h->sigma) hipFree(h->sigma);
        h->mu = h->sigma = nullptr;
        int rc = dev_alloc(h, &h->mu, (size_t)K * h->D); if (rc) return rc;
        rc = dev_alloc(h, &h->sigma, K); if (rc) return rc;
        h->K = K;
    }
    size_t staged = 0;                          // one synchronisation for the whole upload
    int rc = h2d_enqueue(h, h->mu, mu, (size_t)K * h->D, &staged); if (rc) return rc;
    rc = h2d_enqueue(h, h->sigma, sigma, (size_t)K, &staged); if (rc) return rc;
    rc = h2d_enqueue(h, &h->hdp->lmbda, &lmbda, (size_t)1, &staged); if (rc) return rc;
    std::vector<int32_t> zz;
    if (z) {
        zz.resize(TN);
        for (size_t i = 0; i < TN; ++i) zz[i] = (int32_t)z[i];
        rc = h2d_enqueue(h, h->z, zz.data(), TN, &staged); if (rc) return rc;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->lmbda = lmbda;
    h->prior_kind = DLSM_PRIOR_MIXTURE;
    h->have_prior = true;
    return DLSM_OK;
}

// ---------------------------------------------------------------- log-lik
int dlsm_loglik_full(dlsm_chain *h, int m, const double *intercepts, double *out) {
    NEED(h, h && out, "null argument");
    NEED(h, m >= 1 && m <= 16, "m must be in 1..16");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = check_ready_loglik(h); if (rc) return rc;
    const int nic = h->model == DLSM_UNDIRECTED ? 1 : 2;
    if (!intercepts) NEED(h, m == 1, "intercepts == NULL requires m == 1");
    for (int k0 = 0; k0 < m; k0 += 2) {
        const int M = std::min(2, m - k0);
        const double *d_ic = h->intercept;
        if (intercepts) {
            memcpy(h->hsmall, intercepts + (size_t)k0 * nic, sizeof(double) * M * nic);
            HIPCHK(h, hipMemcpyAsync(h->dsmall, h->hsmall, sizeof(double) * M * nic,
                                     hipMemcpyHostToDevice, h->stream));
            d_ic = h->dsmall;
        }
        int nrec = 0;
        rc = loglik_records(h, M, d_ic, h->radii, h->radii, &nrec); if (rc) return rc;
        hipLaunchKernelGGL(k_reduce_loglik, dim3(1), dim3(256), 0, h->stream, h->partials,
                           nrec, h->model, M, d_ic, h->dsmall + 16);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(h->hsmall + 16, h->dsmall + 16, sizeof(double) * M,
                                 hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        for (int k = 0; k < M; ++k) out[k0 + k] = h->hsmall[16 + k];
    }
    return DLSM_OK;
}

int dlsm_loglik_full_radii(dlsm_chain *h, const double *radii_alt, double *out) {
    NEED(h, h && radii_alt && out, "null argument");
    NEED(h, h->model != DLSM_UNDIRECTED, "radii belong to directed models");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = check_ready_loglik(h); if (rc) return rc;
    rc = h2d(h, h->radii_alt, radii_alt, h->N); if (rc) return rc;
    // both candidates use the current intercepts
    HIPCHK(h, hipMemcpyAsync(h->dsmall, h->intercept, 2 * sizeof(double),
                             hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->dsmall + 2, h->intercept, 2 * sizeof(double),
                             hipMemcpyDeviceToDevice, h->stream));
    int nrec = 0;
    rc = loglik_records(h, 2, h->dsmall, h->radii, h->radii_alt, &nrec); if (rc) return rc;
    hipLaunchKernelGGL(k_reduce_loglik, dim3(1), dim3(256), 0, h->stream, h->partials, nrec,
                       h->model, 2, h->dsmall, h->dsmall + 16);
    HIPCHK(h, hipGetLastError());
    return d2h(h, out, h->dsmall + 16, 2);
}

}  // extern "C"

template <int DD>
static int launch_partial(dlsm_chain *h, int with_prior, int t, int j, const double *d_x,
                          double *d_out) {
    ChainView v = h->view();
    dim3 grid = t >= 0 ? dim3(1, 1) : dim3(h->N, h->T);
    hipLaunchKernelGGL((k_partial_all<DD>), grid, dim3(256), 0, h->stream, v, with_prior,
                       t, j, d_x, d_out);
    HIPCHK(h, hipGetLastError());
    return DLSM_OK;
}

static int check_ready_partial(dlsm_chain *h, int with_prior) {
    int rc = check_ready_loglik(h); if (rc) return rc;
    if (with_prior) NEED(h, h->have_prior, "prior not set");
    return DLSM_OK;
}

extern "C" {

int dlsm_loglik_partial(dlsm_chain *h, int t, int j, const double *x, int with_prior,
                        double *out) {
    NEED(h, h && out, "null argument");
    NEED(h, t >= 0 && t < h->T && j >= 0 && j < h->N, "node (t=%d, j=%d) out of range", t, j);
    HIPCHK(h, hipSetDevice(h->device));
    int rc = check_ready_partial(h, with_prior); if (rc) return rc;
    const double *d_x = nullptr;
    if (x) {
        memcpy(h->hsmall, x, sizeof(double) * h->D);
        HIPCHK(h, hipMemcpyAsync(h->dsmall, h->hsmall, sizeof(double) * h->D,
                                 hipMemcpyHostToDevice, h->stream));
        d_x = h->dsmall;
    }
    DISPATCH_D(h, h->D, rc = launch_partial<DD>(h, with_prior, t, j, d_x, h->dsmall + 16));
    if (rc) return rc;
    return d2h(h, out, h->dsmall + 16, 1);
}

int dlsm_loglik_partial_all(dlsm_chain *h, int with_prior, double *out) {
    NEED(h, h && out, "null argument");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = check_ready_partial(h, with_prior); if (rc) return rc;
    const size_t TN = (size_t)h->T * h->N;
    rc = ensure_partials(h, TN); if (rc) return rc;
    DISPATCH_D(h, h->D, rc = launch_partial<DD>(h, with_prior, -1, -1, nullptr, h->partials));
    if (rc) return rc;
    return d2h(h, out, h->partials, TN);
}

// ---------------------------------------------------------------- sweep
}  // extern "C"


// algo 2 / 3: rounds of (chip-wide eval, per-slice resolve) over super-batches of
// S sub-batches of <= 128 nodes (S = 1: algo 2)
// When profiling, the eval launch carries its own start / stop events
// (hipExtLaunchKernelGGL): they read the dispatch's begin / end timestamps, i.e. the
// same kernel duration the rocprofv3 kernel trace reports, without the dispatch
// latency a bracket of hipEventRecord calls would include.
template <int DD, int SBM>
static void launch_spec_eval(dlsm_chain *h, const ChainView &v, const SpecBuf &sg, dim3 grid,
                             hipStream_t q, int parity, int j0, int nsb) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->profiling) { hipEventCreate(&e0); hipEventCreate(&e1); }
    if (h->model == DLSM_UNDIRECTED)
        hipExtLaunchKernelGGL((k_spec_eval<DD, DLSM_UNDIRECTED, SBM>), grid, dim3(SP_EV_THREADS),
                              0, q, e0, e1, 0, v, sg, parity, j0, nsb);
    else
        hipExtLaunchKernelGGL((k_spec_eval<DD, DLSM_DIRECTED, SBM>), grid, dim3(SP_EV_THREADS),
                              0, q, e0, e1, 0, v, sg, parity, j0, nsb);
    if (h->profiling) h->prof[DLSM_K_SWEEP_EVAL].pending.emplace_back(e0, e1);
}

template <int DD>
static int launch_sweep_spec(dlsm_chain *h, IterRef iter, int S, bool alloc_only = false) {
    const int N = h->N, T = h->T;
    if (h->model == DLSM_DIRECTED_CASE_CONTROL) S = 1;          // one wave per node there
    S = std::max(1, std::min(S, SP_SMAX));
    const int B = std::min(S * SP_BMAX, (N + 1) / 2 * 2);       // even: double2 staging
    const int nsl_max = (T + 1) / 2;
    int parts = (1024 + nsl_max * SP_BMAX - 1) / (nsl_max * SP_BMAX);
    if (getenv("DLSM_SPEC_PARTS")) parts = atoi(getenv("DLSM_SPEC_PARTS"));
    parts = std::max(1, std::min(parts, 8));
    if (h->model == DLSM_DIRECTED_CASE_CONTROL) parts = 1;
    auto even2 = [](size_t n) { return (n + 1) / 2 * 2; };
    const size_t n_full0 = even2((size_t)nsl_max * B * parts);
    const size_t n_prop = even2((size_t)nsl_max * N * (DD + 2));
    const size_t n_ht = (size_t)nsl_max * B * B;
    const size_t need = (n_full0 + n_prop + n_ht + 2) * sizeof(double);
    if (h->spec_cap < need) {
        if (h->spec) hipFree(h->spec);
        h->spec = nullptr; h->spec_cap = 0;
        HIPCHK(h, hipMalloc((void **)&h->spec, need));
        h->spec_cap = need;
    }
    SpecBuf sb;
    sb.full0 = h->spec; sb.prop = sb.full0 + n_full0; sb.Ht = sb.prop + n_prop;
    sb.consts = sb.Ht + n_ht;
    sb.B = B; sb.parts = parts; sb.s0 = 0; sb.per = (N + parts - 1) / parts;
    sb.stamps = nullptr;
    ChainView v = h->view();
    auto resolve = k_spec_resolve<DD>;
    const size_t lds = (size_t)SP_BMAX * SP_BMAX * sizeof(double);
    HIPCHK(h, hipFuncSetAttribute((const void *)resolve,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (alloc_only) return DLSM_OK;
    // The slices of a parity are independent, so they can be split over two queues to
    // let one group's resolve overlap the other group's eval.  Measured on MI355X this
    // buys nothing at N=2000 (twice the launches: host bound) and 3.6 % at N=4000 (the
    // 16-wave resolve workgroup is not placed while eval saturates the CUs), so it is
    // opt-in: DLSM_SPEC_QUEUES=2.
    int nq = (getenv("DLSM_SPEC_QUEUES") ? atoi(getenv("DLSM_SPEC_QUEUES")) : 1);
    if (h->profiling || nsl_max < 2) nq = 1;
    if (nq > 1 && !h->stream2) {
        HIPCHK(h, hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_a, hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_b, hipEventDisableTiming));
    }
    hipStream_t qs[2] = {h->stream, nq > 1 ? h->stream2 : h->stream};
    if (nq > 1) {       // fork
        HIPCHK(h, hipEventRecord(h->ev_a, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->stream2, h->ev_a, 0));
    }
    for (int parity = 0; parity < 2; ++parity) {
        const int nsl = (T - parity + 1) / 2;
        if (nsl <= 0) continue;
        const int ng = (nq > 1 && nsl >= 2) ? 2 : 1;
        const int first[3] = {0, ng == 2 ? (nsl + 1) / 2 : nsl, nsl};
        for (int g = 0; g < ng; ++g) {
            SpecBuf sg = sb; sg.s0 = first[g];
            hipLaunchKernelGGL((k_spec_propose<DD>), dim3((N + 255) / 256, first[g + 1] - first[g]),
                               dim3(256), 0, qs[g], v, sg, iter, parity);
        }
        for (int j0 = 0; j0 < N; j0 += B) {
            const int nsb = std::min(B, N - j0);
            for (int g = 0; g < ng; ++g) {
                SpecBuf sg = sb; sg.s0 = first[g];
                const int ns = first[g + 1] - first[g];
                const size_t nblk = (size_t)parts * nsb * ns;
                if (h->profiling && h->stamps && h->model != DLSM_DIRECTED_CASE_CONTROL &&
                    h->stamps_used + nblk <= h->stamps_cap) {
                    sg.stamps = h->stamps + 2 * h->stamps_used;
                    h->stamp_launches.emplace_back(h->stamps_used, nblk);
                    h->stamps_used += nblk;
                }
                {
                    if (h->model == DLSM_DIRECTED_CASE_CONTROL) {
                        ProfScope pe(h, DLSM_K_SWEEP_EVAL);
                        hipLaunchKernelGGL((k_spec_eval_cc<DD>), dim3((unsigned)(ns * nsb)),
                                           dim3(64), 0, qs[g], v, sg, h->nctrl, parity, j0, nsb);
                    } else if (S == 1)
                        launch_spec_eval<DD, SP_BMAX>(h, v, sg, dim3(parts, nsb, ns), qs[g],
                                                      parity, j0, nsb);
                    else if (S == 2)
                        launch_spec_eval<DD, 2 * SP_BMAX>(h, v, sg, dim3(parts, nsb, ns), qs[g],
                                                          parity, j0, nsb);
                    else
                        launch_spec_eval<DD, SP_SMAX * SP_BMAX>(h, v, sg, dim3(parts, nsb, ns),
                                                                qs[g], parity, j0, nsb);
                }
                {
                    ProfScope pr(h, DLSM_K_SWEEP_RESOLVE);
                    hipLaunchKernelGGL(resolve, dim3(ns), dim3(SP_RES_THREADS), lds, qs[g], v, sg,
                                       parity, j0, nsb);
                }
            }
        }
        if (nq > 1) {   // the next parity (or the caller) needs every slice of this one
            HIPCHK(h, hipEventRecord(h->ev_b, h->stream2));
            HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_b, 0));
            if (parity == 0) {
                HIPCHK(h, hipEventRecord(h->ev_a, h->stream));
                HIPCHK(h, hipStreamWaitEvent(h->stream2, h->ev_a, 0));
            }
        }
    }
    HIPCHK(h, hipGetLastError());
    return DLSM_OK;
}


// algo 0 = auto: the pipelined speculative sweep once a slice has several batches
// (exact likelihoods), speculative batches for wide case-control slices, else one
// workgroup per slice
static int resolve_sweep_algo(const dlsm_chain *h, int algo) {
    if (algo != 0) return algo;
    if (h->D > DLSM_D_PIPE_MAX) return h->N >= 256 ? 2 : 1;       // (none today: the pipelined sweep is built to DLSM_D_MAX)
    // case-control: sparse correction lists once a slice has several batches of 512
    if (h->model == DLSM_DIRECTED_CASE_CONTROL)
        return (h->N >= 2048 && h->D <= DLSM_D_CCPIPE_MAX) ? 5 : (h->N >= 512 ? 4 : (h->N >= 256 ? 2 : 1));
    return h->N >= 512 ? 4 : (h->N >= 256 ? 2 : 1);
}

// algo 4: one fused launch per batch, resolve(b) beside eval(b + 1) (kernels_spec_pipe.hpp)
template <int DD, int MODEL, int G = 1>
static void launch_pipe_step(dlsm_chain *h, const ChainView &v, const PipeBuf &pb, dim3 grid,
                             size_t lds, int l) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->profiling) { hipEventCreate(&e0); hipEventCreate(&e1); }
    hipExtLaunchKernelGGL((k_pipe_step<DD, MODEL, G>), grid, dim3(PP_THREADS), lds, h->stream,
                          e0, e1, 0, v, pb, l);
    if (h->profiling) h->prof[DLSM_K_SWEEP_EVAL].pending.emplace_back(e0, e1);
}

// the sticky error word: a bounded in-kernel wait ran out of its budget
static int check_pipe_err(dlsm_chain *h) {
    if (g_fix_err_host && *(volatile int32_t *)g_fix_err_host != 0) {
        const int32_t e = *(volatile int32_t *)g_fix_err_host;
        *(volatile int32_t *)g_fix_err_host = 0;
        FAIL(h, DLSM_E_HIP, "case-control sweep: a resolver's fixed point did not settle inside its spin bound "
             "(word %#x) - the state of this process's case-control chains is undefined; set it again", e);
    }
    if (h->fork_err_host) {
        // (behind a synchronisation of the handle's stream: the word is host memory the device stores to)
        const int32_t e = *(volatile int32_t *)h->fork_err_host;
        if (e != 0) {
            *(volatile int32_t *)h->fork_err_host = 0;
            if (e & PP_ERR_XSERVE)
                FAIL(h, DLSM_E_HIP, "pipelined sweep: a resolver ran out of its poll budget waiting for a cross product "
                     "from the launch's evaluators (word %#x) - the chain's state is undefined; set the state again "
                     "and run with DLSM_PIPE_XSERVE=0", e);
            if (e & CC_ERR_HELPER)
                FAIL(h, DLSM_E_HIP, "case-control sweep: a resolver ran out of its poll budget waiting for its helper "
                     "workgroup (word %#x) - the chain's state is undefined; set the state again and run with "
                     "DLSM_CC_HELPERS=0", e);
            FAIL(h, DLSM_E_HIP, "HDP-LPCM loop: a hand-over between the chain's two queues ran out of its poll "
                 "budget (flags %d) - the chain's state is undefined (more streams alive than hardware "
                 "queues?); set the state again and run with DLSM_HDP_QUEUES=1", e);
        }
    }
    return DLSM_OK;
}

// one batch resolved (and one evaluated) per launch (the kernels' template parameter G = 1: the two-batch
// form, algo 6, and the single persistent launch, algo 7, measured slower and were removed in round 5)
template <int DD>
static int launch_sweep_pipe(dlsm_chain *h, IterRef iter, bool alloc_only = false) {
    constexpr int G = 1;
    const int N = h->N, T = h->T;
    const int nbat = (N + PP_B - 1) / PP_B;
    const bool cc = h->model == DLSM_DIRECTED_CASE_CONTROL;
    int ne_wg = std::max(h->n_cu / 2, h->n_cu - T);
    int parts = (int)((double)ne_wg * PP_WAVES / ((double)T * PP_B) + 0.5);
    if (getenv("DLSM_PIPE_PARTS")) parts = atoi(getenv("DLSM_PIPE_PARTS"));
    parts = std::max(1, std::min(parts, PP_MAXPARTS));
    // (undirected model: the evaluators of kernels_pipe_lds.hpp take ONE round of items - a part less when
    // rounding up would leave items for a second round)
    if (h->model == DLSM_UNDIRECTED && parts > 1 && (long)parts * T * PP_B > (long)ne_wg * PP_WAVES &&
        !getenv("DLSM_PIPE_PARTS"))
        --parts;
    if (cc) {           // CC_PARTS wavefronts per node, four nodes per workgroup round
        parts = CC_PARTS;
        ne_wg = std::min(ne_wg, (T * PP_B * CC_PARTS + PP_WAVES - 1) / PP_WAVES);
    }
    auto even2 = [](size_t n) { return (n + 1) / 2 * 2; };
    const size_t n_prop = even2((size_t)T * N * (2 * DD + 2));
    const int xr = (2 * G - 1) * PP_B;
    const size_t n_full0 = (size_t)2 * G * T * PP_B * parts * 2;
    const size_t n_h = (size_t)2 * G * T * PP_B * PP_B;
    const size_t n_hx = (size_t)2 * G * T * xr * PP_B;
    const size_t n_acc = even2(((size_t)T * 2 * G * PP_ACC + 1) / 2);   // int32 pairs
    const size_t n_xp = (size_t)T * PP_B;
    const size_t need = (n_prop + n_full0 + n_h + n_hx + n_acc + 2 + n_xp) * sizeof(double);
    if (h->pipe_cap < need) {
        if (h->pipe) hipFree(h->pipe);
        h->pipe = nullptr; h->pipe_cap = 0;
        HIPCHK(h, hipMalloc((void **)&h->pipe, need));
        h->pipe_cap = need;
    }
    PipeBuf pb{};
    pb.prop = h->pipe; pb.full0 = pb.prop + n_prop; pb.Hd = pb.full0 + n_full0;
    pb.Hx = pb.Hd + n_h; pb.acc = (int32_t *)(pb.Hx + n_hx);
    pb.consts = pb.Hx + n_hx + n_acc;
    pb.xprod = pb.consts + 2;
    pb.sync = nullptr; pb.nsync = 0; pb.queue0 = ne_wg;
    pb.lsm_draw = h->loop_draws_intercept ? h->lsm : nullptr;
    pb.parts = parts; pb.nbat = nbat;
    pb.G = G; pb.xr = xr;
    pb.per = ((N + parts - 1) / parts + 63) / 64 * 64;      // parts start on a 64-neighbour boundary
    pb.nctrl = h->nctrl;
    // the resolvers' diagonal block by rows of PR_LD (an odd stride: row_resolve), the evaluators' exp table
    const size_t lds = (size_t)PP_B * PR_LD * sizeof(double);
    // undirected model: rows staged in LDS, interleaved parts, H factors inside the trips (kernels_pipe_lds.hpp)
    // whenever the longest part's rows fit beside the exp table
    // and the launch's wavefronts cover its items in one round
    pb.lds_eval = h->model == DLSM_UNDIRECTED && pipe_lds_eval_bytes(N, DD, parts) <= lds &&
                  (long)parts * T * PP_B <= (long)ne_wg * PP_WAVES &&
                  !(getenv("DLSM_PIPE_LDS") && atoi(getenv("DLSM_PIPE_LDS")) == 0);
    // the resolvers' cross products by the evaluators (pipe_xserve_*): one wavefront in xstride takes a row.  A
    // resolver waits INSIDE the launch for wavefronts that wait for nothing - they only have to start
    {
        const char *ex = getenv("DLSM_PIPE_XSERVE"), *ebud = getenv("DLSM_PIPE_XBUDGET");
        pb.err = h->fork_err_dev;
        pb.budget = ebud ? atoi(ebud) : (1 << 22);
        // (one chain on the device: with several, a resolver's wait for evaluator workgroups that other chains' launches
        // keep off the CUs costs more than the cross block - four chains: 6100 it/s served, 6970 not; DLSM_PIPE_XSERVE=2
        // forces it)
        pb.xserve = pb.lds_eval && pb.err != nullptr && T < 128 &&
                    (ex ? (atoi(ex) == 2 || (atoi(ex) != 0 && g_live_chains.load() == 1)) : g_live_chains.load() == 1);
    }
    const int xserve_sweep = pb.xserve;         // (a launch whose workgroups cannot cover the rows keeps the resolvers' own products)
    pb.xserve = 0;
    ChainView v = h->view();
    {   // the evaluators' arguments (kernels_pipe_lds.hpp) in one piece
        PipeLds &a = pb.lds;
        a.X = v.X; a.ybits = v.ybits; a.prop = pb.prop; a.full0 = pb.full0; a.Hd = pb.Hd; a.acc = pb.acc;
        a.consts = pb.consts; a.xprod = pb.xprod;
        a.T = T; a.N = N; a.W = v.W; a.squared = v.squared; a.parts = parts; a.nbat = nbat;
        a.lds_cap = pipe_lds_trip_cap((N + 63) / 64, parts); a.xserve = 0;
    }
    for (int nw = 1; nw <= 4; ++nw)
        for (int p = 0; p < PP_MAXPARTS; ++p)
            pb.lds.plan[nw - 1][p] = p < parts ? pipe_plan_entry((N + 63) / 64, parts, nw, p) : 0u;
    auto ku = k_pipe_step<DD, DLSM_UNDIRECTED>;
    auto kl = k_pipe_step<DD, PIPE_UNDIRECTED_LONG>;
    auto kd = k_pipe_step<DD, DLSM_DIRECTED>;
    auto kc = k_pipe_step<DD, DLSM_DIRECTED_CASE_CONTROL>;
    HIPCHK(h, hipFuncSetAttribute((const void *)ku, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
    HIPCHK(h, hipFuncSetAttribute((const void *)kl, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
    HIPCHK(h, hipFuncSetAttribute((const void *)kd, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
    HIPCHK(h, hipFuncSetAttribute((const void *)kc, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds));
    HIPCHK(h, hipFuncSetAttribute((const void *)k_pipe_last_ride<DD>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (alloc_only) return DLSM_OK;
    {   // the proposal pass, unless the previous iteration's last launch carried it
        const ProposeBuf nb{pb.prop, pb.consts, pb.sync, pb.nsync, pb.queue0, pb.lsm_draw};
        const bool drawn = !iter.ptr && h->prop_drawn_for == (long)iter.value && h->next_prop_ok &&
                           h->next_prop.prop == nb.prop && h->next_prop.sync == nb.sync &&
                           h->next_prop.queue0 == nb.queue0 && h->next_prop.lsm_draw == nb.lsm_draw;
        if (!drawn)
            hipLaunchKernelGGL((k_pipe_propose<DD>), dim3((N + 255) / 256, T), dim3(256), 0, h->stream, v,
                               pb, iter);
        h->next_prop = nb; h->next_prop_ok = true; h->prop_drawn_for = -1; h->pipe_touched = true;
    }
    // launch l: even slices resolve batches G l .. / evaluate G (l + 1) .., odd slices one launch
    // behind; with a single slice (T == 1) the trailing odd-only launch is empty
    const int nlaunch = (nbat + G - 1) / G;
    const int last = T > 1 ? nlaunch : nlaunch - 1;
    for (int l = -1; l <= last; ++l) {
        // (the head - the proposal pass above and launch -1 - may have been enqueued on the chain's second
        // queue already, beside the previous iteration's conjugate draws: capi_hdp.hpp)
        if (l == -1 && !iter.ptr && h->head_done_for == (long)iter.value) continue;
        if (h->sweep_part == 1 && l >= 0) break;
        const bool any_eval = (G * (l + 1) < nbat) || (T > 1 && l >= 0 && G * l < nbat);
        const int grid = T + (any_eval ? ne_wg : 0);
        const bool lng = !pb.lds_eval;       // (undirected model without the LDS evaluators: pipe_eval_item's pipelined trips)
        if (l == last && !any_eval && h->post_ride_want && G == 1 && h->model == DLSM_UNDIRECTED &&
            !h->profiling && l >= 0 && T + 4 <= PS_BLOCKS) {
            // the centring sums ride in the resolve-only launch (kernels_spec_pipe.hpp): nwg rider
            // records + one per slice from the resolvers
            const long rows = (long)T * N;
            const int nwg = (int)std::min<long>(PS_BLOCKS - T, (rows + PP_THREADS - 1) / PP_THREADS);
            PipePostRide pr{h->post_ride_xref, iter, h->partials + (size_t)ll_blocks(h) * 4, nwg,
                            (nbat - 1) * PP_B, T > 1 ? 1 : 0};
            hipLaunchKernelGGL((k_pipe_last_ride<DD>), dim3(T + nwg), dim3(PP_THREADS), lds, h->stream, v, pb,
                               l, pr);
            h->post_ride_done = true; h->post_ride_nwg = nwg + T; h->post_ride_jl = pr.jl; h->post_ride_par = pr.par;
            continue;
        }
        if (h->model == DLSM_UNDIRECTED && lng)
            launch_pipe_step<DD, PIPE_UNDIRECTED_LONG>(h, v, pb, dim3(grid), lds, l);
        else if (h->model == DLSM_UNDIRECTED && pb.lds_eval && any_eval) {
            // kernels_pipe_lds.hpp: plane 0 = the resolvers (padded to the planes' size), plane p + 1 = part p's
            // evaluators, 16 consecutive nodes (x) of one active slice (y) per workgroup
            const int beE = l + 1, beO = l, nE = (T + 1) / 2, nO = T / 2;
            PipeLds &a = pb.lds;
            a.beE = beE; a.beO = beO;
            a.nbE = (beE >= 0 && beE < nbat) ? std::min(PP_B, N - beE * PP_B) : 0;
            a.nbO = (beO >= 0 && beO < nbat) ? std::min(PP_B, N - beO * PP_B) : 0;
            a.nslE = a.nbE > 0 ? nE : 0; a.nslO = a.nbO > 0 ? nO : 0;
            const int gx = PP_B / PP_WAVES, nsl = a.nslE + a.nslO;
            a.nsl_magic = (65536u + (uint32_t)nsl - 1u) / (uint32_t)nsl;
            // serving wavefronts per evaluator workgroup (pipe_xserve_request): every launched one can serve
            a.xstride = (T * PP_B + ne_wg - 1) / ne_wg;
            pb.xserve = a.xserve = xserve_sweep && a.xstride <= PP_WAVES;
            (void)gx;
            launch_pipe_step<DD, DLSM_UNDIRECTED>(h, v, pb, dim3(grid), lds, l);
        } else if (h->model == DLSM_UNDIRECTED)
            launch_pipe_step<DD, DLSM_UNDIRECTED>(h, v, pb, dim3(grid), lds, l);
        else if (h->model == DLSM_DIRECTED)
            launch_pipe_step<DD, DLSM_DIRECTED>(h, v, pb, dim3(grid), lds, l);
        else
            launch_pipe_step<DD, DLSM_DIRECTED_CASE_CONTROL>(h, v, pb, dim3(grid), lds, l);
    }
    HIPCHK(h, hipGetLastError());
    return DLSM_OK;
}

// algo 5: the case-control likelihood with sparse correction lists, batches of CP_B = 512 nodes
// (kernels_ccpipe.hpp)
template <int DD>
static int launch_sweep_ccpipe(dlsm_chain *h, IterRef iter, bool alloc_only = false) {
    const int N = h->N, T = h->T;
    const int nbat = (N + CP_B - 1) / CP_B;
    const int cap = std::max(1, h->Din + h->Dout + 2 * h->C);   // either list may hold every term
    auto even2 = [](size_t n) { return (n + 1) / 2 * 2; };
    const size_t n_prop = even2((size_t)T * N * (2 * DD + 2));
    const size_t n_tot = (size_t)2 * T * CP_B;
    const size_t n_ent = (size_t)2 * T * cap * CP_B;
    const size_t n_cnt = (size_t)2 * T * CP_B * 2;
    const size_t n_rec = (size_t)T * N * cp_record_width(DD);
    // doubles: prop | tot | xval | oval | consts(2) | cur | snap | xsum ; int32: xidx | oidx | cnt ;
    // uint64: accmask
    const size_t n_xsum = (size_t)T * CP_B;
    const size_t need = (n_prop + n_tot + 2 * n_ent + 2 + 2 * n_rec + n_xsum) * sizeof(double) +
                        even2(2 * n_ent + n_cnt) * sizeof(int32_t) +
                        (size_t)T * CP_WAVES * sizeof(unsigned long long);
    if (h->pipe_cap < need) {
        if (h->pipe) hipFree(h->pipe);
        h->pipe = nullptr; h->pipe_cap = 0;
        HIPCHK(h, hipMalloc((void **)&h->pipe, need));
        h->pipe_cap = need;
    }
    // (the gathers address records and proposals as 32-bit lane offsets from per-slice bases: umul24 of the node
    // and a snapshot offset of T N records - round-5 advice)
    if (N >= (1 << 24) || (double)(T + 1) * N * cp_record_width(DD) * sizeof(double) >= 4294967296.0)
        FAIL(h, DLSM_E_LIMIT, "case-control sweep (algo 5): N=%d T=%d is beyond its 32-bit gather offsets "
             "(N < 2^24, (T + 1) N record bytes < 2^32); use sweep_algo 4 or 2", N, T);
    // the nodes' term rows: rebuilt when the edge tables or the controls have changed
    { int rc_ = ensure_cc_rows(h, alloc_only); if (rc_) return rc_; }
    const int tw = h->cc_tw;
    if (alloc_only) return DLSM_OK;
    CcPipeBuf pb;
    pb.prop = h->pipe; pb.tot = pb.prop + n_prop; pb.xval = pb.tot + n_tot; pb.oval = pb.xval + n_ent;
    double *consts = pb.oval + n_ent;
    pb.cur = consts + 2; pb.snap = pb.cur + n_rec;
    pb.xsum = pb.snap + n_rec;
    pb.xidx = (int32_t *)(pb.xsum + n_xsum); pb.oidx = pb.xidx + n_ent; pb.cnt = pb.oidx + n_ent;
    pb.accmask = (unsigned long long *)(pb.xidx + even2(2 * n_ent + n_cnt));
    // a helper workgroup per resolver (kernels_ccpipe.hpp, ccpipe_cross_helper) when the launch still fits the
    // device in one wave of workgroups - a resolver waits for its helper INSIDE the launch, so both must be
    // resident; DLSM_CC_HELPERS=0 keeps the resolvers on their own
    const char *eh = getenv("DLSM_CC_HELPERS");
    // (one chain on the device - as hdp_fork_arm and the pipelined sweep's served cross products decide: with several
    // chains' launches on the CUs a helper may not be resident when its resolver starts to poll, and a slow but
    // correct run would end in the sticky error; DLSM_CC_HELPERS=2 forces the role, 0 switches it off)
    pb.helpers = (eh ? (atoi(eh) == 2 || (atoi(eh) != 0 && g_live_chains.load() == 1)) : g_live_chains.load() == 1) &&
                 h->n_cu >= 4 * T && h->fork_err_dev != nullptr;
    {   // polls of a resolver's wait for its helper before the sticky error word is set (DLSM_CC_HELPER_BUDGET)
        const char *ebud = getenv("DLSM_CC_HELPER_BUDGET");
        pb.budget = ebud ? atoi(ebud) : (1 << 22);
    }
    pb.err = h->fork_err_dev;
    pb.nctrl = h->nctrl; pb.cap = cap; pb.nbat = nbat;
    pb.terms = h->cc_terms; pb.tw = tw;
    PipeBuf pp{};                   // the proposal kernel's view: proposals + its two constants
    pp.prop = pb.prop; pp.consts = consts;
    ChainView v = h->view();
    {   // the proposal pass, unless the previous iteration's last launch carried it
        const ProposeBuf nb{pp.prop, pp.consts, nullptr, 0, 0, nullptr};
        const bool drawn = !iter.ptr && h->prop_drawn_for == (long)iter.value && h->next_prop_ok &&
                           h->next_prop.prop == nb.prop && h->next_prop.sync == nullptr;
        if (!drawn)
            hipLaunchKernelGGL((k_pipe_propose<DD>), dim3((N + 255) / 256, T), dim3(256), 0, h->stream, v,
                               pp, iter);
        h->next_prop = nb; h->next_prop_ok = true; h->prop_drawn_for = -1; h->pipe_touched = true;
    }
    hipLaunchKernelGGL((k_ccpipe_pack<DD>), dim3((unsigned)(((size_t)T * N + 255) / 256)), dim3(256), 0,
                       h->stream, v, pb);
    const int nodes_max = ((T + 1) / 2 + T / 2) * std::min(CP_B, N);
    // one evaluator workgroup per remaining CU; the items are dealt out over all of them (kernels_ccpipe.hpp)
    const int n_front = pb.helpers ? 2 * T : T;         // resolvers (+ their helpers) in front of the evaluators
    const int ne_wg = std::max(1, std::min(std::max(h->n_cu / 2, h->n_cu - n_front), nodes_max));
    const int last = T > 1 ? nbat : nbat - 1;
    for (int l = -1; l <= last; ++l) {
        const bool any_eval = (l + 1 < nbat) || (T > 1 && l >= 0 && l < nbat);
        const int grid = n_front + (any_eval ? ne_wg : 0);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (h->profiling) { hipEventCreate(&e0); hipEventCreate(&e1); }
        hipExtLaunchKernelGGL((k_ccpipe_step<DD>), dim3(grid), dim3(CP_THREADS), 0, h->stream, e0, e1,
                              0, v, pb, l);
        if (h->profiling) h->prof[DLSM_K_SWEEP_EVAL].pending.emplace_back(e0, e1);
    }
    HIPCHK(h, hipGetLastError());
    return DLSM_OK;
}

template <int DD>
static int launch_sweep(dlsm_chain *h, IterRef iter, int algo, bool alloc_only = false) {
    ChainView v = h->view();
    ProfScope ps(h, DLSM_K_SWEEP);
    if (h->model == DLSM_DIRECTED_CASE_CONTROL) {
        // number of valid (non -1) controls per node and direction, and the nodes' term rows
        { int rc_ = ensure_cc_rows(h, alloc_only); if (rc_) return rc_; }
        algo = resolve_sweep_algo(h, algo);
        if (alloc_only) {
            hipStreamSynchronize(h->stream);
            if constexpr (DD <= DLSM_D_CCPIPE_MAX) {
                if (algo == 5) return launch_sweep_ccpipe<DD>(h, iter, true);
            }
            if constexpr (DD <= DLSM_D_PIPE_MAX) {
                if (algo == 4) return launch_sweep_pipe<DD>(h, iter, true);
            }
            return algo >= 2 ? launch_sweep_spec<DD>(h, iter, 1, true) : DLSM_OK;
        }
        if constexpr (DD <= DLSM_D_CCPIPE_MAX) {
            if (algo == 5) return launch_sweep_ccpipe<DD>(h, iter);
        }
        if constexpr (DD <= DLSM_D_PIPE_MAX) {
            if (algo == 4) return launch_sweep_pipe<DD>(h, iter);
        }
        if (algo >= 2) return launch_sweep_spec<DD>(h, iter, 1);
        for (int parity = 0; parity < 2; ++parity) {
            int nsl = (h->T - parity + 1) / 2;
            if (nsl <= 0) continue;
            hipLaunchKernelGGL((k_sweep_casecontrol<DD>), dim3(nsl), dim3(CC_THREADS), 0,
                               h->stream, v, h->nctrl, iter, parity);
        }
        HIPCHK(h, hipGetLastError());
        return DLSM_OK;
    }
    algo = resolve_sweep_algo(h, algo);
    if constexpr (DD <= DLSM_D_PIPE_MAX) {
        if (algo == 4) return launch_sweep_pipe<DD>(h, iter, alloc_only);
    }
    if (algo == 2) return launch_sweep_spec<DD>(h, iter, 1, alloc_only);
    if (algo == 3)
        return launch_sweep_spec<DD>(h, iter, getenv("DLSM_SPEC_S") ? atoi(getenv("DLSM_SPEC_S")) : 2,
                                     alloc_only);
    const size_t lds = sweep_slice_lds_bytes(h->N, DD, h->W, h->model);
    if (lds > 160 * 1024)
        FAIL(h, DLSM_E_LIMIT, "N=%d needs %zu B of LDS in the slice sweep (max 163840)",
             h->N, lds);
    {
        auto ku = k_sweep_slice<DD, DLSM_UNDIRECTED>;
        auto kd = k_sweep_slice<DD, DLSM_DIRECTED>;
        HIPCHK(h, hipFuncSetAttribute((const void *)ku,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(h, hipFuncSetAttribute((const void *)kd,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (alloc_only) return DLSM_OK;
    for (int parity = 0; parity < 2; ++parity) {
        int nsl = (h->T - parity + 1) / 2;
        if (nsl <= 0) continue;
        if (h->model == DLSM_UNDIRECTED)
            hipLaunchKernelGGL((k_sweep_slice<DD, DLSM_UNDIRECTED>), dim3(nsl), dim3(SW_THREADS),
                               lds, h->stream, v, iter, parity);
        else
            hipLaunchKernelGGL((k_sweep_slice<DD, DLSM_DIRECTED>), dim3(nsl), dim3(SW_THREADS),
                               lds, h->stream, v, iter, parity);
    }
    HIPCHK(h, hipGetLastError());
    return DLSM_OK;
}

// the sweep algorithms a caller may name (dlsm_sweep_positions, the loops' configurations)
static int check_sweep_algo(dlsm_chain *h, int algo) {
    NEED(h, algo >= 0 && algo <= 5, "algo must be 0..5 (6 and 7, measured slower than 4, were removed in round 5)");
    NEED(h, algo != 5 || h->model == DLSM_DIRECTED_CASE_CONTROL, "algo 5 is the case-control sweep");
    NEED(h, algo != 4 || h->D <= DLSM_D_PIPE_MAX, "the pipelined sweep (algo 4) is built for n_features <= 8");
    NEED(h, algo != 5 || h->D <= DLSM_D_CCPIPE_MAX,
         "the sparse case-control sweep (algo 5) is not built for this n_features");
    return DLSM_OK;
}

static int check_ready_sweep(dlsm_chain *h) {
    int rc = check_ready_loglik(h); if (rc) return rc;
    NEED(h, h->have_samplers, "samplers not set");
    NEED(h, h->have_prior, "prior not set");
    return DLSM_OK;
}

static int enqueue_sweep(dlsm_chain *h, IterRef iter, int algo, bool alloc_only = false) {
    int rc = DLSM_OK;
    if (!alloc_only) h->pipe_touched = false;
    DISPATCH_D(h, h->D, rc = launch_sweep<DD>(h, iter, algo, alloc_only));
    if (!alloc_only) {              // only a pipelined sweep leaves proposal buffers a tail can fill
        if (!h->pipe_touched) h->next_prop_ok = false;
        h->prop_drawn_for = -1;
        if (h->sweep_part == 0) h->head_done_for = -1;
    }
    return rc;
}

extern "C" {

int dlsm_resolve_sweep_algo(dlsm_chain *h, int algo) {
    NEED(h, h != nullptr, "null handle");
    int rc = check_sweep_algo(h, algo); if (rc) return rc;
    return resolve_sweep_algo(h, algo);
}

int dlsm_sweep_positions(dlsm_chain *h, uint32_t iter, int algo) {
    NEED(h, h != nullptr, "null handle");
    int rc = check_sweep_algo(h, algo); if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    rc = check_ready_sweep(h); if (rc) return rc;
    h->prop_drawn_for = -1;
    rc = enqueue_sweep(h, IterRef{iter, nullptr}, algo); if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return check_pipe_err(h);
}

}  // extern "C"

template <int DD>
static int launch_post(dlsm_chain *h, const double *d_xref, int n_iter_procrustes,
                       int do_center, LsmDeviceState *lsm, IterRef iter, double *d_R,
                       bool alloc_only = false, double *trace_X = nullptr, double *xr = nullptr,
                       int ride_nwg = 0, int ride_jl = -1, int ride_par = 0) {
    ChainView v = h->view();
    const long rows = (long)h->T * h->N;
    const int nb = (int)std::min<long>(PS_BLOCKS, (rows + PS2_THREADS - 1) / PS2_THREADS);
    constexpr int W = PostRec<DD>::W;
    // the records live behind the log-likelihood records (both are consumed before
    // the next producer runs: same stream)
    int rc = ensure_partials(h, (size_t)ll_blocks(h) * 4 + (size_t)PS_BLOCKS * W);
    if (rc) return rc;
    if (alloc_only) return DLSM_OK;
    double *rec = h->partials + (size_t)ll_blocks(h) * 4;
    ProfScope ps(h, DLSM_K_CENTER);
    if (ride_nwg > 0) {     // the sums rode in the sweep's last launch, but for the rows it was still moving
        hipLaunchKernelGGL((k_post_apply<DD>), dim3(nb), dim3(PS2_THREADS), 0, h->stream, v,
                           d_xref ? 1 : 0, n_iter_procrustes, do_center, rec, ride_nwg, lsm, iter, d_R,
                           trace_X, xr, ride_jl, ride_par, d_xref);
        HIPCHK(h, hipGetLastError());
        return DLSM_OK;
    }
    hipLaunchKernelGGL((k_post_reduce<DD>), dim3(nb), dim3(PS2_THREADS), 0, h->stream, v,
                       d_xref, n_iter_procrustes, iter, rec);
    hipLaunchKernelGGL((k_post_apply<DD>), dim3(nb), dim3(PS2_THREADS), 0, h->stream, v,
                       d_xref ? 1 : 0, n_iter_procrustes, do_center, rec, nb, lsm, iter, d_R,
                       trace_X, xr);
    HIPCHK(h, hipGetLastError());
    return DLSM_OK;
}

namespace {
// The label block update's two launches.  LM_NODES nodes per workgroup on the f64 matrix cores when
// the transition matrices and the nodes' tables fit in LDS together (config 3: 58 KB); the
// wavefront-per-node kernel otherwise (DLSM_LABELS_KERNEL=wave forces it, for measurements).
static bool labels_wave_forced() {
    static const bool v = [] { const char *e = getenv("DLSM_LABELS_KERNEL"); return e && !strcmp(e, "wave"); }();
    return v;
}

template <int KS>
int launch_labels_mfma(dlsm_chain *h, const ChainView &v, uint32_t iter, hipStream_t q, int32_t *flag,
                       int32_t flag_val) {
    auto kern = k_sample_labels_mfma<KS>;
    const size_t lds = lm_lds_bytes(h->T, h->K, h->D);
    static size_t armed = 0;                    // per instantiation: the largest size asked for
    if (lds > armed) {
        HIPCHK(h, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));
        armed = lds;
    }
    hipLaunchKernelGGL(kern, dim3((h->N + LM_NODES - 1) / LM_NODES), dim3(LM_THREADS), lds, q, v, h->lab_w, iter,
                       h->z, flag, flag_val);
    return DLSM_OK;
}

// the label update goes to the matrix-core kernel (and can carry the HDP loop's hand-over flag)
static bool labels_mfma_path(const dlsm_chain *h) {
    return h->K <= 4 * LM_MAX_KS && lm_lds_bytes(h->T, h->K, h->D) <= 150 * 1024 && !labels_wave_forced();
}

template <int DD>
int launch_sample_labels(dlsm_chain *h, const ChainView &v, uint32_t iter, uint8_t *trace_row,
                         hipStream_t q, int32_t *flag = nullptr, int32_t flag_val = 0, bool counts = true) {
    const int T = h->T, K = h->K, N = h->N;
    // The matrix-core kernel up to 32 components (KS = K / 4 <= 8 unrolled steps: no scratch memory;
    // above that its unrolled tiles spill - 44 to 303 scratch instructions at KS = 9 .. 16 - and the
    // wavefront-per-node kernel, which has no such limit, takes over)
    if (labels_mfma_path(h)) {
        int rc;
        switch (lm_ksteps(K)) {
#define DLSM_LM_CASE(KS_) case KS_: rc = launch_labels_mfma<KS_>(h, v, iter, q, flag, flag_val); break;
            DLSM_LM_CASE(1) DLSM_LM_CASE(2) DLSM_LM_CASE(3) DLSM_LM_CASE(4) DLSM_LM_CASE(5)
            DLSM_LM_CASE(6) DLSM_LM_CASE(7)
            default: rc = launch_labels_mfma<LM_MAX_KS>(h, v, iter, q, flag, flag_val); break;
#undef DLSM_LM_CASE
        }
        if (rc) return rc;
    } else {
        const size_t lds_tables = (size_t)LAB_WAVES * 2 * T * K * sizeof(double);
        const size_t lds_w = (size_t)T * K * lab_row_pad(K) * sizeof(double);
        if (lds_tables > 160 * 1024)
            FAIL(h, DLSM_E_LIMIT, "T*K=%d too large for the label kernel", T * K);
        const bool w_lds = lds_tables + lds_w <= 80 * 1024;     // two workgroups per CU
        const size_t lds = lds_tables + (w_lds ? lds_w : 0);
        auto kern = w_lds ? k_sample_labels<DD, true> : k_sample_labels<DD, false>;
        if (lds > 64 * 1024)
            HIPCHK(h, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)lds));
        hipLaunchKernelGGL(kern, dim3((N + LAB_WAVES - 1) / LAB_WAVES), dim3(64 * LAB_WAVES), lds, q, v,
                           h->lab_w, iter, h->z);
    }
    // (the HDP-LPCM loop counts inside its next launch: k_hdp_stage1's counts + tables role)
    if (counts)
        hipLaunchKernelGGL(k_label_counts, dim3(T), dim3(LC_THREADS), (size_t)(K * K + K) * sizeof(int32_t), q,
                           h->z, N, K, (int32_t *)h->lab_n, (int32_t *)h->lab_nk, trace_row);
    return DLSM_OK;
}

}  // namespace

extern "C" {

int dlsm_center(dlsm_chain *h) {
    NEED(h, h != nullptr, "null handle");
    NEED(h, h->have_X, "latent positions not set");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = DLSM_OK;
    DISPATCH_D(h, h->D, rc = launch_post<DD>(h, nullptr, -1, 1, nullptr, IterRef{0, nullptr}, nullptr));
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DLSM_OK;
}

int dlsm_procrustes(dlsm_chain *h, const double *X_ref, double *R_out) {
    NEED(h, h && X_ref, "null argument");
    NEED(h, h->have_X, "latent positions not set");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = h2d(h, h->xref, X_ref, (size_t)h->T * h->N * h->D); if (rc) return rc;
    DISPATCH_D(h, h->D, rc = launch_post<DD>(h, h->xref, -1, 0, nullptr, IterRef{0, nullptr}, h->dsmall + 64));
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (R_out) return d2h(h, R_out, h->dsmall + 64, (size_t)h->D * h->D);
    return DLSM_OK;
}

// ---------------------------------------------------------------- labels
int dlsm_gaussian_likelihood(dlsm_chain *h, int node, int normalize, double *out) {
    NEED(h, h && out, "null argument");
    NEED(h, node >= 0 && node < h->N, "node out of range");
    NEED(h, h->have_X && h->prior_kind == DLSM_PRIOR_MIXTURE && h->have_prior,
         "needs positions and the mixture prior");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t n = (size_t)h->T * h->K;
    int rc = ensure_partials(h, n); if (rc) return rc;
    ChainView v = h->view();
    DISPATCH_D(h, h->D, hipLaunchKernelGGL((k_gauss_table<DD>), dim3(1), dim3(64), 0,
                                           h->stream, v, node, normalize, h->partials));
    HIPCHK(h, hipGetLastError());
    return d2h(h, out, h->partials, n);
}

int dlsm_sample_labels(dlsm_chain *h, uint32_t iter, const double *w, int64_t *z,
                       double *n, int64_t *nk) {
    NEED(h, h && w && z && n && nk, "null argument");
    NEED(h, h->have_X && h->prior_kind == DLSM_PRIOR_MIXTURE && h->have_prior,
         "needs positions and the mixture prior");
    HIPCHK(h, hipSetDevice(h->device));
    const int T = h->T, K = h->K, N = h->N;
    const size_t nn = (size_t)T * K * K, nnk = (size_t)T * K;
    { int rc = ensure_label_bufs(h); if (rc) return rc; }
    size_t staged = 0;
    { int rc = h2d_enqueue(h, h->lab_w, w, nn, &staged); if (rc) return rc; }
    ChainView v = h->view();
    {
        ProfScope ps(h, DLSM_K_LABELS);
        DISPATCH_D(h, h->D, {
            int rc = launch_sample_labels<DD>(h, v, iter, nullptr, h->stream); if (rc) return rc;
        });
    }
    HIPCHK(h, hipGetLastError());
    // labels and counts come back as int32 in one batch: into the pinned buffer behind one
    // synchronisation when they fit, converted from there
    const size_t ntn = (size_t)T * N;
    const size_t b_z = ntn * sizeof(int32_t), b_n = nn * sizeof(int32_t), b_nk = nnk * sizeof(int32_t);
    const size_t o_n = (b_z + 255) & ~(size_t)255, o_nk = o_n + ((b_n + 255) & ~(size_t)255);
    if (h->stage && o_nk + b_nk <= STAGE_BYTES) {
        char *st = (char *)h->stage;
        HIPCHK(h, hipMemcpyAsync(st, h->z, b_z, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(st + o_n, h->lab_n, b_n, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(st + o_nk, h->lab_nk, b_nk, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        const int32_t *zz = (const int32_t *)st, *cn = (const int32_t *)(st + o_n),
                      *cnk = (const int32_t *)(st + o_nk);
        for (size_t i = 0; i < ntn; ++i) z[i] = zz[i];
        for (size_t i = 0; i < nn; ++i) n[i] = (double)cn[i];
        for (size_t i = 0; i < nnk; ++i) nk[i] = cnk[i];
        return DLSM_OK;
    }
    std::vector<int32_t> zz(ntn), cn(nn), cnk(nnk);
    int rc = d2h(h, zz.data(), h->z, zz.size()); if (rc) return rc;
    rc = d2h(h, cn.data(), h->lab_n, nn); if (rc) return rc;
    rc = d2h(h, cnk.data(), h->lab_nk, nnk); if (rc) return rc;
    for (size_t i = 0; i < zz.size(); ++i) z[i] = zz[i];
    for (size_t i = 0; i < nn; ++i) n[i] = (double)cn[i];
    for (size_t i = 0; i < nnk; ++i) nk[i] = cnk[i];
    return DLSM_OK;
}

// ---------------------------------------------------------------- LSM chain
int dlsm_lsm_configure(dlsm_chain *h, const dlsm_lsm_config *cfg) {
    NEED(h, h && cfg, "null argument");
    drop_graph(h);
    NEED(h, cfg->intercept_variance_prior > 0, "intercept_variance_prior must be positive");
    NEED(h, cfg->i_tune_interval > 0, "tune_interval must be positive");
    { int rc_ = check_sweep_algo(h, cfg->sweep_algo); if (rc_) return rc_; }
    HIPCHK(h, hipSetDevice(h->device));
    LsmDeviceState s;
    memset(&s, 0, sizeof(s));
    for (int k = 0; k < 2; ++k) {
        s.intercept_prior[k] = cfg->intercept_prior[k];
        s.i_step[k] = cfg->i_step_size[k];
        s.i_nacc[k] = cfg->i_n_accepted[k];
        s.i_nsteps[k] = cfg->i_n_steps[k];
        s.i_until[k] = cfg->i_steps_until_tune[k];
    }
    s.intercept_var = cfg->intercept_variance_prior;
    s.i_tune = cfg->i_tune < 0 ? -1 : cfg->i_tune;
    s.i_tune_interval = cfg->i_tune_interval;
    s.r_step = cfg->r_step_size;
    s.r_nacc = cfg->r_n_accepted; s.r_nsteps = cfg->r_n_steps;
    s.r_until = cfg->r_steps_until_tune;
    s.r_tune = cfg->r_tune < 0 ? -1 : cfg->r_tune;
    s.r_tune_interval = cfg->r_tune_interval > 0 ? cfg->r_tune_interval : 100;
    HIPCHK(h, hipMemcpy(h->lsm, &s, sizeof(s), hipMemcpyHostToDevice));
    h->lsm_cfg = *cfg;
    h->lsm_configured = true;
    return DLSM_OK;
}

int dlsm_lsm_get_config(dlsm_chain *h, dlsm_lsm_config *cfg) {
    NEED(h, h && cfg, "null argument");
    NEED(h, h->lsm_configured, "LSM chain not configured");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    LsmDeviceState s;
    HIPCHK(h, hipMemcpy(&s, h->lsm, sizeof(s), hipMemcpyDeviceToHost));
    *cfg = h->lsm_cfg;
    for (int k = 0; k < 2; ++k) {
        cfg->i_step_size[k] = s.i_step[k];
        cfg->i_n_accepted[k] = s.i_nacc[k];
        cfg->i_n_steps[k] = s.i_nsteps[k];
        cfg->i_steps_until_tune[k] = s.i_until[k];
    }
    cfg->r_step_size = s.r_step;
    cfg->r_n_accepted = s.r_nacc; cfg->r_n_steps = s.r_nsteps;
    cfg->r_steps_until_tune = s.r_until;
    return DLSM_OK;
}

int dlsm_trace_alloc(dlsm_chain *h, int n_total, double logp0) {
    NEED(h, h && n_total >= 1, "bad argument");
    drop_graph(h);
    NEED(h, h->have_X, "latent positions not set");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t row = (size_t)h->T * h->N * h->D;
    void *old[] = {h->trace_X, h->trace_ic, h->trace_logp, h->trace_radii};
    for (void *p : old) if (p) hipFree(p);
    h->trace_X = h->trace_ic = h->trace_logp = h->trace_radii = nullptr;
    int rc = dev_alloc(h, &h->trace_X, row * n_total); if (rc) return rc;
    rc = dev_alloc(h, &h->trace_ic, (size_t)2 * n_total); if (rc) return rc;
    rc = dev_alloc(h, &h->trace_logp, n_total); if (rc) return rc;
    if (h->model != DLSM_UNDIRECTED) {
        NEED(h, h->have_radii, "radii not set");
        rc = dev_alloc(h, &h->trace_radii, (size_t)h->N * n_total); if (rc) return rc;
        HIPCHK(h, hipMemcpyAsync(h->trace_radii, h->radii, h->N * sizeof(double),
                                 hipMemcpyDeviceToDevice, h->stream));
    }
    h->trace_n = n_total;
    HIPCHK(h, hipMemsetAsync(h->trace_ic, 0, sizeof(double) * 2 * n_total, h->stream));
    HIPCHK(h, hipMemsetAsync(h->trace_logp, 0, sizeof(double) * n_total, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->trace_X, h->X, row * sizeof(double),
                             hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->trace_ic, h->intercept, 2 * sizeof(double),
                             hipMemcpyDeviceToDevice, h->stream));
    h->hsmall[0] = logp0;
    HIPCHK(h, hipMemcpyAsync(h->trace_logp, h->hsmall, sizeof(double),
                             hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DLSM_OK;
}

// One Gibbs iteration of the undirected LSM on the handle's stream (lsm.py:474-572).
// With `counter` the iteration index is read from device memory (captured graph: the
// first kernel advances it), otherwise it is the value `it`.
static int enqueue_lsm_iteration(dlsm_chain *h, int it, bool counter, int procrustes_ref,
                                 bool alloc_only = false, bool draw_next = false) {
    const size_t row = (size_t)h->T * h->N * h->D;
    const IterRef ir{(uint32_t)it, counter ? &h->lsm->iter : nullptr};
    int rc = DLSM_OK;
    if (counter && !alloc_only)
        hipLaunchKernelGGL(k_advance_iter, dim3(1), dim3(1), 0, h->stream, &h->lsm->iter);
    const double *xref = procrustes_ref >= 0 ? h->trace_X + row * procrustes_ref : nullptr;
    // undirected loop: the centring sums ride in the pipelined sweep's last launch when there is one
    // (DLSM_POST_RIDE=0: a launch of their own); whether the rotation is on is decided here
    h->post_ride_want = h->model == DLSM_UNDIRECTED && !counter && !alloc_only &&
                        !(getenv("DLSM_POST_RIDE") && atoi(getenv("DLSM_POST_RIDE")) == 0);
    h->post_ride_done = false;
    if (h->post_ride_want) {
        rc = ensure_partials(h, (size_t)ll_blocks(h) * 4 + (size_t)PS_BLOCKS * POST_W_MAX); if (rc) return rc;
        const int nip = h->lsm_cfg.n_iter_procrustes;
        h->post_ride_xref = (xref && (nip < 0 || it > nip)) ? xref : nullptr;
    }
    h->loop_draws_intercept = h->post_ride_want;     // the proposal pass also draws the intercept proposal
    rc = enqueue_sweep(h, ir, h->lsm_cfg.sweep_algo, alloc_only);
    h->post_ride_want = false; h->loop_draws_intercept = false;
    if (rc) return rc;
    const bool rode = h->post_ride_done;
    h->post_ride_done = false;
    if (rode && draw_next && h->next_prop_ok && h->next_prop.lsm_draw &&
        !(getenv("DLSM_POST_FUSE") && atoi(getenv("DLSM_POST_FUSE")) == 0)) {
        // The likelihood pass on the positions as the sweep left them (distances do not change under
        // the centring pass's rotation and shift; its intercept proposal came with the sweep's
        // proposals), then ONE launch for the rest of the iteration: centring, accept / reject, trace
        // row, the next sweep's proposal pass (k_lsm_finalize_apply_propose).
        int nrec_f = 0;
        rc = loglik_records(h, 2, h->lsm->cand, nullptr, nullptr, &nrec_f); if (rc) return rc;
        ProfScope ps(h, DLSM_K_FINALIZE);
        ChainView v = h->view();
        const PostFusedArgs pa{xref ? 1 : 0, h->lsm_cfg.n_iter_procrustes,
                               h->partials + (size_t)ll_blocks(h) * 4, h->post_ride_nwg, h->post_ride_jl,
                               h->post_ride_par, xref, h->trace_X};
        DISPATCH_D(h, h->D, hipLaunchKernelGGL((k_lsm_finalize_apply_propose<DD>),
                                               dim3(1 + propose_blocks(h->T, h->N)), dim3(256), 0, h->stream,
                                               h->partials, nrec_f, h->lsm, h->intercept, h->trace_ic,
                                               h->trace_logp, ir, v, h->next_prop, pa));
        h->prop_drawn_for = (long)it + 1;
        HIPCHK(h, hipGetLastError());
        return DLSM_OK;
    }
    // case-control: the centring pass also writes the log-likelihood's gather records
    const bool pf = cc_prefetch_form(h);
    if (pf) { DISPATCH_D(h, h->D, rc = ensure_xr<DD>(h)); if (rc) return rc; }
    if (h->model == DLSM_UNDIRECTED) {
        DISPATCH_D(h, h->D, rc = launch_post<DD>(h, xref, h->lsm_cfg.n_iter_procrustes, 1, h->lsm,
                                                 ir, nullptr, alloc_only, h->trace_X, nullptr,
                                                 rode ? h->post_ride_nwg : 0, h->post_ride_jl, h->post_ride_par));
        if (rc) return rc;
    }
    if (alloc_only)     // (the directed loops' records: likelihood | centring | radii proposal)
        return ensure_partials(h, (size_t)ll_blocks(h) * 4 + (size_t)PS_BLOCKS * POST_W_MAX +
                                  (size_t)((h->N + DP_THREADS - 1) / DP_THREADS) * (1 + DP_COLS));
    int nrec = 0;
    if (h->model != DLSM_UNDIRECTED) {
        // intercept_in, intercept_out, radii: propose -> fused two-candidate pass -> accept
        ChainView v = h->view();
        double *ll2 = h->dsmall + 16;
        // The sweep moved the positions, so the first step evaluates proposal and current
        // state; after it the current state's log-likelihood is carried (lsm->ll_cur) and the
        // later steps evaluate their proposal only: 4 candidate evaluations, not 6.
        // Launches besides the three passes: 5, none of them the radii proposal's own (round 2:
        // eleven).  The centring launches draw the first intercept proposal, write the passes'
        // gather records and carry the radii proposal's gamma variates (pass 1) and its
        // normalisation + density terms (pass 2: they file the proposal in the records' second
        // radius slot); the sum of a pass's records, the accept / reject and the next step's
        // proposal share a launch, the first of which carries the proposal's closing workgroup;
        // the last pass's sum, the radii's accept / reject and the trace row share the last
        // launch, which can carry the next sweep's proposal pass.
        const int nblk = (h->N + DP_THREADS - 1) / DP_THREADS;
        constexpr int PW_MAX = POST_W_MAX;
        const size_t n_post = (size_t)PS_BLOCKS * PW_MAX;
        rc = ensure_partials(h, (size_t)ll_blocks(h) * 4 + n_post + (size_t)nblk * (1 + DP_COLS));
        if (rc) return rc;
        double *prec = h->partials + (size_t)ll_blocks(h) * 4;
        double *rrec = prec + n_post, *rrec2 = rrec + nblk;
        double *xr = pf ? h->xr : nullptr;
        const long rows = (long)h->T * h->N;
        const int nbp = (int)std::min<long>(PS_BLOCKS, (rows + PS2_THREADS - 1) / PS2_THREADS);
        {
            ProfScope psc(h, DLSM_K_CENTER);
            const DirRider rg{1, nblk, h->radii, h->radii_alt, rrec, rrec2, xr};
            DISPATCH_D(h, h->D, hipLaunchKernelGGL((k_post_reduce_dir<DD>), dim3(nbp + nblk), dim3(PS2_THREADS), 0,
                                                   h->stream, v, xref, h->lsm_cfg.n_iter_procrustes, ir, prec,
                                                   nbp, h->lsm, rg));
            DISPATCH_D(h, h->D, hipLaunchKernelGGL((k_post_apply_dir<DD>), dim3(nbp + nblk), dim3(PS2_THREADS), 0,
                                                   h->stream, v, xref ? 1 : 0, h->lsm_cfg.n_iter_procrustes, 1,
                                                   prec, nbp, h->lsm, ir, h->trace_X, xr, nbp, rg));
        }
        ProfScope ps(h, DLSM_K_FINALIZE);
        if (pf && !getenv("DLSM_CC_TWO_PASSES")) {
            // case-control: both intercept steps behind ONE four-candidate pass (kernels_dirloop.hpp)
            rc = loglik_records(h, 4, h->lsm->cand8, h->radii, h->radii, &nrec, true);
            if (rc) return rc;
            const DirRider rd{3, nblk, h->radii, h->radii_alt, rrec, rrec2, xr};
            DISPATCH_D(h, h->D, hipLaunchKernelGGL((k_dir_reduce_accept_both<DD>), dim3(2), dim3(256), 0, h->stream,
                                                   h->partials, nrec, ll2, v, h->lsm, h->intercept, ir, rd));
        } else
        for (int which = 0; which < 2; ++which) {
            const int M = which == 0 ? 2 : 1;
            rc = loglik_records(h, M, h->lsm->cand, h->radii, h->radii, &nrec, pf || which == 1);
            if (rc) return rc;
            const DirRider rd{which == 0 ? 3 : 0, nblk, h->radii, h->radii_alt, rrec, rrec2, xr};
            DISPATCH_D(h, h->D, hipLaunchKernelGGL((k_dir_reduce_accept_intercept<DD>),
                                                   dim3(1 + (which == 0 ? 1 : 0)), dim3(256), 0, h->stream,
                                                   h->partials, nrec, M, ll2, v, h->lsm, h->intercept, which,
                                                   which, which == 0 ? 1 : -1, ir, rd));
        }
        // the proposed radii at the current intercepts (one candidate)
        rc = loglik_records(h, 1, h->intercept, h->radii_alt, h->radii_alt, &nrec, pf, 1);
        if (rc) return rc;
        const bool ride = draw_next && !counter && h->next_prop_ok;
        const int grid = 1 + (ride ? (propose_blocks(h->T, h->N) + DR_THREADS / 256 - 1) / (DR_THREADS / 256) : 0);
        DISPATCH_D(h, h->D, hipLaunchKernelGGL((k_dir_tail<DD>), dim3(grid), dim3(DR_THREADS), 0, h->stream,
                                               h->partials, nrec, ll2, v, h->lsm, h->radii, h->radii_alt,
                                               h->intercept, h->trace_ic, h->trace_radii, h->trace_logp,
                                               ir, h->next_prop, ride ? 1 : 0));
        if (ride) h->prop_drawn_for = (long)it + 1;
        HIPCHK(h, hipGetLastError());
        return DLSM_OK;
    }
    rc = loglik_records(h, 2, h->lsm->cand, nullptr, nullptr, &nrec); if (rc) return rc;
    {
        ProfScope ps(h, DLSM_K_FINALIZE);
        if (draw_next && !counter && h->next_prop_ok) {
            // the next sweep's proposal pass rides along (kernels_tail_propose.hpp)
            ChainView v = h->view();
            DISPATCH_D(h, h->D, hipLaunchKernelGGL((k_lsm_finalize_propose<DD>),
                                                   dim3(1 + propose_blocks(h->T, h->N)), dim3(256), 0,
                                                   h->stream, h->partials, nrec, h->lsm, h->intercept,
                                                   h->trace_ic, h->trace_logp, ir, v, h->next_prop));
            h->prop_drawn_for = (long)it + 1;
        } else
            hipLaunchKernelGGL(k_lsm_finalize, dim3(1), dim3(256), 0, h->stream, h->partials, nrec,
                               h->lsm, h->intercept, h->trace_ic, h->trace_logp, ir);
    }
    HIPCHK(h, hipGetLastError());
    return DLSM_OK;
}

int dlsm_lsm_run(dlsm_chain *h, int first, int count, int procrustes_ref) {
    NEED(h, h != nullptr, "null handle");
    NEED(h, h->lsm_configured && h->trace_X, "configure the chain and allocate the trace first");
    NEED(h, h->model == DLSM_UNDIRECTED || h->trace_radii, "no radii trace allocated");
    NEED(h, h->prior_kind == DLSM_PRIOR_RANDOM_WALK, "LSM uses the random-walk prior");
    NEED(h, first >= 1 && count >= 0 && first + count <= h->trace_n, "iteration range out of the trace");
    NEED(h, procrustes_ref < h->trace_n, "procrustes_ref out of the trace");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = check_ready_sweep(h); if (rc) return rc;
    if (count == 0) return DLSM_OK;
    // Replay path (opt-in, DLSM_GRAPH=1): one iteration captured into a hipGraph, so an
    // iteration costs the host one call instead of ~40 launches; kernel arguments are
    // frozen in a graph, hence the device-side iteration counter.  On MI355X the GPU is
    // the limit either way (eager 1874 it/s, replay 1840 it/s at C2), so eager is the
    // default; replay is for hosts that cannot spare a core per chain.
    const bool want_graph = h->model == DLSM_UNDIRECTED && !h->profiling && !h->graph_failed &&
                            count >= 2 &&
                            getenv("DLSM_GRAPH") && atoi(getenv("DLSM_GRAPH")) == 1;
    if (want_graph) {
        if (!h->graph_exec || h->graph_ref != procrustes_ref ||
            h->graph_algo != h->lsm_cfg.sweep_algo) {
            drop_graph(h);
            rc = enqueue_lsm_iteration(h, first, true, procrustes_ref, true);   // allocations
            if (rc) return rc;
            HIPCHK(h, hipStreamSynchronize(h->stream));
            bool ok = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                rc = enqueue_lsm_iteration(h, first, true, procrustes_ref);
                hipError_t e = hipStreamEndCapture(h->stream, &h->graph);
                ok = rc == DLSM_OK && e == hipSuccess && h->graph &&
                     hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0) == hipSuccess;
            }
            if (!ok) {
                (void)hipGetLastError();
                drop_graph(h);
                h->graph_failed = true;       // fall back to eager launches for good
            } else {
                h->graph_ref = procrustes_ref;
                h->graph_algo = h->lsm_cfg.sweep_algo;
            }
        }
        if (h->graph_exec) {
            uint32_t *hp = (uint32_t *)h->hsmall;
            hp[0] = (uint32_t)(first - 1);
            HIPCHK(h, hipMemcpyAsync(&h->lsm->iter, hp, sizeof(uint32_t), hipMemcpyHostToDevice,
                                     h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));   // hsmall is reused by other calls
            for (int i = 0; i < count; ++i) HIPCHK(h, hipGraphLaunch(h->graph_exec, h->stream));
            return DLSM_OK;
        }
    }
    // (read per call: the tests switch it inside one process)
    const bool ride = !(getenv("DLSM_TAIL_PROPOSE") && atoi(getenv("DLSM_TAIL_PROPOSE")) == 0);
    h->prop_drawn_for = -1;
    for (int it = first; it < first + count; ++it) {
        rc = enqueue_lsm_iteration(h, it, false,
                                   it > h->lsm_cfg.n_iter_procrustes ? procrustes_ref : -1, false,
                                   ride && it + 1 < first + count);
        if (rc) return rc;
    }
    h->prop_drawn_for = -1;
    return DLSM_OK;
}

int dlsm_trace_read(dlsm_chain *h, int first, int count, double *Xs, double *intercepts,
                    double *logps) {
    NEED(h, h != nullptr, "null handle");
    NEED(h, h->trace_X, "no trace allocated");
    NEED(h, first >= 0 && count >= 0 && first + count <= h->trace_n, "range out of the trace");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    { int rc_ = check_pipe_err(h); if (rc_) return rc_; }
    const size_t row = (size_t)h->T * h->N * h->D;
    if (Xs) HIPCHK(h, hipMemcpy(Xs, h->trace_X + row * first, row * count * sizeof(double), hipMemcpyDeviceToHost));
    if (intercepts) HIPCHK(h, hipMemcpy(intercepts, h->trace_ic + (size_t)2 * first, sizeof(double) * 2 * count, hipMemcpyDeviceToHost));
    if (logps) HIPCHK(h, hipMemcpy(logps, h->trace_logp + first, sizeof(double) * count, hipMemcpyDeviceToHost));
    return DLSM_OK;
}

int dlsm_hdp_label_sums(dlsm_chain *h, int stage, const double *mu, const double *sigma,
                        double lmbda, const double *w, double a, double b, double *out) {
    NEED(h, h && out, "null argument");
    NEED(h, stage >= 0 && stage <= 3, "stage must be 0..3");
    NEED(h, h->have_X && h->prior_kind == DLSM_PRIOR_MIXTURE && h->have_prior,
         "needs positions and the mixture prior (labels)");
    NEED(h, stage == HDP_SUMS_MEAN || mu, "mu is NULL");
    NEED(h, stage < HDP_SUMS_LAMBDA || sigma, "sigma is NULL");
    NEED(h, stage != HDP_SUMS_LOGP || w, "w is NULL");
    HIPCHK(h, hipSetDevice(h->device));
    const int T = h->T, K = h->K, D = h->D;
    const int nv = stage == HDP_SUMS_MEAN ? D : (stage == HDP_SUMS_LAMBDA ? 2 : 1);
    const size_t n_out = (size_t)T * K * nv;
    const size_t n_mu = (size_t)K * D, n_w = (size_t)T * K * K;
    int rc = ensure_partials(h, n_out + n_mu + K + n_w); if (rc) return rc;
    double *d_out = h->partials, *d_mu = d_out + n_out, *d_sigma = d_mu + n_mu,
           *d_w = d_sigma + K;
    size_t staged = 0;
    if (mu) { rc = h2d_enqueue(h, d_mu, mu, n_mu, &staged); if (rc) return rc; }
    if (sigma) { rc = h2d_enqueue(h, d_sigma, sigma, (size_t)K, &staged); if (rc) return rc; }
    if (stage == HDP_SUMS_LOGP) { rc = h2d_enqueue(h, d_w, w, n_w, &staged); if (rc) return rc; }
    ChainView v = h->view();
    HdpParams hp{d_mu, d_sigma, d_w, lmbda, a, b};
    const dim3 grid(K, T), block(HDP_THREADS);
    {
        ProfScope ps(h, DLSM_K_LABELS);
        switch (stage) {
            case HDP_SUMS_MEAN:
                DISPATCH_D(h, D, hipLaunchKernelGGL((k_hdp_label_sums<DD, HDP_SUMS_MEAN>), grid, block, 0, h->stream, v, hp, d_out));
                break;
            case HDP_SUMS_RESIDUAL:
                DISPATCH_D(h, D, hipLaunchKernelGGL((k_hdp_label_sums<DD, HDP_SUMS_RESIDUAL>), grid, block, 0, h->stream, v, hp, d_out));
                break;
            case HDP_SUMS_LAMBDA:
                DISPATCH_D(h, D, hipLaunchKernelGGL((k_hdp_label_sums<DD, HDP_SUMS_LAMBDA>), grid, block, 0, h->stream, v, hp, d_out));
                break;
            default:
                DISPATCH_D(h, D, hipLaunchKernelGGL((k_hdp_label_sums<DD, HDP_SUMS_LOGP>), grid, block, 0, h->stream, v, hp, d_out));
        }
    }
    HIPCHK(h, hipGetLastError());
    return d2h(h, out, d_out, n_out);
}

int dlsm_trace_read_radii(dlsm_chain *h, int first, int count, double *radii) {
    NEED(h, h && radii, "null argument");
    NEED(h, h->trace_radii, "no radii trace (directed models, after dlsm_trace_alloc)");
    NEED(h, first >= 0 && count >= 0 && first + count <= h->trace_n, "range out of the trace");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(radii, h->trace_radii + (size_t)h->N * first,
                        sizeof(double) * h->N * count, hipMemcpyDeviceToHost));
    return DLSM_OK;
}

// ---------------------------------------------------------------- measurement
int dlsm_profile_enable(dlsm_chain *h, int on) {
    NEED(h, h != nullptr, "null handle");
    drop_graph(h);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    drain_profile(h);
    if (on) {
        // in-kernel wall-clock stamps of the sweep's eval launches: [start, end] per
        // workgroup (plain stores, no atomics: nothing contends)
        const size_t cap = (size_t)2 << 20;                     // pairs (32 MB)
        if (!h->stamps) {
            HIPCHK(h, hipMalloc((void **)&h->stamps, cap * 2 * sizeof(unsigned long long)));
            h->stamps_cap = cap;
        }
        HIPCHK(h, hipMemset(h->stamps, 0, h->stamps_cap * 2 * sizeof(unsigned long long)));
        h->stamps_used = 0;
        h->stamp_launches.clear();
    }
    if (on) for (int k = 0; k < DLSM_K_COUNT; ++k) { h->prof[k].ms = 0; h->prof[k].launches = 0; }
    h->profiling = on != 0;
    return DLSM_OK;
}

int dlsm_profile_read(dlsm_chain *h, int kernel, double *total_ms, int *launches) {
    NEED(h, h && total_ms && launches, "null argument");
    NEED(h, kernel >= 0 && kernel < DLSM_K_COUNT, "bad kernel id");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    drain_profile(h);
    *total_ms = h->prof[kernel].ms;
    *launches = h->prof[kernel].launches;
    return DLSM_OK;
}

int dlsm_profile_read_eval_stamps(dlsm_chain *h, double *mean_us, int *launches) {
    NEED(h, h && mean_us && launches, "null argument");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *mean_us = 0.0; *launches = 0;
    if (!h->stamps || h->stamps_used == 0) return DLSM_OK;
    std::vector<unsigned long long> st(h->stamps_used * 2);
    HIPCHK(h, hipMemcpy(st.data(), h->stamps, st.size() * sizeof(unsigned long long),
                        hipMemcpyDeviceToHost));
    double tot = 0.0; int n = 0;
    for (auto &L : h->stamp_launches) {
        unsigned long long lo = ~0ull, hi = 0ull;
        for (size_t b = L.first; b < L.first + L.second; ++b) {
            if (st[2 * b] && st[2 * b] < lo) lo = st[2 * b];
            if (st[2 * b + 1] > hi) hi = st[2 * b + 1];
        }
        if (hi > lo) { tot += (double)(hi - lo) * 0.01; ++n; }
    }
    if (n) *mean_us = tot / n;
    *launches = n;
    return DLSM_OK;
}

int dlsm_timer_start(dlsm_chain *h) {
    NEED(h, h != nullptr, "null handle");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipEventRecord(h->timer0, h->stream));
    return DLSM_OK;
}

int dlsm_timer_stop(dlsm_chain *h, double *ms) {
    NEED(h, h && ms, "null argument");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipEventRecord(h->timer1, h->stream));
    HIPCHK(h, hipEventSynchronize(h->timer1));
    float f = 0.f;
    HIPCHK(h, hipEventElapsedTime(&f, h->timer0, h->timer1));
    *ms = f;
    return DLSM_OK;
}

}  // extern "C"

#include "capi_init.hpp"
#include "capi_post.hpp"
#include "capi_forecast.hpp"
#include "capi_hdp.hpp"

extern "C" int dlsm_host_sample_tables(void *numpy_bitgen, int T, int K, const double *n,
                                       const double *beta, double alpha_init, double alpha,
                                       double kappa, int64_t *m) {
    if (!numpy_bitgen || !n || !beta || !m || T < 1 || K < 1) return -1;
    return dlsm::host_sample_tables((dlsm::NumpyBitGen *)numpy_bitgen, T, K, n, beta, alpha_init,
                                    alpha, kappa, m);
}

#ifdef DLSM_PIPE_TIMING
extern "C" int dlsm_debug_pipe_timing(unsigned long long *items, unsigned long long *res) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(items, HIP_SYMBOL(dlsm::g_pipe_item_t), sizeof(dlsm::g_pipe_item_t)) != hipSuccess) return -2;
    if (hipMemcpyFromSymbol(res, HIP_SYMBOL(dlsm::g_pipe_res_t), sizeof(dlsm::g_pipe_res_t)) != hipSuccess) return -3;
    return 0;
}
extern "C" int dlsm_debug_ccpipe_timing(unsigned long long *res, unsigned long long *items) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(res, HIP_SYMBOL(dlsm::g_cc_res_t), sizeof(dlsm::g_cc_res_t)) != hipSuccess) return -2;
    if (hipMemcpyFromSymbol(items, HIP_SYMBOL(dlsm::g_cc_item_t), sizeof(dlsm::g_cc_item_t)) != hipSuccess) return -3;
    return 0;
}
extern "C" int dlsm_debug_labels_timing(unsigned long long *waves) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(waves, HIP_SYMBOL(dlsm::g_lab_t), sizeof(dlsm::g_lab_t)) != hipSuccess) return -2;
    return 0;
}
extern "C" int dlsm_debug_hdp_globals_phases(unsigned long long *out) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dlsm::g_hdp_phase), sizeof(dlsm::g_hdp_phase)) != hipSuccess) return -2;
    return 0;
}
extern "C" int dlsm_debug_hdp_tail_timing(unsigned long long *out) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dlsm::g_hdp_t), sizeof(dlsm::g_hdp_t)) != hipSuccess) return -2;
    return 0;
}
extern "C" int dlsm_debug_loglik_timing(unsigned long long *waves) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(waves, HIP_SYMBOL(dlsm::g_ll_t), sizeof(dlsm::g_ll_t)) != hipSuccess) return -2;
    return 0;
}
#endif
