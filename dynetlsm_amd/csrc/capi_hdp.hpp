// C-ABI of the device-resident HDP-LPCM loop (included by capi.hip; the FAIL / HIPCHK / NEED
// macros, the sweep / post / log-likelihood launchers and the copy helpers come from there).
#pragma once

namespace {

// carve the loop's auxiliary buffers out of one allocation
HdpLoopBuf hdp_loop_buf(dlsm_chain *h) {
    const size_t T = h->T, K = h->K, D = h->D;
    HdpLoopBuf b;
    double *p = h->hdp_buf;
    b.beta = p; p += K;
    b.mbar = p; p += K;
    b.S = p; p += T * K * D;
    b.Q = p; p += T * K;
    b.L = p; p += 2 * T * K;
    b.LP = p; p += T * K;
    b.LPD = p; p += T * K;
    b.scr = p; p += HS_COUNT;
    b.m = (int32_t *)p;
    b.wover = b.m + T * K * K;
    b.w = h->lab_w; b.n = h->lab_n; b.nk = h->lab_nk;
    b.mu = h->mu; b.sigma = h->sigma;
    b.K = (int)K;
    return b;
}

size_t hdp_loop_buf_doubles(const dlsm_chain *h) {
    const size_t T = h->T, K = h->K, D = h->D;
    return 2 * K + T * K * D + 5 * T * K + HS_COUNT + (T * K * K + T * K + 1) / 2 + 2;
}

void hdp_free_trace(dlsm_chain *h) {
    void *ptrs[] = {h->htr_mu, h->htr_sigma, h->htr_beta, h->htr_w, h->htr_lambda, h->htr_hyper,
                    h->htr_z};
    for (void *p : ptrs) if (p) hipFree(p);
    h->htr_mu = h->htr_sigma = h->htr_beta = h->htr_w = h->htr_lambda = h->htr_hyper = nullptr;
    h->htr_z = nullptr; h->htr_n = 0; h->htr_K = 0;
}

// polls of ~1 us of the waits INSIDE kernels (kernels_hdploop.hpp, HdpFork); DLSM_HDP_FORK_BUDGET overrides it
static int hdp_fork_budget() {
    const char *e = getenv("DLSM_HDP_FORK_BUDGET");
    return e ? atoi(e) : (1 << 22);
}

// "queue `s` goes on when flags[which] has reached the ticket": a wait of the QUEUE itself
// (hipStreamWaitValue32: the command processor polls the word, no wavefront does and no budget runs out -
// round-4 advice: a gate kernel enqueued while the chain's queue still held seconds of earlier work used up
// its polls and turned a slow but correct run into an error) where the device offers it, else the
// one-wavefront gate kernel with its poll budget.  DLSM_HDP_GATE=kernel forces the latter.
static int hdp_queue_wait(dlsm_chain *h, hipStream_t s, const HdpFork &fk, int which) {
    if (h->fork_wait_value) {
        HIPCHK(h, hipStreamWaitValue32(s, h->fork_flags + which, (uint32_t)fk.ticket, hipStreamWaitValueGte,
                                       0xFFFFFFFFu));
        return DLSM_OK;
    }
    hipLaunchKernelGGL(k_hdp_gate, dim3(1), dim3(64), 0, s, fk, which);
    return DLSM_OK;
}

template <int DD>
int enqueue_hdp_iteration(dlsm_chain *h, int it, bool draw_next) {
    const IterRef ir{(uint32_t)it, nullptr};
    const int T = h->T, K = h->K, N = h->N;
    // (the centring sums ride in the pipelined sweep's last launch when there is one)
    h->post_ride_want = h->model == DLSM_UNDIRECTED &&
                        !(getenv("DLSM_POST_RIDE") && atoi(getenv("DLSM_POST_RIDE")) == 0);
    h->post_ride_done = false; h->post_ride_xref = nullptr;
    int rc = DLSM_OK;
    if (h->post_ride_want) { rc = ensure_partials(h, (size_t)ll_blocks(h) * 4 + (size_t)PS_BLOCKS * POST_W_MAX); if (rc) return rc; }
    rc = enqueue_sweep(h, ir, h->hdp_cfg.sweep_algo);
    h->post_ride_want = false;
    if (rc) return rc;
    const bool rode = h->post_ride_done;
    h->post_ride_done = false;
    const bool directed = h->model != DLSM_UNDIRECTED;
    if (directed) {
        // hdp_lpcm.py:855-874 with is_directed: centring, intercept_in, intercept_out and the radii
        // step around three likelihood passes - the launches of dlsm_lsm_run's directed loop
        // (enqueue_lsm_iteration, kernels_dirloop.hpp), without a Procrustes reference and with the
        // last launch leaving the network log-likelihood of the stored state in the row's
        // log-posterior slot (the batched pass behind the run turns it into the log-posterior)
        const bool pf = cc_prefetch_form(h);
        if (pf) { rc = ensure_xr<DD>(h); if (rc) return rc; }
        ChainView vd = h->view();
        double *ll2 = h->dsmall + 16;
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
            hipLaunchKernelGGL((k_post_reduce_dir<DD>), dim3(nbp + nblk), dim3(PS2_THREADS), 0, h->stream, vd,
                               (const double *)nullptr, 0, ir, prec, nbp, h->lsm, rg);
            hipLaunchKernelGGL((k_post_apply_dir<DD>), dim3(nbp + nblk), dim3(PS2_THREADS), 0, h->stream, vd, 0, 0,
                               1, prec, nbp, h->lsm, ir, h->trace_X, xr, nbp, rg);
        }
        {
            ProfScope psf(h, DLSM_K_FINALIZE);
            int nrec_d = 0;
            if (pf && !getenv("DLSM_CC_TWO_PASSES")) {      // (as the directed LSM loop: capi.hip)
                rc = loglik_records(h, 4, h->lsm->cand8, h->radii, h->radii, &nrec_d, true);
                if (rc) return rc;
                const DirRider rd{3, nblk, h->radii, h->radii_alt, rrec, rrec2, xr};
                hipLaunchKernelGGL((k_dir_reduce_accept_both<DD>), dim3(2), dim3(256), 0, h->stream, h->partials,
                                   nrec_d, ll2, vd, h->lsm, h->intercept, ir, rd);
            } else
            for (int which = 0; which < 2; ++which) {
                const int M = which == 0 ? 2 : 1;
                rc = loglik_records(h, M, h->lsm->cand, h->radii, h->radii, &nrec_d, pf || which == 1);
                if (rc) return rc;
                const DirRider rd{which == 0 ? 3 : 0, nblk, h->radii, h->radii_alt, rrec, rrec2, xr};
                hipLaunchKernelGGL((k_dir_reduce_accept_intercept<DD>), dim3(1 + (which == 0 ? 1 : 0)), dim3(256), 0,
                                   h->stream, h->partials, nrec_d, M, ll2, vd, h->lsm, h->intercept, which, which,
                                   which == 0 ? 1 : -1, ir, rd);
            }
            rc = loglik_records(h, 1, h->intercept, h->radii_alt, h->radii_alt, &nrec_d, pf, 1);
            if (rc) return rc;
            hipLaunchKernelGGL((k_dir_tail<DD>), dim3(1), dim3(DR_THREADS), 0, h->stream, h->partials, nrec_d, ll2,
                               vd, h->lsm, h->radii, h->radii_alt, h->intercept, h->trace_ic, h->trace_radii,
                               h->trace_logp, ir, h->next_prop, 0, 1);
        }
        HIPCHK(h, hipGetLastError());
    } else {
    // centring; workgroup 0 draws the intercept proposal; the positions' trace row
    rc = launch_post<DD>(h, nullptr, 0, 1, h->lsm, ir, nullptr, false, h->trace_X, nullptr,
                         rode ? h->post_ride_nwg : 0, h->post_ride_jl, h->post_ride_par);
    if (rc) return rc;
    }
    // The label block update needs the centred positions and last iteration's mixture, not the
    // intercept's likelihood records (31 us at config 3), and those do not need the labels: with
    // `fork` the pass and the intercept step go to a queue of their own beside the label update and
    // the conjugate draws, handed over through device flags (kernels_hdploop.hpp, HdpFork; the
    // event form of round 3 lost more in its two hand-overs than it hid: profiles/r03_labels_notes.md)
    const bool fork = h->fork_armed && !directed;
    ChainView v = h->view();
    HdpLoopBuf hb = hdp_loop_buf(h);
    HdpFork fk{nullptr, 0, 0, nullptr};
    if (fork) fk = HdpFork{h->fork_flags, ++h->fork_ticket, hdp_fork_budget(), h->fork_err_dev};
    {   // label block update (sample_labels.py:134-190) with the transition matrices on the device
        ProfScope ps(h, DLSM_K_LABELS);
        rc = launch_sample_labels<DD>(h, v, (uint32_t)it, nullptr, h->stream,
                                      fork ? h->fork_flags + HF_CENTRED : nullptr, fk.ticket, false);
        if (rc) return rc;
    }
    int nrec = 0;
    bool head = false;
    if (fork) {
        rc = hdp_queue_wait(h, h->fork_stream, fk, (int)HF_CENTRED); if (rc) return rc;
        hipStream_t keep = h->stream;
        h->stream = h->fork_stream; h->ll_beside_chain = true;
        rc = loglik_records(h, 2, h->lsm->cand, nullptr, nullptr, &nrec);
        h->stream = keep; h->ll_beside_chain = false;
        if (rc) return rc;
        // ... and, when another iteration follows and its sweep is the pipelined one, the HEAD of that sweep:
        // the proposal pass and the first, evaluate-only launch need the settled intercept, the positions and
        // the step sizes - nothing the label update or the conjugate draws produce - so they run here, beside
        // those, and the chain's queue starts the sweep at its second launch.  The flag then says "head done".
        head = draw_next && h->next_prop_ok && resolve_sweep_algo(h, h->hdp_cfg.sweep_algo) == 4 &&
               !(getenv("DLSM_HDP_HEAD") && atoi(getenv("DLSM_HDP_HEAD")) == 0);
        if (head)       // (the intercept step and the proposal pass in one launch)
            hipLaunchKernelGGL((k_hdp_intercept_fork_propose<DD>), dim3(1 + propose_blocks(T, N)), dim3(256), 0,
                               h->fork_stream, h->partials, nrec, h->lsm, h->hdp, h->intercept, h->trace_ic, it,
                               v, h->next_prop);
        else
        hipLaunchKernelGGL(k_hdp_intercept_fork, dim3(1), dim3(HDP_THREADS), 0, h->fork_stream, h->partials, nrec,
                           h->lsm, h->hdp, h->intercept, h->trace_ic, it, fk, (int)HF_SETTLED);
        if (head) {
            h->prop_drawn_for = (long)it + 1;      // (the sweep's head below finds its proposals drawn)
            h->stream = h->fork_stream; h->sweep_part = 1;
            rc = enqueue_sweep(h, IterRef{(uint32_t)(it + 1), nullptr}, h->hdp_cfg.sweep_algo);
            h->sweep_part = 0; h->stream = keep;
            if (rc) return rc;
            // ("head done" as a write packet of the queue - hipStreamWriteValue32 - instead of this one-wavefront
            // launch: 3980 against 4008 it/s, round 5)
            hipLaunchKernelGGL(k_fork_set, dim3(1), dim3(64), 0, h->fork_stream, fk, (int)HF_SETTLED);
            h->prop_drawn_for = (long)it + 1; h->head_done_for = (long)it + 1;
        }
    } else if (!directed) { rc = loglik_records(h, 2, h->lsm->cand, nullptr, nullptr, &nrec); if (rc) return rc; }
    ProfScope ps(h, DLSM_K_HDP_TAIL);
    const int n_tab = T * hdp_tab_groups(K);    // (the label counts are a role of this launch)
    // (directed models: the intercepts were settled above - the launch goes without the role's workgroup)
    hipLaunchKernelGGL((k_hdp_stage1<DD>), dim3(n_tab + K * T + (directed || fork ? 0 : 1)), dim3(HDP_THREADS),
                       (size_t)(K * K + K) * sizeof(int32_t), h->stream, v, hb, h->hdp, h->lsm, h->partials, nrec,
                       h->intercept, h->trace_ic, ir, h->htr_z + (size_t)it * T * N);
    hipLaunchKernelGGL((k_hdp_stage2<DD>), dim3(2 + K * T), dim3(HDP_THREADS), 0, h->stream, v, hb,
                       h->hdp, ir);
    hipLaunchKernelGGL((k_hdp_stage3<DD>), dim3(HW_SPLIT * (T - 1) + 1 + K * T), dim3(HDP_THREADS),
                       (size_t)(K * K + K) * sizeof(double), h->stream, v, hb, h->hdp, ir);
    HdpTrace tr{h->trace_ic, h->trace_logp, h->htr_mu, h->htr_sigma, h->htr_beta, h->htr_w,
                h->htr_lambda, h->htr_hyper};
    if (head) {                             // the sweep's head is on the second queue: this launch waits for it
        ProposeBuf none = h->next_prop;
        none.consts = nullptr;
        hipLaunchKernelGGL((k_hdp_hypers_propose<DD>), dim3(1), dim3(HH_THREADS), 0, h->stream, v, hb, h->hdp, tr,
                           ir, none, fk);
    } else if (draw_next && h->next_prop_ok) {     // with the next sweep's proposal pass (kernels_tail_propose.hpp)
        hipLaunchKernelGGL((k_hdp_hypers_propose<DD>), dim3(1 + propose_blocks(T, N)), dim3(HH_THREADS), 0,
                           h->stream, v, hb, h->hdp, tr, ir, h->next_prop, fk);
        h->prop_drawn_for = (long)it + 1;
    } else {
        hipLaunchKernelGGL(k_hdp_hypers, dim3(1), dim3(HH_THREADS), 0, h->stream, v, hb, h->hdp, tr, ir);
        // whatever follows on the chain's queue (a sweep with its own proposal pass) starts from the
        // settled intercept
        if (fork) { rc = hdp_queue_wait(h, h->stream, fk, (int)HF_SETTLED); if (rc) return rc; }
    }
    HIPCHK(h, hipGetLastError());
    return DLSM_OK;
}

// the log-posterior of the stored samples first .. first + count - 1, in chunks of samples
template <int DD>
int enqueue_hdp_logp_batch(dlsm_chain *h, int first, int count) {
    const int T = h->T, K = h->K;
    constexpr int CHUNK = 512;
    const size_t per = (size_t)T * K;
    const size_t need = (size_t)std::min(count, CHUNK) * per;           // doubles + int32
    int rc = ensure_partials(h, 2 * need + (need + 1) / 2); if (rc) return rc;
    double *LP = h->partials, *LPD = h->partials + need;
    int32_t *cnt = (int32_t *)(h->partials + 2 * need);
    ChainView v = h->view();
    HdpTraceView tv{h->trace_X, h->htr_z, h->trace_ic, h->htr_mu, h->htr_sigma, h->htr_beta, h->htr_w,
                    h->htr_lambda, h->htr_hyper, h->trace_logp};
    ProfScope ps(h, DLSM_K_HDP_TAIL);
    for (int s0 = first; s0 < first + count; s0 += CHUNK) {
        const int ns = std::min(CHUNK, first + count - s0);
        hipLaunchKernelGGL((k_hdp_logp_batch_sums<DD>), dim3(K, T, ns), dim3(HDP_THREADS), 0, h->stream,
                           v, tv, s0, h->hdp_cfg.a, LP, cnt, LPD);
        hipLaunchKernelGGL((k_hdp_logp_batch_finish<DD>), dim3(ns), dim3(HF_THREADS), 0, h->stream, v, tv, s0,
                           h->hdp, h->lsm, LP, cnt, LPD);
    }
    HIPCHK(h, hipGetLastError());
    return DLSM_OK;
}

// The likelihood pass on a queue of its own: the undirected model with the matrix-core label kernel
// (which carries the hand-over flag), outside profiling runs (their per-launch events serialise the
// queues).  DLSM_HDP_QUEUES=1 never, =2 always; otherwise only while this is the process's only live
// chain.  Every waiter is ENQUEUED after the launch it waits for, so even two queues that the runtime
// has mapped onto one hardware queue (more streams alive than it has queues) cannot wait for each
// other for ever - they only lose the overlap; the poll budget is the net under that argument.  The
// rule is about what was measured: one chain alone on its device gains, chains that share a device
// (threads of one process: untested; processes: 2.4 times slower, multichain.launch_ranks) do not.
int hdp_fork_arm(dlsm_chain *h) {
    const char *e = getenv("DLSM_HDP_QUEUES");
    const int mode = e ? atoi(e) : 0;
    h->fork_armed = false;
    if (mode == 1 || h->model != DLSM_UNDIRECTED || h->profiling || !labels_mfma_path(h)) return DLSM_OK;
    if (mode != 2 && g_live_chains.load() != 1) return DLSM_OK;
    if (!h->fork_stream) {
        int lo = 0, hi = 0;                        // (numerically greatest = lowest priority: the pass
        HIPCHK(h, hipDeviceGetStreamPriorityRange(&lo, &hi));   // yields to the chain's small launches)
        // (measured, profiles/r04_hdp_two_queues.md: the priority changes nothing; capping the pass's
        // workgroups per CU with unused LDS changes nothing; a CU mask of 160 is +0.5 %, 192 .. 240 are -1 %)
        const char *ec = getenv("DLSM_HDP_FORK_CUS");
        if (ec && atoi(ec) > 0) {                  // (experiment: the pass on a subset of the CUs)
            const int ncu = atoi(ec);
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < ncu && i < 256; ++i) mask[i >> 5] |= 1u << (i & 31);
            HIPCHK(h, hipExtStreamCreateWithCUMask(&h->fork_stream, 8, mask));
        } else
        HIPCHK(h, hipStreamCreateWithPriority(&h->fork_stream, hipStreamNonBlocking, lo));
        HIPCHK(h, hipEventCreateWithFlags(&h->fork_ev, hipEventDisableTiming));
        HIPCHK(h, hipMalloc((void **)&h->fork_flags, 64));
        HIPCHK(h, hipMemset(h->fork_flags, 0, 64));
        NEED(h, h->fork_err_dev != nullptr, "no mapped host memory for the error word of the in-kernel waits");
        int can = 0;
        if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, h->device) != hipSuccess) can = 0;
        (void)hipGetLastError();
        const char *eg = getenv("DLSM_HDP_GATE");
        // (under rocprofv3's counter collection - it serialises the dispatches of all queues - the queue-level
        // wait never returned: a process the profiler's tool library is loaded into keeps the gate kernel)
        const bool profiler = getenv("ROCP_TOOL_LIBRARIES") != nullptr || getenv("ROCPROFILER_REGISTER_FORCE_LOAD") != nullptr;
        h->fork_wait_value = can != 0 && !profiler && !(eg && strcmp(eg, "kernel") == 0);
        h->fork_ticket = 0;
    }
    h->fork_armed = true;
    if (getenv("DLSM_DEBUG")) fprintf(stderr, "dlsm: HDP-LPCM loop on two queues (live chains %d)\n", g_live_chains.load());
    return DLSM_OK;
}

int check_ready_hdp(dlsm_chain *h) {
    NEED(h, h->model == DLSM_UNDIRECTED || h->have_radii, "radii not set");
    NEED(h, h->have_prior && h->prior_kind == DLSM_PRIOR_MIXTURE,
         "set the mixture prior (mu, sigma, lmbda, z) first");
    return check_ready_sweep(h);
}

}  // namespace

extern "C" {

int dlsm_hdp_configure(dlsm_chain *h, const dlsm_hdp_config *cfg, const double *beta,
                       const double *weights) {
    NEED(h, h && cfg && beta && weights, "null argument");
    drop_graph(h);
    HIPCHK(h, hipSetDevice(h->device));
    int rc = check_ready_hdp(h); if (rc) return rc;
    NEED(h, cfg->intercept_variance_prior > 0 && cfg->i_tune_interval > 0 &&
            cfg->lambda_variance_prior > 0 && cfg->mean_variance_prior > 0 && cfg->b > 0,
         "variances / tune_interval must be positive");
    NEED(h, cfg->gamma > 0 && cfg->alpha_init > 0 && cfg->alpha > 0 && cfg->kappa >= 0,
         "concentration parameters must be positive");
    rc = check_sweep_algo(h, cfg->sweep_algo); if (rc) return rc;
    const int T = h->T, K = h->K;
    rc = ensure_label_bufs(h); if (rc) return rc;
    const size_t need = hdp_loop_buf_doubles(h);
    if (h->hdp_buf_cap < need) {
        if (h->hdp_buf) hipFree(h->hdp_buf);
        h->hdp_buf = nullptr; h->hdp_buf_cap = 0;
        rc = dev_alloc(h, &h->hdp_buf, need); if (rc) return rc;
        h->hdp_buf_cap = need;
    }
    HIPCHK(h, hipMemsetAsync(h->hdp_buf, 0, need * sizeof(double), h->stream));
    h->hdp_K = K;
    // device state: everything but lmbda (dlsm_set_prior_mixture owns it)
    HdpDeviceState s;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(&s, h->hdp, sizeof(s), hipMemcpyDeviceToHost));
    s.gamma = cfg->gamma; s.alpha_init = cfg->alpha_init; s.alpha = cfg->alpha; s.kappa = cfg->kappa;
    s.mvp = cfg->mean_variance_prior; s.b = cfg->b;
    s.a = cfg->a; s.a0 = cfg->a0; s.b0 = cfg->b0; s.c0 = cfg->c0; s.d0 = cfg->d0;
    s.has_a0 = cfg->has_a0 ? 1 : 0; s.has_c0 = cfg->has_c0 ? 1 : 0;
    s.lambda_prior = cfg->lambda_prior; s.lambda_var = cfg->lambda_variance_prior;
    s.gamma_shape = cfg->gamma_prior_shape; s.gamma_rate = cfg->gamma_prior_rate;
    s.alpha0_shape = cfg->alpha_init_shape; s.alpha0_rate = cfg->alpha_init_rate;
    s.ak_shape = cfg->alpha_kappa_shape; s.ak_rate = cfg->alpha_kappa_rate;
    s.ll = 0.0;
    s.mbar_total = s.mbar_positive = s.m00_total = s.m_rest_total = s.override_total = 0.0;
    HIPCHK(h, hipMemcpy(h->hdp, &s, sizeof(s), hipMemcpyHostToDevice));
    // the intercept's sampler shares the LSM loop's device state
    LsmDeviceState ls;
    memset(&ls, 0, sizeof(ls));
    ls.intercept_prior[0] = cfg->intercept_prior;
    ls.intercept_var = cfg->intercept_variance_prior;
    ls.i_step[0] = cfg->i_step_size;
    ls.i_nacc[0] = cfg->i_n_accepted; ls.i_nsteps[0] = cfg->i_n_steps;
    ls.i_until[0] = cfg->i_steps_until_tune;
    ls.i_tune = cfg->i_tune < 0 ? -1 : cfg->i_tune;
    ls.i_tune_interval = cfg->i_tune_interval;
    if (h->model != DLSM_UNDIRECTED) {      // intercept_out and the radii sampler
        NEED(h, cfg->i_step_size_out > 0 && cfg->r_step_size > 0, "directed models: step sizes must be positive");
        ls.intercept_prior[1] = cfg->intercept_prior_out;
        ls.i_step[1] = cfg->i_step_size_out;
        ls.i_nacc[1] = cfg->i_n_accepted_out; ls.i_nsteps[1] = cfg->i_n_steps_out;
        ls.i_until[1] = cfg->i_steps_until_tune_out;
        ls.r_step = cfg->r_step_size;
        ls.r_nacc = cfg->r_n_accepted; ls.r_nsteps = cfg->r_n_steps; ls.r_until = cfg->r_steps_until_tune;
        ls.r_tune = cfg->r_tune < 0 ? -1 : cfg->r_tune;
        ls.r_tune_interval = cfg->r_tune_interval > 0 ? cfg->r_tune_interval : 100;
    }
    HIPCHK(h, hipMemcpy(h->lsm, &ls, sizeof(ls), hipMemcpyHostToDevice));
    HdpLoopBuf hb = hdp_loop_buf(h);
    rc = h2d(h, hb.beta, beta, (size_t)K); if (rc) return rc;
    rc = h2d(h, h->lab_w, weights, (size_t)T * K * K); if (rc) return rc;
    // allocations of the sweep / post / log-likelihood launchers (so that the run only enqueues)
    rc = enqueue_sweep(h, IterRef{0, nullptr}, cfg->sweep_algo, true); if (rc) return rc;
    if (h->model == DLSM_UNDIRECTED) {
        DISPATCH_D(h, h->D, rc = launch_post<DD>(h, nullptr, 0, 1, h->lsm, IterRef{0, nullptr}, nullptr, true));
        if (rc) return rc;
    }
    h->hdp_cfg = *cfg;
    h->hdp_configured = true;
    return DLSM_OK;
}

int dlsm_hdp_get_config(dlsm_chain *h, dlsm_hdp_config *cfg) {
    NEED(h, h && cfg, "null argument");
    NEED(h, h->hdp_configured, "HDP-LPCM loop not configured");
    NEED(h, h->hdp_K == h->K, "n_components changed since dlsm_hdp_configure: configure again");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HdpDeviceState s;
    LsmDeviceState ls;
    HIPCHK(h, hipMemcpy(&s, h->hdp, sizeof(s), hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(&ls, h->lsm, sizeof(ls), hipMemcpyDeviceToHost));
    *cfg = h->hdp_cfg;
    cfg->gamma = s.gamma; cfg->alpha_init = s.alpha_init; cfg->alpha = s.alpha; cfg->kappa = s.kappa;
    cfg->mean_variance_prior = s.mvp; cfg->b = s.b;
    cfg->i_step_size = ls.i_step[0]; cfg->i_n_accepted = ls.i_nacc[0];
    cfg->i_n_steps = ls.i_nsteps[0]; cfg->i_steps_until_tune = ls.i_until[0];
    if (h->model != DLSM_UNDIRECTED) {
        cfg->i_step_size_out = ls.i_step[1]; cfg->i_n_accepted_out = ls.i_nacc[1];
        cfg->i_n_steps_out = ls.i_nsteps[1]; cfg->i_steps_until_tune_out = ls.i_until[1];
        cfg->r_step_size = ls.r_step; cfg->r_n_accepted = ls.r_nacc; cfg->r_n_steps = ls.r_nsteps;
        cfg->r_steps_until_tune = ls.r_until;
    }
    return DLSM_OK;
}

int dlsm_hdp_trace_alloc(dlsm_chain *h, int n_total, double logp0) {
    NEED(h, h && n_total >= 1, "bad argument");
    NEED(h, h->hdp_configured, "configure the HDP-LPCM loop first");
    NEED(h, h->hdp_K == h->K, "n_components changed since dlsm_hdp_configure: configure again");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = dlsm_trace_alloc(h, n_total, logp0); if (rc) return rc;   // Xs_, intercepts_, logps_
    hdp_free_trace(h);
    const size_t T = h->T, N = h->N, K = h->K, D = h->D, n = n_total;
    rc = dev_alloc(h, &h->htr_mu, n * K * D); if (rc) return rc;
    rc = dev_alloc(h, &h->htr_sigma, n * K); if (rc) return rc;
    rc = dev_alloc(h, &h->htr_beta, n * K); if (rc) return rc;
    rc = dev_alloc(h, &h->htr_w, n * T * K * K); if (rc) return rc;
    rc = dev_alloc(h, &h->htr_lambda, n); if (rc) return rc;
    rc = dev_alloc(h, &h->htr_hyper, n * 6); if (rc) return rc;
    rc = dev_alloc(h, &h->htr_z, n * T * N); if (rc) return rc;
    h->htr_n = n_total; h->htr_K = (int)K;
    // row 0 = the current state
    HdpLoopBuf hb = hdp_loop_buf(h);
    HdpDeviceState s;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(&s, h->hdp, sizeof(s), hipMemcpyDeviceToHost));
    const double hy[6] = {s.gamma, s.alpha_init, s.alpha, s.kappa, s.mvp, s.b};
    const double nanv = std::nan("");     // (both live until the synchronisation that ends the copies)
    HIPCHK(h, hipMemcpyAsync(h->htr_mu, h->mu, K * D * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->htr_sigma, h->sigma, K * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->htr_beta, hb.beta, K * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->htr_w, h->lab_w, T * K * K * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->htr_lambda, &h->hdp->lmbda, sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->htr_hyper, hy, sizeof(hy), hipMemcpyHostToDevice, h->stream));
    // row 0's network log-likelihood is not known here (NaN: "evaluate it if you need it"); the
    // directed models keep intercept_out in that slot
    if (h->model == DLSM_UNDIRECTED)
        HIPCHK(h, hipMemcpyAsync(h->trace_ic + 1, &nanv, sizeof(double), hipMemcpyHostToDevice, h->stream));
    const long tn = (long)(T * N);
    hipLaunchKernelGGL(k_hdp_trace_labels, dim3((unsigned)((tn + 255) / 256)), dim3(256), 0, h->stream,
                       h->z, tn, h->htr_z);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DLSM_OK;
}

int dlsm_hdp_run(dlsm_chain *h, int first, int count) {
    NEED(h, h != nullptr, "null handle");
    NEED(h, h->hdp_configured && h->trace_X && h->htr_z, "configure the loop and allocate its trace first");
    NEED(h, h->hdp_K == h->K && h->htr_K == h->K, "n_components changed since dlsm_hdp_configure");
    NEED(h, first >= 1 && count >= 0 && first + count <= h->trace_n && first + count <= h->htr_n,
         "iteration range out of the trace");
    HIPCHK(h, hipSetDevice(h->device));
    int rc = check_ready_hdp(h); if (rc) return rc;
    // (read per call: the tests switch it inside one process)
    const bool ride = !(getenv("DLSM_TAIL_PROPOSE") && atoi(getenv("DLSM_TAIL_PROPOSE")) == 0);
    h->prop_drawn_for = -1; h->head_done_for = -1;
    rc = hdp_fork_arm(h); if (rc) return rc;
    for (int it = first; it < first + count; ++it) {
        DISPATCH_D(h, h->D, rc = enqueue_hdp_iteration<DD>(h, it, ride && it + 1 < first + count));
        if (rc) break;
    }
    if (h->fork_armed && count > 0) {       // ONE event per call: the second queue's last intercept step
        HIPCHK(h, hipEventRecord(h->fork_ev, h->fork_stream));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->fork_ev, 0));
    }
    h->head_done_for = -1;
    if (rc) return rc;
    h->prop_drawn_for = -1;
    // the log-posterior trace of these rows: one batched pass over the trace, behind the iterations
    if (count > 0) DISPATCH_D(h, h->D, rc = enqueue_hdp_logp_batch<DD>(h, first, count));
    return rc;
}

int dlsm_hdp_queues(dlsm_chain *h, int *queues) {
    NEED(h, h && queues, "null argument");
    *queues = h->fork_armed ? 2 : 1;
    return DLSM_OK;
}

int dlsm_hdp_trace_read(dlsm_chain *h, int first, int count, double *Xs, double *intercepts,
                        double *logps, double *mus, double *sigmas, int64_t *zs, double *betas,
                        double *weights, double *lambdas, double *hypers) {
    NEED(h, h != nullptr, "null handle");
    NEED(h, h->htr_z, "no HDP-LPCM trace allocated");
    NEED(h, first >= 0 && count >= 0 && first + count <= h->htr_n, "range out of the trace");
    int rc = dlsm_trace_read(h, first, count, Xs, intercepts, logps); if (rc) return rc;
    const size_t T = h->T, N = h->N, K = h->htr_K, D = h->D, f = first, c = count;
    auto get = [&](double *dst, const double *src, size_t per) -> hipError_t {
        return dst ? hipMemcpy(dst, src + f * per, c * per * sizeof(double), hipMemcpyDeviceToHost)
                   : hipSuccess;
    };
    HIPCHK(h, get(mus, h->htr_mu, K * D));
    HIPCHK(h, get(sigmas, h->htr_sigma, K));
    HIPCHK(h, get(betas, h->htr_beta, K));
    HIPCHK(h, get(weights, h->htr_w, T * K * K));
    HIPCHK(h, get(lambdas, h->htr_lambda, 1));
    HIPCHK(h, get(hypers, h->htr_hyper, 6));
    if (zs) {
        std::vector<uint8_t> tmp(c * T * N);
        HIPCHK(h, hipMemcpy(tmp.data(), h->htr_z + f * T * N, tmp.size(), hipMemcpyDeviceToHost));
        for (size_t q = 0; q < tmp.size(); ++q) zs[q] = tmp[q];
    }
    return DLSM_OK;
}

int dlsm_hdp_trace_write(dlsm_chain *h, int first, int count, const double *Xs,
                         const double *intercepts, const double *logps, const double *mus,
                         const double *sigmas, const int64_t *zs, const double *betas,
                         const double *weights, const double *lambdas) {
    NEED(h, h != nullptr, "null handle");
    NEED(h, h->htr_z && h->trace_X, "no HDP-LPCM trace allocated");
    NEED(h, first >= 0 && count >= 0 && first + count <= h->htr_n && first + count <= h->trace_n,
         "range out of the trace");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const size_t T = h->T, N = h->N, K = h->htr_K, D = h->D, f = first, c = count;
    auto put = [&](double *dst, const double *src, size_t per) -> hipError_t {
        return src ? hipMemcpy(dst + f * per, src, c * per * sizeof(double), hipMemcpyHostToDevice)
                   : hipSuccess;
    };
    HIPCHK(h, put(h->trace_X, Xs, T * N * D));
    HIPCHK(h, put(h->trace_logp, logps, 1));
    HIPCHK(h, put(h->htr_mu, mus, K * D));
    HIPCHK(h, put(h->htr_sigma, sigmas, K));
    HIPCHK(h, put(h->htr_beta, betas, K));
    HIPCHK(h, put(h->htr_w, weights, T * K * K));
    HIPCHK(h, put(h->htr_lambda, lambdas, 1));
    if (intercepts && h->model != DLSM_UNDIRECTED) {
        HIPCHK(h, hipMemcpy(h->trace_ic + 2 * f, intercepts, c * 2 * sizeof(double), hipMemcpyHostToDevice));
    } else if (intercepts) {                // the device keeps two per row (undirected: the second unused)
        std::vector<double> tmp(c * 2, std::nan(""));    // (second slot: network log-likelihood, unknown)
        for (size_t q = 0; q < c; ++q) tmp[2 * q] = intercepts[q];
        HIPCHK(h, hipMemcpy(h->trace_ic + 2 * f, tmp.data(), tmp.size() * sizeof(double),
                            hipMemcpyHostToDevice));
    }
    if (zs) {
        std::vector<uint8_t> tmp(c * T * N);
        for (size_t q = 0; q < tmp.size(); ++q) {
            if (zs[q] < 0 || zs[q] >= (int64_t)K) FAIL(h, DLSM_E_DATA, "label out of range at %zu", q);
            tmp[q] = (uint8_t)zs[q];
        }
        HIPCHK(h, hipMemcpy(h->htr_z + f * T * N, tmp.data(), tmp.size(), hipMemcpyHostToDevice));
    }
    return DLSM_OK;
}

int dlsm_hdp_get_aux(dlsm_chain *h, int64_t *m, double *m_bar, int64_t *w_over, int64_t *n,
                     int64_t *nk) {
    NEED(h, h != nullptr, "null handle");
    NEED(h, h->hdp_configured, "HDP-LPCM loop not configured");
    NEED(h, h->hdp_K == h->K, "n_components changed since dlsm_hdp_configure: configure again");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const size_t T = h->T, K = h->K;
    HdpLoopBuf hb = hdp_loop_buf(h);
    std::vector<int32_t> tmp(T * K * K);
    auto geti = [&](int64_t *dst, const int32_t *src, size_t cnt) -> hipError_t {
        if (!dst) return hipSuccess;
        hipError_t e = hipMemcpy(tmp.data(), src, cnt * sizeof(int32_t), hipMemcpyDeviceToHost);
        for (size_t q = 0; q < cnt; ++q) dst[q] = tmp[q];
        return e;
    };
    HIPCHK(h, geti(m, hb.m, T * K * K));
    HIPCHK(h, geti(w_over, hb.wover, (T - 1) * K));
    HIPCHK(h, geti(n, hb.n, T * K * K));
    HIPCHK(h, geti(nk, hb.nk, T * K));
    if (m_bar) HIPCHK(h, hipMemcpy(m_bar, hb.mbar, K * sizeof(double), hipMemcpyDeviceToHost));
    return DLSM_OK;
}

}  // extern "C"
