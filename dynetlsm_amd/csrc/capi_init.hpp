// C-ABI of the initialisation pipeline (included by capi.hip, which provides
// the FAIL / HIPCHK / NEED / DISPATCH_D macros and the small helpers).
#pragma once
#include "host_tridiag.hpp"

namespace {

using dlsm_host::tridiag_top;

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    template <typename T> T *as() const { return (T *)p; }
};

int init_need_hops(dlsm_chain *h) {
    NEED(h, h->have_hops, "hop matrices not computed (dlsm_init_shortest_paths)");
    return DLSM_OK;
}

template <int DD>
int run_smacof(dlsm_chain *h, int t, int n_init, const double *X0, int max_iter, double eps,
               double *X_out, double *stress_out, int32_t *n_iter_out) {
    const int N = h->N;
    const size_t per = (size_t)N * DD;
    const int nblk = (N + SM_ROWS - 1) / SM_ROWS;
    DevBuf bX, bRec, bSt;
    HIPCHK(h, hipMalloc(&bX.p, 2 * n_init * per * sizeof(double)));
    HIPCHK(h, hipMalloc(&bRec.p, (size_t)n_init * nblk * 2 * sizeof(double)));
    HIPCHK(h, hipMalloc(&bSt.p, n_init * sizeof(SmacofState)));
    double *Xb[2] = {bX.as<double>(), bX.as<double>() + n_init * per};
    HIPCHK(h, hipMemsetAsync(bSt.p, 0, n_init * sizeof(SmacofState), h->stream));
    HIPCHK(h, hipMemcpyAsync(Xb[0], X0, n_init * per * sizeof(double), hipMemcpyHostToDevice,
                             h->stream));
    const uint16_t *hops = h->hops + (size_t)t * N * N;
    std::vector<SmacofState> st(n_init);
    {
        ProfScope ps(h, DLSM_K_INIT);
        for (int p = 0; p <= max_iter; ++p) {
            hipLaunchKernelGGL((k_smacof_pass<DD>), dim3(nblk, n_init), dim3(SM_THREADS), 0,
                               h->stream, hops, N, Xb[p & 1], Xb[(p + 1) & 1],
                               bRec.as<double>(), bSt.as<SmacofState>());
            hipLaunchKernelGGL(k_smacof_check, dim3(n_init), dim3(256), 0, h->stream,
                               bRec.as<double>(), nblk, p, max_iter, eps,
                               bSt.as<SmacofState>());
            if ((p & 15) == 15 && p < max_iter) {
                HIPCHK(h, hipMemcpyAsync(st.data(), bSt.p, n_init * sizeof(SmacofState),
                                         hipMemcpyDeviceToHost, h->stream));
                HIPCHK(h, hipStreamSynchronize(h->stream));
                bool all = true;
                for (auto &s : st) all = all && s.done;
                if (all) break;
            }
        }
    }
    HIPCHK(h, hipGetLastError());
    int rc = d2h(h, st.data(), bSt.as<SmacofState>(), (size_t)n_init);
    if (rc) return rc;
    for (int r = 0; r < n_init; ++r) {
        rc = d2h(h, X_out + r * per, Xb[st[r].answer] + r * per, per);
        if (rc) return rc;
        stress_out[r] = st[r].stress;
        n_iter_out[r] = st[r].n_iter;
    }
    return DLSM_OK;
}

template <int DD>
int run_gmds_step(dlsm_chain *h, int t, const double *X_prev, double lmbda, int max_lanczos,
                  double tol, double *X_out, double *evals_out, int32_t *n_lanczos_out,
                  double *resid_out) {
    const int N = h->N;
    const int kmax = std::min(max_lanczos, N);
    NEED(h, kmax >= DD, "max_lanczos must be at least n_features");
    NEED(h, (size_t)(kmax + 1) * sizeof(double) <= 60000, "max_lanczos too large");
    const double alpha_w = 1.0 / (1.0 + lmbda), beta_w = lmbda / (1.0 + lmbda);
    DevBuf bQ, bV, bXp, bXo, bS;
    HIPCHK(h, hipMalloc(&bQ.p, (size_t)(kmax + 1) * N * sizeof(double)));
    HIPCHK(h, hipMalloc(&bV.p, ((size_t)2 * N + 2 * kmax + 16) * sizeof(double)));
    HIPCHK(h, hipMalloc(&bXp.p, (size_t)N * DD * sizeof(double)));
    HIPCHK(h, hipMalloc(&bXo.p, (size_t)N * DD * sizeof(double)));
    HIPCHK(h, hipMalloc(&bS.p, ((size_t)DD * kmax + DD) * sizeof(double)));
    double *Q = bQ.as<double>();
    double *wB = bV.as<double>(), *w = wB + N, *ab = w + N, *small = ab + 2 * kmax;
    double *Xp = bXp.as<double>(), *Xo = bXo.as<double>();
    double *S = bS.as<double>(), *theta_d = S + (size_t)DD * kmax;
    HIPCHK(h, hipMemcpyAsync(Xp, X_prev, (size_t)N * DD * sizeof(double), hipMemcpyHostToDevice,
                             h->stream));
    const uint16_t *hops = h->hops + (size_t)t * N * N;
    const int nblk = (N + SM_ROWS - 1) / SM_ROWS;
    std::vector<double> hab(2 * kmax);
    std::vector<std::vector<double>> Sh;
    double theta[DLSM_D_MAX] = {};
    int k_used = 0;
    double resid = 0.0;
    {
        ProfScope ps(h, DLSM_K_INIT);
        hipLaunchKernelGGL((k_lanczos_init<DD>), dim3(1), dim3(LZ_THREADS), 0, h->stream, N, t,
                           h->seed, Xp, Q, small);
        for (int k = 0; k < kmax; ++k) {
            hipLaunchKernelGGL(k_gmds_matvec, dim3(nblk), dim3(SM_THREADS), 0, h->stream, hops,
                               N, Q + (size_t)k * N, small, wB);
            hipLaunchKernelGGL((k_lanczos_step<DD>), dim3(1), dim3(LZ_THREADS),
                               (kmax + 1) * sizeof(double), h->stream, N, k, kmax, alpha_w,
                               beta_w, Xp, Q, wB, w, small, ab);
            const int kk = k + 1;
            if (kk < DD) continue;
            if ((kk % 8) != 0 && kk != kmax) continue;
            HIPCHK(h, hipMemcpyAsync(hab.data(), ab, 2 * kmax * sizeof(double),
                                     hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            tridiag_top(hab.data(), hab.data() + kmax, kk, DD, theta, Sh);
            const double bk = hab[kmax + k];
            double scale = 0.0;
            for (int m = 0; m < DD; ++m) scale = std::max(scale, fabs(theta[m]));
            resid = 0.0;
            for (int m = 0; m < DD; ++m)
                resid = std::max(resid, fabs(bk * Sh[m][kk - 1]) / std::max(scale, 1e-300));
            k_used = kk;
            if (resid <= tol || kk == N) break;
        }
        std::vector<double> flat((size_t)DD * k_used + DD);
        for (int m = 0; m < DD; ++m)
            for (int i = 0; i < k_used; ++i) flat[(size_t)m * k_used + i] = Sh[m][i];
        HIPCHK(h, hipMemcpyAsync(S, flat.data(), (size_t)DD * k_used * sizeof(double),
                                 hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(theta_d, theta, DD * sizeof(double), hipMemcpyHostToDevice,
                                 h->stream));
        hipLaunchKernelGGL((k_gmds_finish<DD>), dim3(1), dim3(LZ_THREADS), 0, h->stream, N,
                           k_used, S, theta_d, Q, Xp, Xo);
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    HIPCHK(h, hipGetLastError());
    int rc = d2h(h, X_out, Xo, (size_t)N * DD);
    if (rc) return rc;
    if (evals_out) for (int m = 0; m < DD; ++m) evals_out[m] = theta[m];
    if (n_lanczos_out) *n_lanczos_out = k_used;
    if (resid_out) *resid_out = resid;
    return DLSM_OK;
}

}  // namespace

extern "C" {

int dlsm_init_shortest_paths(dlsm_chain *h) {
    NEED(h, h != nullptr, "null handle");
    NEED(h, h->model != DLSM_DIRECTED_CASE_CONTROL && h->have_network,
         "needs the bit-packed network (dlsm_upload_network)");
    NEED(h, h->N < 65535, "N must be < 65535 for uint16 hop counts");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t n = (size_t)h->T * h->N * h->N;
    if (!h->hops) {
        HIPCHK(h, hipMalloc((void **)&h->hops, n * sizeof(uint16_t)));
        HIPCHK(h, hipMalloc((void **)&h->hops_max, h->T * sizeof(int)));
    }
    HIPCHK(h, hipMemsetAsync(h->hops_max, 0, h->T * sizeof(int), h->stream));
    const size_t lds = (size_t)3 * h->W * sizeof(uint32_t) + (size_t)h->N * sizeof(uint16_t);
    NEED(h, lds <= 160 * 1024, "N too large for the BFS workgroup's LDS");
    ChainView v = h->view();
    {
        ProfScope ps(h, DLSM_K_INIT);
        if (lds > 64 * 1024)
            HIPCHK(h, hipFuncSetAttribute((const void *)k_hops_bfs,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_hops_bfs, dim3(h->N, h->T), dim3(BFS_THREADS), lds, h->stream, v,
                           h->hops, h->hops_max);
        const int fb = (int)std::min<size_t>(4096, ((size_t)h->N * h->N + 255) / 256);
        hipLaunchKernelGGL(k_hops_fill, dim3(fb, h->T), dim3(256), 0, h->stream, h->hops, h->N,
                           h->hops_max);
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_hops = true;
    return DLSM_OK;
}

int dlsm_init_get_dissimilarity(dlsm_chain *h, int t, double *out) {
    NEED(h, h && out, "null argument");
    int rc = init_need_hops(h); if (rc) return rc;
    NEED(h, t >= 0 && t < h->T, "t out of range");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t n2 = (size_t)h->N * h->N;
    DevBuf b;
    HIPCHK(h, hipMalloc(&b.p, n2 * sizeof(double)));
    const int nb = (int)std::min<size_t>(4096, (n2 + 255) / 256);
    hipLaunchKernelGGL(k_hops_to_double, dim3(nb), dim3(256), 0, h->stream,
                       h->hops + (size_t)t * n2, n2, b.as<double>());
    HIPCHK(h, hipGetLastError());
    return d2h(h, out, b.as<double>(), n2);
}

int dlsm_init_smacof(dlsm_chain *h, int t, int n_init, const double *X0, int max_iter,
                     double eps, double *X_out, double *stress_out, int32_t *n_iter_out) {
    NEED(h, h && X0 && X_out && stress_out && n_iter_out, "null argument");
    int rc = init_need_hops(h); if (rc) return rc;
    NEED(h, t >= 0 && t < h->T, "t out of range");
    NEED(h, n_init >= 1 && n_init <= 64 && max_iter >= 1, "bad n_init / max_iter");
    HIPCHK(h, hipSetDevice(h->device));
    DISPATCH_D(h, h->D, rc = run_smacof<DD>(h, t, n_init, X0, max_iter, eps, X_out,
                                            stress_out, n_iter_out));
    return rc;
}

int dlsm_init_gmds_step(dlsm_chain *h, int t, const double *X_prev, double lmbda,
                        int max_lanczos, double tol, double *X_out, double *evals_out,
                        int32_t *n_lanczos_out, double *resid_out) {
    NEED(h, h && X_prev && X_out, "null argument");
    int rc = init_need_hops(h); if (rc) return rc;
    NEED(h, t >= 0 && t < h->T, "t out of range");
    NEED(h, lmbda >= 0.0 && tol > 0.0, "bad lmbda / tol");
    HIPCHK(h, hipSetDevice(h->device));
    DISPATCH_D(h, h->D, rc = run_gmds_step<DD>(h, t, X_prev, lmbda, max_lanczos, tol, X_out,
                                               evals_out, n_lanczos_out, resid_out));
    return rc;
}

int dlsm_init_mle_sums(dlsm_chain *h, double p0, double p1, double *out) {
    NEED(h, h && out, "null argument");
    NEED(h, h->model != DLSM_DIRECTED_CASE_CONTROL && h->have_network,
         "needs the bit-packed network (dlsm_upload_network)");
    NEED(h, h->have_X, "latent positions not set");
    if (h->model == DLSM_DIRECTED) NEED(h, h->have_radii, "radii not set");
    HIPCHK(h, hipSetDevice(h->device));
    const int nblk = (h->N + SM_ROWS - 1) / SM_ROWS;
    const size_t nrec = (size_t)nblk * h->T;
    int rc = ensure_partials(h, nrec * 3); if (rc) return rc;
    ChainView v = h->view();
    {
        ProfScope ps(h, DLSM_K_INIT);
        if (h->model == DLSM_UNDIRECTED) {
            DISPATCH_D(h, h->D, hipLaunchKernelGGL((k_mle_sums<DD, DLSM_UNDIRECTED>),
                       dim3(nblk, h->T), dim3(SM_THREADS), 0, h->stream, v, p0, p1, h->partials));
        } else {
            DISPATCH_D(h, h->D, hipLaunchKernelGGL((k_mle_sums<DD, DLSM_DIRECTED>),
                       dim3(nblk, h->T), dim3(SM_THREADS), 0, h->stream, v, p0, p1, h->partials));
        }
        hipLaunchKernelGGL((k_sum_records<3>), dim3(1), dim3(1024), 0, h->stream, h->partials,
                           nrec, h->dsmall);
    }
    HIPCHK(h, hipGetLastError());
    return d2h(h, out, h->dsmall, 3);
}

int dlsm_init_release(dlsm_chain *h) {
    NEED(h, h != nullptr, "null handle");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->hops) hipFree(h->hops);
    if (h->hops_max) hipFree(h->hops_max);
    h->hops = nullptr; h->hops_max = nullptr; h->have_hops = false;
    return DLSM_OK;
}

int dlsm_init_kmeans_lloyd(dlsm_chain *h, const double *X, int N, int F, int K,
                           const double *centers_init, int max_iter, double tol, double *centers_out,
                           int32_t *labels_out, int32_t *n_iter_out, int32_t *empty_out) {
    NEED(h, h && X && centers_init && centers_out && labels_out && n_iter_out && empty_out, "null argument");
    NEED(h, N >= 1 && F >= 1 && K >= 1 && K <= N && max_iter >= 1, "bad sizes");
    const size_t lds = ((size_t)K * F + K) * sizeof(double);
    NEED(h, lds <= 64 * 1024, "the centres (n_clusters x T d doubles) do not fit the assignment kernel's LDS");
    HIPCHK(h, hipSetDevice(h->device));
    DevBuf dX, dC, dL, dS;
    HIPCHK(h, hipMalloc(&dX.p, (size_t)N * F * sizeof(double)));
    HIPCHK(h, hipMalloc(&dC.p, (size_t)2 * K * F * sizeof(double)));
    HIPCHK(h, hipMalloc(&dL.p, (size_t)N * sizeof(int32_t)));
    HIPCHK(h, hipMalloc(&dS.p, (size_t)K * (sizeof(double) + sizeof(int32_t)) + 16));
    double *cen[2] = {dC.as<double>(), dC.as<double>() + (size_t)K * F};
    double *d_shift = dS.as<double>();
    int32_t *d_counts = (int32_t *)(d_shift + K), *d_changed = d_counts + K;
    HIPCHK(h, hipMemcpyAsync(dX.p, X, (size_t)N * F * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(cen[0], centers_init, (size_t)K * F * sizeof(double), hipMemcpyHostToDevice,
                             h->stream));
    HIPCHK(h, hipMemsetAsync(dL.p, 0xFF, (size_t)N * sizeof(int32_t), h->stream));      // labels = -1
    std::vector<double> shift(K);
    std::vector<int32_t> counts(K + 1);
    const int nb = (N + 255) / 256;
    int cur = 0, it = 0;
    bool strict = false;
    *empty_out = 0;
    ProfScope ps(h, DLSM_K_INIT);
    // _kmeans_single_lloyd (sklearn/cluster/_kmeans.py): E + M step, then "labels unchanged"
    // (strict convergence) or the squared centre shift within tol; when the loop ends otherwise,
    // one more E-step so that the labels match the returned centres
    for (it = 0; it < max_iter; ++it) {
        HIPCHK(h, hipMemsetAsync(d_changed, 0, sizeof(int32_t), h->stream));
        hipLaunchKernelGGL(k_kmeans_assign, dim3(nb), dim3(256), lds, h->stream, dX.as<double>(), N, F, K,
                           cen[cur], dL.as<int32_t>(), d_changed);
        hipLaunchKernelGGL(k_kmeans_update, dim3(K), dim3(256), 0, h->stream, dX.as<double>(), N, F,
                           dL.as<int32_t>(), cen[cur], cen[cur ^ 1], d_counts, d_shift);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(shift.data(), d_shift, K * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(counts.data(), d_counts, (K + 1) * sizeof(int32_t), hipMemcpyDeviceToHost,
                                 h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        for (int k = 0; k < K; ++k)
            if (counts[k] == 0) { *empty_out = 1; *n_iter_out = it + 1; return DLSM_OK; }   // relocation: the caller's
        cur ^= 1;
        if (counts[K] == 0) { strict = true; ++it; break; }
        double tot = 0.0;
        for (int k = 0; k < K; ++k) tot += shift[k];
        if (tot <= tol) { ++it; break; }
    }
    if (!strict) {
        HIPCHK(h, hipMemsetAsync(d_changed, 0, sizeof(int32_t), h->stream));
        hipLaunchKernelGGL(k_kmeans_assign, dim3(nb), dim3(256), lds, h->stream, dX.as<double>(), N, F, K,
                           cen[cur], dL.as<int32_t>(), d_changed);
        HIPCHK(h, hipGetLastError());
    }
    *n_iter_out = std::min(it, max_iter);
    HIPCHK(h, hipMemcpyAsync(centers_out, cen[cur], (size_t)K * F * sizeof(double), hipMemcpyDeviceToHost,
                             h->stream));
    HIPCHK(h, hipMemcpyAsync(labels_out, dL.p, (size_t)N * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DLSM_OK;
}

}  // extern "C"
