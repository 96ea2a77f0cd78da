// C-ABI of the post-loop processing (included by capi.hip after capi_init.hpp, which
// provides DevBuf; the FAIL / HIPCHK / NEED macros come from capi.hip).
#pragma once

extern "C" {

int dlsm_post_release(dlsm_chain *h) {
    NEED(h, h != nullptr, "null handle");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->post_zt) hipFree(h->post_zt);
    if (h->post_cooc) hipFree(h->post_cooc);
    h->post_zt = nullptr; h->post_cooc = nullptr; h->post_S = h->post_Spad = 0;
    return DLSM_OK;
}

int dlsm_post_cooccurrence(dlsm_chain *h, const int64_t *zs, int S, int K, double *cooc_out) {
    NEED(h, h && zs, "null argument");
    NEED(h, S >= 1, "needs at least one sample");
    NEED(h, K >= 1 && K <= 256, "n_components must be in 1..256 (labels are kept as bytes)");
    HIPCHK(h, hipSetDevice(h->device));
    const int T = h->T, N = h->N;
    const size_t per = (size_t)T * N;
    for (size_t q = 0; q < (size_t)S * per; ++q)
        if (zs[q] < 0 || zs[q] >= K) FAIL(h, DLSM_E_DATA, "label out of range at %zu", q);
    int rc = dlsm_post_release(h); if (rc) return rc;
    const int Spad = (S + 63) / 64 * 64;
    const size_t n2 = (size_t)T * N * N;
    HIPCHK(h, hipMalloc((void **)&h->post_zt, per * Spad));
    HIPCHK(h, hipMalloc((void **)&h->post_cooc, n2 * sizeof(double)));
    HIPCHK(h, hipMemsetAsync(h->post_zt, 0, per * Spad, h->stream));
    h->post_S = S; h->post_Spad = Spad;
    // labels: staged in chunks of samples, packed to bytes and transposed on the device
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>(S, ((size_t)64 << 20) / (per * 8)));
    DevBuf stage, counts;
    HIPCHK(h, hipMalloc(&stage.p, (size_t)chunk * per * sizeof(int64_t)));
    HIPCHK(h, hipMalloc(&counts.p, n2 * sizeof(uint32_t)));
    {
        ProfScope ps(h, DLSM_K_LABELS);
        for (int s0 = 0; s0 < S; s0 += chunk) {
            const int ns = std::min(chunk, S - s0);
            HIPCHK(h, hipMemcpyAsync(stage.p, zs + (size_t)s0 * per, (size_t)ns * per * sizeof(int64_t),
                                     hipMemcpyHostToDevice, h->stream));
            const int nb = (int)std::min<size_t>(4096, ((size_t)ns * per + 255) / 256);
            hipLaunchKernelGGL(k_post_pack_labels<int64_t>, dim3(nb), dim3(256), 0, h->stream,
                               stage.as<int64_t>(), ns, s0, T, N, Spad, h->post_zt);
            HIPCHK(h, hipStreamSynchronize(h->stream));     // the staging buffer is reused
        }
        const int nt = (N + PC_TILE - 1) / PC_TILE;
        hipLaunchKernelGGL(k_post_cooccurrence, dim3(nt, nt, T), dim3(256), 0, h->stream,
                           h->post_zt, N, S, Spad, counts.as<uint32_t>());
        const int nb = (int)std::min<size_t>(8192, (n2 + 255) / 256);
        hipLaunchKernelGGL(k_post_counts_to_proba, dim3(nb), dim3(256), 0, h->stream,
                           counts.as<uint32_t>(), n2, (double)S, h->post_cooc);
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (cooc_out) return d2h(h, cooc_out, h->post_cooc, n2);
    return DLSM_OK;
}

int dlsm_post_expected_vi_sums(dlsm_chain *h, double *out) {
    NEED(h, h && out, "null argument");
    NEED(h, h->post_zt && h->post_cooc, "co-occurrences not computed (dlsm_post_cooccurrence)");
    HIPCHK(h, hipSetDevice(h->device));
    const int T = h->T, N = h->N, S = h->post_S, Spad = h->post_Spad;
    const int ngroups = (N + PV_ROWS_PER_WG - 1) / PV_ROWS_PER_WG;
    DevBuf part, res;
    HIPCHK(h, hipMalloc(&part.p, (size_t)T * ngroups * Spad * sizeof(double)));
    HIPCHK(h, hipMalloc(&res.p, (size_t)T * S * sizeof(double)));
    {
        ProfScope ps(h, DLSM_K_LABELS);
        hipLaunchKernelGGL(k_post_vi_rows, dim3(ngroups, Spad / 64, T), dim3(256), 0, h->stream,
                           h->post_zt, h->post_cooc, N, S, Spad, part.as<double>());
        hipLaunchKernelGGL(k_post_vi_reduce, dim3((S + 255) / 256, T), dim3(256), 0, h->stream,
                           part.as<double>(), ngroups, S, Spad, res.as<double>());
    }
    HIPCHK(h, hipGetLastError());
    return d2h(h, out, res.as<double>(), (size_t)T * S);
}

// ---- the same processing on the device-resident trace of dlsm_hdp_run ---------------------------
static int check_trace_rows(dlsm_chain *h, int first, int count) {
    NEED(h, h != nullptr, "null handle");
    NEED(h, h->htr_z && h->trace_X, "no device-resident HDP-LPCM trace (dlsm_hdp_trace_alloc)");
    NEED(h, first >= 0 && count >= 1 && first + count <= h->htr_n && first + count <= h->trace_n,
         "rows out of the trace");
    return DLSM_OK;
}

int dlsm_post_trace_label_counts(dlsm_chain *h, int first, int count, int32_t *nk) {
    int rc = check_trace_rows(h, first, count); if (rc) return rc;
    NEED(h, nk != nullptr, "null argument");
    NEED(h, h->htr_K <= 256, "n_components beyond 256");
    HIPCHK(h, hipSetDevice(h->device));
    const int T = h->T, N = h->N, K = h->htr_K;
    DevBuf out;
    const size_t n = (size_t)count * T * K;
    HIPCHK(h, hipMalloc(&out.p, n * sizeof(int32_t)));
    {
        ProfScope ps(h, DLSM_K_LABELS);
        hipLaunchKernelGGL(k_post_label_counts, dim3(T, count), dim3(256), 0, h->stream,
                           h->htr_z + (size_t)first * T * N, N, K, out.as<int32_t>());
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(nk, out.p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return DLSM_OK;
}

int dlsm_post_trace_cooccurrence(dlsm_chain *h, int first, int count, double *cooc_out,
                                 double *row_sums) {
    int rc = check_trace_rows(h, first, count); if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    const int T = h->T, N = h->N, S = count;
    const size_t per = (size_t)T * N;
    rc = dlsm_post_release(h); if (rc) return rc;
    const int Spad = (S + 63) / 64 * 64;
    const size_t n2 = (size_t)T * N * N;
    HIPCHK(h, hipMalloc((void **)&h->post_zt, per * Spad));
    HIPCHK(h, hipMalloc((void **)&h->post_cooc, n2 * sizeof(double)));
    HIPCHK(h, hipMemsetAsync(h->post_zt, 0, per * Spad, h->stream));
    h->post_S = S; h->post_Spad = Spad;
    DevBuf counts, sums;
    HIPCHK(h, hipMalloc(&counts.p, n2 * sizeof(uint32_t)));
    HIPCHK(h, hipMalloc(&sums.p, per * sizeof(double)));
    {
        ProfScope ps(h, DLSM_K_LABELS);
        const int nbp = (int)std::min<size_t>(8192, ((size_t)S * per + 255) / 256);
        hipLaunchKernelGGL(k_post_pack_labels<uint8_t>, dim3(nbp), dim3(256), 0, h->stream,
                           h->htr_z + (size_t)first * per, S, 0, T, N, Spad, h->post_zt);
        const int nt = (N + PC_TILE - 1) / PC_TILE;
        hipLaunchKernelGGL(k_post_cooccurrence, dim3(nt, nt, T), dim3(256), 0, h->stream,
                           h->post_zt, N, S, Spad, counts.as<uint32_t>());
        const int nb = (int)std::min<size_t>(8192, (n2 + 255) / 256);
        hipLaunchKernelGGL(k_post_counts_to_proba, dim3(nb), dim3(256), 0, h->stream,
                           counts.as<uint32_t>(), n2, (double)S, h->post_cooc);
        hipLaunchKernelGGL(k_post_row_sums, dim3((unsigned)((per + 3) / 4)), dim3(256), 0, h->stream,
                           h->post_cooc, per, N, sums.as<double>());
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (row_sums) { rc = d2h(h, row_sums, sums.as<double>(), per); if (rc) return rc; }
    if (cooc_out) return d2h(h, cooc_out, h->post_cooc, n2);
    return DLSM_OK;
}

int dlsm_post_get_cooccurrence(dlsm_chain *h, double *cooc_out) {
    NEED(h, h && cooc_out, "null argument");
    NEED(h, h->post_cooc, "co-occurrences not computed");
    HIPCHK(h, hipSetDevice(h->device));
    return d2h(h, cooc_out, h->post_cooc, (size_t)h->T * h->N * h->N);
}

int dlsm_post_trace_align(dlsm_chain *h, int first, int count, int ref_row) {
    int rc = check_trace_rows(h, first, count); if (rc) return rc;
    NEED(h, ref_row >= 0 && ref_row < h->trace_n, "reference row out of the trace");
    HIPCHK(h, hipSetDevice(h->device));
    const int T = h->T, N = h->N, D = h->D, K = h->htr_K;
    const size_t row = (size_t)T * N * D;
    if (!h->xref) { rc = dev_alloc(h, &h->xref, row); if (rc) return rc; }
    // the reference is one of the rows: aligned onto a copy of itself (its rotation is the identity)
    HIPCHK(h, hipMemcpyAsync(h->xref, h->trace_X + row * ref_row, row * sizeof(double),
                             hipMemcpyDeviceToDevice, h->stream));
    {
        ProfScope ps(h, DLSM_K_CENTER);
        DISPATCH_D(h, D, hipLaunchKernelGGL((k_post_align<DD>), dim3(count), dim3(256), 0, h->stream,
                                            h->trace_X + row * first, h->htr_mu + (size_t)first * K * D,
                                            h->xref, T * N, K));
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return DLSM_OK;
}

int dlsm_post_trace_mean(dlsm_chain *h, int first, int count, double *X_mean) {
    int rc = check_trace_rows(h, first, count); if (rc) return rc;
    NEED(h, X_mean != nullptr, "null argument");
    HIPCHK(h, hipSetDevice(h->device));
    const size_t row = (size_t)h->T * h->N * h->D;
    DevBuf part, out;
    HIPCHK(h, hipMalloc(&part.p, (size_t)PM_CHUNKS * row * sizeof(double)));
    HIPCHK(h, hipMalloc(&out.p, row * sizeof(double)));
    const unsigned nb = (unsigned)((row + 255) / 256);
    hipLaunchKernelGGL(k_post_mean_partial, dim3(nb, PM_CHUNKS), dim3(256), 0, h->stream,
                       h->trace_X + row * first, row, count, part.as<double>());
    hipLaunchKernelGGL(k_post_mean_final, dim3(nb), dim3(256), 0, h->stream, part.as<double>(), row,
                       count, out.as<double>());
    HIPCHK(h, hipGetLastError());
    return d2h(h, X_mean, out.as<double>(), row);
}

int dlsm_post_latent_marginal_loglik(dlsm_chain *h, int row, const double *init_w,
                                     const double *trans_w, const double *mu, const double *sigma,
                                     double lmbda, int K, double *out) {
    NEED(h, h && init_w && trans_w && mu && sigma && out, "null argument");
    NEED(h, K >= 1 && K <= 64, "1..64 components (a wavefront's lanes)");
    NEED(h, row < 0 || (h->trace_X && row < h->trace_n), "row out of the trace");
    NEED(h, row >= 0 || h->have_X, "positions not set");
    HIPCHK(h, hipSetDevice(h->device));
    const int T = h->T, N = h->N, D = h->D;
    const size_t n_w = (size_t)T * K * K, n_mu = (size_t)K * D;
    const int nwg = (N + 3) / 4;
    DevBuf par, part;
    HIPCHK(h, hipMalloc(&par.p, (K + n_w + n_mu + K) * sizeof(double)));
    HIPCHK(h, hipMalloc(&part.p, (size_t)nwg * sizeof(double)));
    double *d_init = par.as<double>(), *d_w = d_init + K, *d_mu = d_w + n_w, *d_sig = d_mu + n_mu;
    HIPCHK(h, hipMemcpyAsync(d_init, init_w, K * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d_w, trans_w, n_w * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d_mu, mu, n_mu * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d_sig, sigma, K * sizeof(double), hipMemcpyHostToDevice, h->stream));
    const double *X = row >= 0 ? h->trace_X + (size_t)T * N * D * row : h->X;
    DISPATCH_D(h, D, hipLaunchKernelGGL((k_post_forward_loglik<DD>), dim3(nwg), dim3(256), 0, h->stream,
                                        X, T, N, d_init, d_w, d_mu, d_sig, lmbda, K, part.as<double>()));
    HIPCHK(h, hipGetLastError());
    std::vector<double> hp(nwg);
    HIPCHK(h, hipStreamSynchronize(h->stream));     // the caller's parameter arrays are free again
    HIPCHK(h, hipMemcpy(hp.data(), part.p, nwg * sizeof(double), hipMemcpyDeviceToHost));
    double tot = 0.0;
    for (int g = 0; g < nwg; ++g) tot += hp[g];     // fixed order
    *out = tot;
    return DLSM_OK;
}

}  // extern "C"
