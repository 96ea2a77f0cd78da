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
            hipLaunchKernelGGL(k_post_pack_labels, dim3(nb), dim3(256), 0, h->stream,
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

}  // extern "C"
