// C-ABI of the one-step-ahead forecasts (included by capi.hip after capi_init.hpp).
#pragma once

extern "C" {

int dlsm_forecast_mean_probas(dlsm_chain *h, const double *Xs, const double *intercepts, int S,
                              int zero_diag, double *out) {
    NEED(h, h && Xs && intercepts && out, "null argument");
    NEED(h, S >= 1, "needs at least one sample");
    HIPCHK(h, hipSetDevice(h->device));
    const int N = h->N, D = h->D;
    DevBuf bX, bB, bO;
    HIPCHK(h, hipMalloc(&bX.p, (size_t)S * N * D * sizeof(double)));
    HIPCHK(h, hipMalloc(&bB.p, (size_t)S * sizeof(double)));
    HIPCHK(h, hipMalloc(&bO.p, (size_t)N * N * sizeof(double)));
    HIPCHK(h, hipMemcpyAsync(bX.p, Xs, (size_t)S * N * D * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(bB.p, intercepts, (size_t)S * sizeof(double), hipMemcpyHostToDevice, h->stream));
    const int nt = (N + FC_TILE - 1) / FC_TILE;
    {
        ProfScope ps(h, DLSM_K_LOGLIK);
        DISPATCH_D(h, D, hipLaunchKernelGGL((k_forecast_mean<DD>), dim3(nt, nt), dim3(256), 0, h->stream,
                                            bX.as<double>(), bB.as<double>(), S, N, zero_diag,
                                            bO.as<double>()));
    }
    HIPCHK(h, hipGetLastError());
    return d2h(h, out, bO.as<double>(), (size_t)N * N);
}

int dlsm_forecast_marginal(dlsm_chain *h, const double *x, const double *W, const double *intercepts,
                           int S, double *out) {
    NEED(h, h && x && W && intercepts && out, "null argument");
    NEED(h, S >= 1, "needs at least one sample");
    HIPCHK(h, hipSetDevice(h->device));
    const int N = h->N, D = h->D;
    DevBuf bX, bW, bB, bO;
    HIPCHK(h, hipMalloc(&bX.p, (size_t)N * D * sizeof(double)));
    HIPCHK(h, hipMalloc(&bW.p, (size_t)S * N * sizeof(double)));
    HIPCHK(h, hipMalloc(&bB.p, (size_t)S * sizeof(double)));
    HIPCHK(h, hipMalloc(&bO.p, (size_t)N * N * sizeof(double)));
    HIPCHK(h, hipMemcpyAsync(bX.p, x, (size_t)N * D * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(bW.p, W, (size_t)S * N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(bB.p, intercepts, (size_t)S * sizeof(double), hipMemcpyHostToDevice, h->stream));
    const int nt = (N + FC_TILE - 1) / FC_TILE;
    {
        ProfScope ps(h, DLSM_K_LOGLIK);
        DISPATCH_D(h, D, hipLaunchKernelGGL((k_forecast_marginal<DD>), dim3(nt, nt), dim3(256), 0,
                                            h->stream, bX.as<double>(), bW.as<double>(),
                                            bB.as<double>(), S, N, bO.as<double>()));
    }
    HIPCHK(h, hipGetLastError());
    return d2h(h, out, bO.as<double>(), (size_t)N * N);
}

}  // extern "C"
