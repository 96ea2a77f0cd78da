// Issue cost of the float64 instructions in the engine's sqrt / exp expansions, relative
// to v_fma_f64: one workgroup of 256 threads per CU, 8 independent chains per thread so
// that latency is hidden and the loop is issue bound.
// hipcc -O3 --offload-arch=gfx950 -o f64_rates f64_rates.cpp
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAINS 8
#define ITERS 4096

template <int OP>
__global__ __launch_bounds__(256) void k(double *out, double seed) {
    double v[CHAINS];
    for (int c = 0; c < CHAINS; ++c) v[c] = seed + 0.001 * (threadIdx.x + c);
    int e = (int)seed;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (OP == 0) v[c] = fma(v[c], 1.0000001, 1e-9);
            if (OP == 1) v[c] = __builtin_amdgcn_rsq(v[c]) + 1.5;
            if (OP == 2) v[c] = rint(v[c]) + 0.25;
            if (OP == 3) v[c] = (double)((int)v[c]) + 1.5;            // cvt_i32_f64 + cvt_f64_i32
            if (OP == 4) v[c] = __builtin_ldexp(v[c], e) ;
            if (OP == 5) v[c] = __builtin_amdgcn_rcp(v[c]) + 1.5;
            if (OP == 6) v[c] = v[c] * 1.0000001;
            if (OP == 7) v[c] = v[c] + 1e-9;
            if (OP == 8) v[c] = fmax(v[c], 1.0) ;
            if (OP == 9) v[c] = __builtin_amdgcn_sqrt(v[c]) + 1.5;
        }
    }
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += v[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
float run(double *d, const char *name, float base) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256 * 4), dim3(256), 0, 0, d, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256 * 4), dim3(256), 0, 0, d, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.3f ms  x%.2f of fma\n", name, ms, base > 0 ? ms / base : 1.0f);
    return ms;
}

int main() {
    double *d; hipMalloc(&d, 256 * 4 * 256 * 8);
    float b = run<0>(d, "v_fma_f64", 0);
    run<6>(d, "v_mul_f64", b);
    run<7>(d, "v_add_f64", b);
    run<8>(d, "v_max_f64", b);
    run<1>(d, "v_rsq_f64 (+add)", b);
    run<9>(d, "v_sqrt_f64 (+add)", b);
    run<5>(d, "v_rcp_f64 (+add)", b);
    run<2>(d, "v_rndne_f64 (+add)", b);
    run<3>(d, "cvt f64->i32->f64 (+add)", b);
    run<4>(d, "v_ldexp_f64", b);
    return 0;
}
