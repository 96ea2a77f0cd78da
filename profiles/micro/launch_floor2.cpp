// What makes a dependent launch of 256 x 1024 threads cost more than the empty-kernel floor:
// VGPR allocation, kernel-argument size, or an early exit behind an argument load?
// hipcc -O3 --offload-arch=gfx950 -o launch_floor2 launch_floor2.cpp
#include <hip/hip_runtime.h>
#include <cstdio>

struct Big { double a[36]; int b[8]; };

__global__ __launch_bounds__(1024) void k_small(int *p) {
    if (p == nullptr) p[0] = 1;
}
__global__ __launch_bounds__(1024) void k_bigarg(Big g, int *p) {
    if (g.b[7] == 12345) p[0] = (int)g.a[3];
}
// ~120 live VGPRs on a path that is never taken at run time
__global__ __launch_bounds__(1024) void k_vgpr(int *p, int flag) {
    if (flag) {
        double v[56];
        for (int i = 0; i < 56; ++i) v[i] = p[i] * 1.5;
        for (int it = 0; it < flag; ++it)
            for (int i = 0; i < 56; ++i) v[i] = v[i] * v[(i + 7) % 56] + 1.0;
        double s = 0; for (int i = 0; i < 56; ++i) s += v[i];
        p[0] = (int)s;
    }
}
__global__ __launch_bounds__(256) void k_vgpr256(int *p, int flag) {
    if (flag) {
        double v[56];
        for (int i = 0; i < 56; ++i) v[i] = p[i] * 1.5;
        for (int it = 0; it < flag; ++it)
            for (int i = 0; i < 56; ++i) v[i] = v[i] * v[(i + 7) % 56] + 1.0;
        double s = 0; for (int i = 0; i < 56; ++i) s += v[i];
        p[0] = (int)s;
    }
}

template <typename F> void timeit(const char *name, F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) launch();
    hipDeviceSynchronize();
    const int n = 2000;
    hipEventRecord(e0, 0);
    for (int i = 0; i < n; ++i) launch();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.2f us per launch\n", name, 1e3 * ms / n);
}

int main() {
    int *d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
    Big g{}; 
    const size_t lds = 128 << 10;
    hipFuncSetAttribute((const void *)k_small, hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
    hipFuncSetAttribute((const void *)k_vgpr, hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
    timeit("256 x 1024, few VGPRs, 8 B args", [&] { hipLaunchKernelGGL(k_small, dim3(256), dim3(1024), lds, 0, d); });
    timeit("256 x 1024, few VGPRs, 328 B args", [&] { hipLaunchKernelGGL(k_bigarg, dim3(256), dim3(1024), 0, 0, g, d); });
    timeit("256 x 1024, ~120 VGPRs allocated", [&] { hipLaunchKernelGGL(k_vgpr, dim3(256), dim3(1024), lds, 0, d, 0); });
    timeit("1024 x 256, ~120 VGPRs allocated", [&] { hipLaunchKernelGGL(k_vgpr256, dim3(1024), dim3(256), 0, 0, d, 0); });
    timeit("10 x 1024, ~120 VGPRs allocated", [&] { hipLaunchKernelGGL(k_vgpr, dim3(10), dim3(1024), lds, 0, d, 0); });
    return 0;
}
