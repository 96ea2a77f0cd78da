// Launch floor of an (almost) empty kernel as a function of grid, workgroup size and
// dynamic LDS: chains of dependent launches on one stream, wall time by HIP events.
// hipcc -O3 --offload-arch=gfx950 -o launch_floor launch_floor.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_empty(int *p) {
    extern __shared__ int lds[];
    if (p == nullptr && threadIdx.x == 12345) lds[0] = 1;
}

int main() {
    struct Cfg { int grid, block; size_t lds; };
    std::vector<Cfg> cfgs = {
        {251, 1024, 128 << 10}, {251, 1024, 64 << 10}, {251, 1024, 0}, {256, 1024, 0},
        {502, 512, 0}, {1004, 256, 0}, {2560, 256, 0}, {2560, 256, 8 << 10}, {5, 1024, 128 << 10},
        {251, 256, 0}, {64, 1024, 128 << 10}, {128, 1024, 128 << 10}, {4016, 64, 0},
        {256, 1024, 60 << 10}, {512, 1024, 60 << 10}};
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 << 10);
    int *d; hipMalloc(&d, 4);
    for (auto &c : cfgs) {
        for (int w = 0; w < 50; ++w) hipLaunchKernelGGL(k_empty, dim3(c.grid), dim3(c.block), c.lds, s, d);
        hipStreamSynchronize(s);
        const int n = 2000;
        hipEventRecord(e0, s);
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(c.grid), dim3(c.block), c.lds, s, d);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("grid %5d block %4d lds %6zu : %.2f us per launch\n", c.grid, c.block, c.lds, 1e3 * ms / n);
    }
    return 0;
}
