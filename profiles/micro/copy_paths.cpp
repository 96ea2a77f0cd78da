// Small host <-> device copies: pageable vs pinned host memory, synchronous cost per copy.
//   hipcc -O2 --offload-arch=gfx950 -o copy_paths copy_paths.cpp && ./copy_paths
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
int main() {
    const size_t sizes[] = {64, 4096, 32768, 163840, 327680};
    void *dev; hipMalloc(&dev, 1 << 20);
    void *pin; hipHostMalloc(&pin, 1 << 20, hipHostMallocDefault);
    void *pag = malloc(1 << 20); memset(pag, 1, 1 << 20);
    hipStream_t s; hipStreamCreate(&s);
    for (size_t n : sizes) {
        for (int dir = 0; dir < 2; ++dir)
            for (int kind = 0; kind < 3; ++kind) {
                const int reps = 300;
                auto t0 = std::chrono::steady_clock::now();
                for (int r = 0; r < reps; ++r) {
                    void *host = kind == 0 ? pag : pin;
                    if (kind == 2) { if (dir == 0) memcpy(pin, pag, n); }
                    if (dir == 0) hipMemcpyAsync(dev, host, n, hipMemcpyHostToDevice, s);
                    else hipMemcpyAsync(host, dev, n, hipMemcpyDeviceToHost, s);
                    hipStreamSynchronize(s);
                    if (kind == 2 && dir == 1) memcpy(pag, pin, n);
                }
                double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
                printf("%7zu B  %s  %-22s %7.2f us\n", n, dir == 0 ? "H2D" : "D2H",
                       kind == 0 ? "pageable" : kind == 1 ? "pinned" : "pageable via pinned", us);
            }
    }
    return 0;
}
