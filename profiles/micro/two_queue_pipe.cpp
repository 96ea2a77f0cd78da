// Feasibility of overlapping consecutive evaluator launches: dummy kernels that
// spin for a set time, scheduled (a) as one fused launch per batch on one stream - today's
// pipelined sweep - and (b) with the resolver and evaluator roles as separate kernels, the
// evaluators alternating between two streams, tied by events:
//   resolve(l) waits for eval(l - 1); eval(l) waits for resolve(l - 1).
//   hipcc -O2 --offload-arch=gfx950 -o two_queue_pipe two_queue_pipe.cpp && ./two_queue_pipe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(1024) void k_spin(int n_fast, int ticks_fast, int ticks_slow) {
    // blocks [0, n_fast) spin ticks_fast, the others ticks_slow (100 MHz wall clock ticks)
    const unsigned long long t0 = wall_clock64();
    const int ticks = (int)blockIdx.x < n_fast ? ticks_fast : ticks_slow;
    while ((long long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(1);
}

int main() {
    const int T = 10, NE = 246, NB = 18, REPS = 200;
    const int tR = 300, tE = 750;            // 3.0 us resolver, 7.5 us evaluator
    hipStream_t sR, sE[2];
    hipStreamCreateWithFlags(&sR, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sE[0], hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sE[1], hipStreamNonBlocking);
    std::vector<hipEvent_t> eR(NB + 3), eE(NB + 3);
    for (auto &e : eR) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    for (auto &e : eE) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    auto now = [] { return std::chrono::steady_clock::now(); };
    // (a) fused: one launch per batch, T resolver blocks + NE evaluator blocks
    for (int w = 0; w < 2; ++w) {
        auto t0 = now();
        for (int r = 0; r < REPS; ++r)
            for (int l = 0; l < NB; ++l)
                hipLaunchKernelGGL(k_spin, dim3(T + NE), dim3(1024), 0, sR, T, tR, tE);
        hipStreamSynchronize(sR);
        double us = std::chrono::duration<double, std::micro>(now() - t0).count() / (REPS * NB);
        if (w) printf("fused, one stream          : %.2f us per batch\n", us);
    }
    // (b) split roles, evaluators alternating between two streams
    for (int w = 0; w < 2; ++w) {
        auto t0 = now();
        for (int r = 0; r < REPS; ++r) {
            for (int l = 0; l < NB; ++l) {
                hipStream_t se = sE[l & 1];
                if (l > 0) hipStreamWaitEvent(se, eR[l - 1], 0);
                hipLaunchKernelGGL(k_spin, dim3(NE), dim3(1024), 0, se, 0, tR, tE);
                hipEventRecord(eE[l], se);
                if (l > 0) hipStreamWaitEvent(sR, eE[l - 1], 0);
                hipLaunchKernelGGL(k_spin, dim3(T), dim3(1024), 0, sR, T, tR, tE);
                hipEventRecord(eR[l], sR);
            }
            hipStreamWaitEvent(sR, eE[NB - 1], 0);
            // next repetition's first evaluators must follow this one's last resolver
            hipEventRecord(eR[NB], sR);
            hipStreamWaitEvent(sE[0], eR[NB], 0);
            hipStreamWaitEvent(sE[1], eR[NB], 0);
        }
        hipStreamSynchronize(sR); hipStreamSynchronize(sE[0]); hipStreamSynchronize(sE[1]);
        double us = std::chrono::duration<double, std::micro>(now() - t0).count() / (REPS * NB);
        if (w) printf("split roles, three streams : %.2f us per batch\n", us);
    }
    // host cost of (b)'s calls alone: the same sequence with empty kernels
    {
        auto t0 = now();
        for (int r = 0; r < REPS; ++r)
            for (int l = 0; l < NB; ++l) {
                hipStream_t se = sE[l & 1];
                if (l > 0) hipStreamWaitEvent(se, eR[l - 1], 0);
                hipLaunchKernelGGL(k_spin, dim3(NE), dim3(1024), 0, se, 0, 0, 0);
                hipEventRecord(eE[l], se);
                if (l > 0) hipStreamWaitEvent(sR, eE[l - 1], 0);
                hipLaunchKernelGGL(k_spin, dim3(T), dim3(1024), 0, sR, T, 0, 0);
                hipEventRecord(eR[l], sR);
            }
        double us_host = std::chrono::duration<double, std::micro>(now() - t0).count() / (REPS * NB);
        hipStreamSynchronize(sR); hipStreamSynchronize(sE[0]); hipStreamSynchronize(sE[1]);
        double us = std::chrono::duration<double, std::micro>(now() - t0).count() / (REPS * NB);
        printf("split roles, empty kernels : %.2f us per batch to enqueue, %.2f us to finish\n", us_host, us);
    }
    // (c) the split form captured once as a graph with parallel branches, replayed
    {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(sR, hipStreamCaptureModeGlobal);
        hipEventRecord(eR[NB + 1], sR);                      // fork
        hipStreamWaitEvent(sE[0], eR[NB + 1], 0);
        hipStreamWaitEvent(sE[1], eR[NB + 1], 0);
        for (int l = 0; l < NB; ++l) {
            hipStream_t se = sE[l & 1];
            if (l > 0) hipStreamWaitEvent(se, eR[l - 1], 0);
            hipLaunchKernelGGL(k_spin, dim3(NE), dim3(1024), 0, se, 0, tR, tE);
            hipEventRecord(eE[l], se);
            if (l > 0) hipStreamWaitEvent(sR, eE[l - 1], 0);
            hipLaunchKernelGGL(k_spin, dim3(T), dim3(1024), 0, sR, T, tR, tE);
            hipEventRecord(eR[l], sR);
        }
        hipStreamWaitEvent(sR, eE[NB - 1], 0);               // join
        hipStreamWaitEvent(sR, eE[NB - 2], 0);
        hipError_t e1 = hipStreamEndCapture(sR, &g);
        hipError_t e2 = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        if (e1 != hipSuccess || e2 != hipSuccess) { printf("graph capture failed: %d %d\n", e1, e2); return 0; }
        for (int w = 0; w < 2; ++w) {
            auto t0 = now();
            for (int r = 0; r < REPS; ++r) hipGraphLaunch(ge, sR);
            hipStreamSynchronize(sR);
            double us = std::chrono::duration<double, std::micro>(now() - t0).count() / (REPS * NB);
            if (w) printf("split roles, graph replay  : %.2f us per batch\n", us);
        }
        // and the fused form as a (linear) graph
        hipGraph_t g2; hipGraphExec_t ge2;
        hipStreamBeginCapture(sR, hipStreamCaptureModeGlobal);
        for (int l = 0; l < NB; ++l)
            hipLaunchKernelGGL(k_spin, dim3(T + NE), dim3(1024), 0, sR, T, tR, tE);
        hipStreamEndCapture(sR, &g2);
        hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0);
        for (int w = 0; w < 2; ++w) {
            auto t0 = now();
            for (int r = 0; r < REPS; ++r) hipGraphLaunch(ge2, sR);
            hipStreamSynchronize(sR);
            double us = std::chrono::duration<double, std::micro>(now() - t0).count() / (REPS * NB);
            if (w) printf("fused, graph replay        : %.2f us per batch\n", us);
        }
    }
    return 0;
}
