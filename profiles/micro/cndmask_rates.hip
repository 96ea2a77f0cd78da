// Issue cost of the float64 (and helper) VALU instructions the sweep's evaluator is made of,
// measured on one SIMD the way the evaluator uses it: 4 wavefronts per SIMD (1024-thread
// workgroups, one per CU), each running an unrolled block of the SAME instruction on
// independent registers.  Prints ns and shader cycles (at the clock measured beside it) per
// wavefront-instruction.
//   hipcc -O3 --offload-arch=gfx950 -o valu_rates profiles/micro/valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <string>

#define REP8(X) X X X X X X X X
#define BODY(NAME, ASM)                                                                          \
    __global__ __launch_bounds__(1024) void NAME(unsigned long long *out, int iters, double seed) { \
        double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, \
               a6 = seed + 6, a7 = seed + 7;                                                     \
        double b = 1.0000001, c = 0.5;                                                           \
        int ia = threadIdx.x, ib = 3;                                                            \
        unsigned long long t0, t1, c0, c1;                                                       \
        asm volatile("s_mov_b32 s20, 0x55555555\n\ts_mov_b32 s21, 0x0f0f0f0f" ::: "s20", "s21");            \
        asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(c0)); \
        for (int i = 0; i < iters; ++i) {                                                        \
            asm volatile(REP8(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), \
                         "+v"(a6), "+v"(a7), "+v"(ia) : "v"(b), "v"(c), "v"(ib) : "vcc", "s20", "s21", "s22", "s23");       \
        }                                                                                        \
        asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(c1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(ia)); \
        if ((threadIdx.x & 63) == 0) {                                                           \
            const size_t w = (size_t)blockIdx.x * 16 + (threadIdx.x >> 6);                       \
            out[4 * w] = t0; out[4 * w + 1] = t1; out[4 * w + 2] = c0; out[4 * w + 3] = c1;      \
        }                                                                                        \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 1.2345 && ia == 77) out[0] = 0;            \
    }

// v_cndmask_b32 priced at 19.4 cycles by valu_rates.hip (round 2): a dependent chain through vcc.
// Which part of that is the instruction?  Variants: the same chain; the VOP3 form with the lane mask in an SGPR pair (what the sweep's inverse-ballot masks compile
// to); the 64-bit select the evaluator actually executes (two cndmask: yb ? 1.0 : 0.0 has a zero
// low word, so one) followed by the fma that consumes it; and the exec-masked alternative
// (s_and_saveexec / v_add_f64 / s_mov exec).
BODY(k_cnd_dep, "v_cndmask_b32 %8, %8, %11, vcc\n v_cndmask_b32 %8, %8, %11, vcc\n v_cndmask_b32 %8, %8, %11, vcc\n v_cndmask_b32 %8, %8, %11, vcc\n v_cndmask_b32 %8, %8, %11, vcc\n v_cndmask_b32 %8, %8, %11, vcc\n v_cndmask_b32 %8, %8, %11, vcc\n v_cndmask_b32 %8, %8, %11, vcc\n")
BODY(k_cnd_sgpr, "v_cndmask_b32_e64 %8, %8, %11, s[20:21]\n v_cndmask_b32_e64 %8, %8, %11, s[20:21]\n v_cndmask_b32_e64 %8, %8, %11, s[20:21]\n v_cndmask_b32_e64 %8, %8, %11, s[20:21]\n v_cndmask_b32_e64 %8, %8, %11, s[20:21]\n v_cndmask_b32_e64 %8, %8, %11, s[20:21]\n v_cndmask_b32_e64 %8, %8, %11, s[20:21]\n v_cndmask_b32_e64 %8, %8, %11, s[20:21]\n")
BODY(k_cnd_fma, "v_cndmask_b32_e64 %8, 0, %11, s[20:21]\n v_fma_f64 %0, %0, %9, %10\n v_cndmask_b32_e64 %8, 0, %11, s[20:21]\n v_fma_f64 %1, %1, %9, %10\n v_cndmask_b32_e64 %8, 0, %11, s[20:21]\n v_fma_f64 %2, %2, %9, %10\n v_cndmask_b32_e64 %8, 0, %11, s[20:21]\n v_fma_f64 %3, %3, %9, %10\n")
BODY(k_fma4, "v_fma_f64 %0, %0, %9, %10\n v_fma_f64 %1, %1, %9, %10\n v_fma_f64 %2, %2, %9, %10\n v_fma_f64 %3, %3, %9, %10\n")
BODY(k_exec_add, "s_and_saveexec_b64 s[22:23], s[20:21]\n v_add_f64 %0, %0, %10\n s_mov_b64 exec, s[22:23]\n s_and_saveexec_b64 s[22:23], s[20:21]\n v_add_f64 %1, %1, %10\n s_mov_b64 exec, s[22:23]\n s_and_saveexec_b64 s[22:23], s[20:21]\n v_add_f64 %2, %2, %10\n s_mov_b64 exec, s[22:23]\n s_and_saveexec_b64 s[22:23], s[20:21]\n v_add_f64 %3, %3, %10\n s_mov_b64 exec, s[22:23]\n")


// the same harness with eight 32-bit chains (independent destinations)
#define IBODY(NAME, ASM)                                                                         \
    __global__ __launch_bounds__(1024) void NAME(unsigned long long *out, int iters, double seed) { \
        int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        int ib = (int)seed;                                                                      \
        unsigned long long t0, t1, c0, c1;                                                       \
        asm volatile("s_mov_b32 s20, 0x55555555\n\ts_mov_b32 s21, 0x0f0f0f0f\n\ts_mov_b64 vcc, s[20:21]" ::: "s20", "s21", "vcc"); \
        asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(c0)); \
        for (int i = 0; i < iters; ++i) {                                                        \
            asm volatile(REP8(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), \
                         "+v"(a6), "+v"(a7) : "v"(ib) : "s20", "s21");                            \
        }                                                                                        \
        asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(c1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7)); \
        if ((threadIdx.x & 63) == 0) {                                                           \
            const size_t w = (size_t)blockIdx.x * 16 + (threadIdx.x >> 6);                       \
            out[4 * w] = t0; out[4 * w + 1] = t1; out[4 * w + 2] = c0; out[4 * w + 3] = c1;      \
        }                                                                                        \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345) out[0] = 0;                          \
    }
IBODY(k_icnd_vcc_ind, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
IBODY(k_icnd_sgpr_ind, "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n")
IBODY(k_iadd_ind, "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")

typedef void (*kern_t)(unsigned long long *, int, double);

void run(const char *name, kern_t k, int waves_per_simd, int iters) {
    const int wg = 256;                       // one workgroup per CU
    const int threads = 256 * waves_per_simd; // 4 SIMDs x waves_per_simd x 64
    unsigned long long *d;
    hipMalloc(&d, sizeof(unsigned long long) * 4 * wg * 16);
    std::vector<unsigned long long> h(4 * wg * 16);
    std::vector<double> ns, cyc, ghz;
    for (int rep = 0; rep < 6; ++rep) {
        hipLaunchKernelGGL(k, dim3(wg), dim3(threads), 0, 0, d, iters, 1.5);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, sizeof(unsigned long long) * 4 * wg * 16, hipMemcpyDeviceToHost);
        if (rep < 2) continue;
        // per-SIMD rate: the waves of a SIMD share it; a workgroup's span is first entry -> last
        // exit (the arbiter favours the oldest wavefront, so the waves of a SIMD do not finish
        // together and a mean of per-wave spans would undercount)
        double span = 0, cspan = 0; int n = 0;
        for (int b = 0; b < wg; ++b) {
            unsigned long long t0 = ~0ull, t1 = 0, c0 = ~0ull, c1 = 0;
            for (int w = 0; w < threads / 64; ++w) {
                const size_t i = (size_t)b * 16 + w;
                t0 = std::min(t0, h[4 * i]); t1 = std::max(t1, h[4 * i + 1]);
                c0 = std::min(c0, h[4 * i + 2]); c1 = std::max(c1, h[4 * i + 3]);
            }
            span += (double)(t1 - t0) * 10.0; cspan += (double)(c1 - c0); ++n;
        }
        span /= n; cspan /= n;
        const double instr = (double)iters * 64.0 * waves_per_simd;    // per SIMD
        ns.push_back(span / instr); cyc.push_back(cspan / instr); ghz.push_back(cspan / span);
    }
    std::sort(ns.begin(), ns.end());
    printf("%-12s %d waves/SIMD : %.3f ns per wavefront-instruction, s_memtime ticks per instr %.2f (ticks/ns %.3f)\n",
           name, waves_per_simd, ns[ns.size() / 2], cyc[cyc.size() / 2], ghz[ghz.size() / 2]);
    hipFree(d);
}

int main() {
    const int it = 400;
    for (int w : {4, 1}) {
        run("cnd_dep_vcc x8", k_cnd_dep, w, it);
        run("cnd_indep_vcc x8", k_icnd_vcc_ind, w, it);
        run("cnd_indep_sgpr x8", k_icnd_sgpr_ind, w, it);
        run("add_u32 indep x8", k_iadd_ind, w, it);
        run("cnd_sgpr_mask x8", k_cnd_sgpr, w, it);
        run("fma x4 (per 8)", k_fma4, w, it);
        run("cnd+fma x4 (per 8)", k_cnd_fma, w, it);
        run("exec-masked add x4 (per 8)", k_exec_add, w, it);
    }
    return 0;
}
