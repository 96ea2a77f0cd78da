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
        asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(c0)); \
        for (int i = 0; i < iters; ++i) {                                                        \
            asm volatile(REP8(ASM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), \
                         "+v"(a6), "+v"(a7), "+v"(ia) : "v"(b), "v"(c), "v"(ib) : "vcc");       \
        }                                                                                        \
        asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(c1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(ia)); \
        if ((threadIdx.x & 63) == 0) {                                                           \
            const size_t w = (size_t)blockIdx.x * 16 + (threadIdx.x >> 6);                       \
            out[4 * w] = t0; out[4 * w + 1] = t1; out[4 * w + 2] = c0; out[4 * w + 3] = c1;      \
        }                                                                                        \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 1.2345 && ia == 77) out[0] = 0;            \
    }

// each ASM string: 8 independent instructions (one per chain)
BODY(k_fma, "v_fma_f64 %0, %0, %9, %10\n v_fma_f64 %1, %1, %9, %10\n v_fma_f64 %2, %2, %9, %10\n v_fma_f64 %3, %3, %9, %10\n v_fma_f64 %4, %4, %9, %10\n v_fma_f64 %5, %5, %9, %10\n v_fma_f64 %6, %6, %9, %10\n v_fma_f64 %7, %7, %9, %10\n")
BODY(k_fma_dep, "v_fma_f64 %0, %0, %9, %10\n v_fma_f64 %0, %0, %9, %10\n v_fma_f64 %0, %0, %9, %10\n v_fma_f64 %0, %0, %9, %10\n v_fma_f64 %0, %0, %9, %10\n v_fma_f64 %0, %0, %9, %10\n v_fma_f64 %0, %0, %9, %10\n v_fma_f64 %0, %0, %9, %10\n")
BODY(k_mul, "v_mul_f64 %0, %0, %9\n v_mul_f64 %1, %1, %9\n v_mul_f64 %2, %2, %9\n v_mul_f64 %3, %3, %9\n v_mul_f64 %4, %4, %9\n v_mul_f64 %5, %5, %9\n v_mul_f64 %6, %6, %9\n v_mul_f64 %7, %7, %9\n")
BODY(k_add, "v_add_f64 %0, %0, %10\n v_add_f64 %1, %1, %10\n v_add_f64 %2, %2, %10\n v_add_f64 %3, %3, %10\n v_add_f64 %4, %4, %10\n v_add_f64 %5, %5, %10\n v_add_f64 %6, %6, %10\n v_add_f64 %7, %7, %10\n")
BODY(k_rsq, "v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n v_rsq_f64 %4, %4\n v_rsq_f64 %5, %5\n v_rsq_f64 %6, %6\n v_rsq_f64 %7, %7\n")
BODY(k_rcp, "v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7\n")
BODY(k_sqrt, "v_sqrt_f64 %0, %0\n v_sqrt_f64 %1, %1\n v_sqrt_f64 %2, %2\n v_sqrt_f64 %3, %3\n v_sqrt_f64 %4, %4\n v_sqrt_f64 %5, %5\n v_sqrt_f64 %6, %6\n v_sqrt_f64 %7, %7\n")
BODY(k_rndne, "v_rndne_f64 %0, %0\n v_rndne_f64 %1, %1\n v_rndne_f64 %2, %2\n v_rndne_f64 %3, %3\n v_rndne_f64 %4, %4\n v_rndne_f64 %5, %5\n v_rndne_f64 %6, %6\n v_rndne_f64 %7, %7\n")
BODY(k_ldexp, "v_ldexp_f64 %0, %0, %11\n v_ldexp_f64 %1, %1, %11\n v_ldexp_f64 %2, %2, %11\n v_ldexp_f64 %3, %3, %11\n v_ldexp_f64 %4, %4, %11\n v_ldexp_f64 %5, %5, %11\n v_ldexp_f64 %6, %6, %11\n v_ldexp_f64 %7, %7, %11\n")
BODY(k_cvt, "v_cvt_i32_f64 %8, %0\n v_cvt_i32_f64 %8, %1\n v_cvt_i32_f64 %8, %2\n v_cvt_i32_f64 %8, %3\n v_cvt_i32_f64 %8, %4\n v_cvt_i32_f64 %8, %5\n v_cvt_i32_f64 %8, %6\n v_cvt_i32_f64 %8, %7\n")
BODY(k_cndmask, "v_cndmask_b32 %8, %8, %11, vcc\n v_cndmask_b32 %8, %8, %11, vcc\n v_cndmask_b32 %8, %8, %11, vcc\n v_cndmask_b32 %8, %8, %11, vcc\n v_cndmask_b32 %8, %8, %11, vcc\n v_cndmask_b32 %8, %8, %11, vcc\n v_cndmask_b32 %8, %8, %11, vcc\n v_cndmask_b32 %8, %8, %11, vcc\n")
BODY(k_addu32, "v_add_u32 %8, %8, %11\n v_add_u32 %8, %8, %11\n v_add_u32 %8, %8, %11\n v_add_u32 %8, %8, %11\n v_add_u32 %8, %8, %11\n v_add_u32 %8, %8, %11\n v_add_u32 %8, %8, %11\n v_add_u32 %8, %8, %11\n")
BODY(k_lshladd, "v_lshl_add_u32 %8, %8, 20, %11\n v_lshl_add_u32 %8, %8, 20, %11\n v_lshl_add_u32 %8, %8, 20, %11\n v_lshl_add_u32 %8, %8, 20, %11\n v_lshl_add_u32 %8, %8, 20, %11\n v_lshl_add_u32 %8, %8, 20, %11\n v_lshl_add_u32 %8, %8, 20, %11\n v_lshl_add_u32 %8, %8, 20, %11\n")
BODY(k_mullo, "v_mul_lo_u32 %8, %8, %11\n v_mul_lo_u32 %8, %8, %11\n v_mul_lo_u32 %8, %8, %11\n v_mul_lo_u32 %8, %8, %11\n v_mul_lo_u32 %8, %8, %11\n v_mul_lo_u32 %8, %8, %11\n v_mul_lo_u32 %8, %8, %11\n v_mul_lo_u32 %8, %8, %11\n")
BODY(k_movdpp, "v_mov_b32_dpp %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
BODY(k_fmaf32, "v_fma_f32 %8, %8, %11, %11\n v_fma_f32 %8, %8, %11, %11\n v_fma_f32 %8, %8, %11, %11\n v_fma_f32 %8, %8, %11, %11\n v_fma_f32 %8, %8, %11, %11\n v_fma_f32 %8, %8, %11, %11\n v_fma_f32 %8, %8, %11, %11\n v_fma_f32 %8, %8, %11, %11\n")

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
        run("fma_f64", k_fma, w, it);
        run("fma_f64_dep", k_fma_dep, w, it);
        run("mul_f64", k_mul, w, it);
        run("add_f64", k_add, w, it);
        run("rsq_f64", k_rsq, w, it);
        run("rcp_f64", k_rcp, w, it);
        run("sqrt_f64", k_sqrt, w, it);
        run("rndne_f64", k_rndne, w, it);
        run("ldexp_f64", k_ldexp, w, it);
        run("cvt_i32_f64", k_cvt, w, it);
        run("cndmask_b32", k_cndmask, w, it);
        run("add_u32", k_addu32, w, it);
        run("lshl_add_u32", k_lshladd, w, it);
        run("mul_lo_u32", k_mullo, w, it);
        run("mov_dpp", k_movdpp, w, it);
        run("fma_f32", k_fmaf32, w, it);
    }
    return 0;
}
