// Device-side building blocks shared by every kernel of the engine:
// Philox4x32-10 counter RNG, wave/workgroup reductions for 64-wide wavefronts,
// and the per-dyad log-likelihood algebra.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "exp2_table.hpp"

namespace dlsm {

// ---- RNG streams (counter word 3, low byte); shared with the CPU oracle ----
enum : uint32_t {
    STREAM_SWEEP_NORMAL = 0,
    STREAM_SWEEP_UNIFORM = 1,
    STREAM_INTERCEPT = 2,
    STREAM_LABELS = 3,
    STREAM_CONTROLS = 4
};

struct U4 { uint32_t x, y, z, w; };

// Iteration index of a launch: a plain value, or (captured graphs, where kernel
// arguments are frozen) a counter that lives in device memory.
struct IterRef {
    uint32_t value;
    const uint32_t *ptr;
    __device__ __forceinline__ uint32_t get() const { return ptr ? *ptr : value; }
};

__host__ __device__ __forceinline__ U4 philox4x32_10(uint64_t seed, uint32_t c0,
                                                     uint32_t c1, uint32_t c2,
                                                     uint32_t c3) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

// 53 random bits -> (0, 1]; every value exactly representable
__host__ __device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {
    double k = (double)(hi >> 5) * 67108864.0 + (double)(lo >> 6);
    return (k + 1.0) * (1.0 / 9007199254740992.0);
}

__host__ __device__ __forceinline__ uint32_t stream_word(uint32_t chain,
                                                         uint32_t stream) {
    return (chain << 8) | stream;
}

__device__ __forceinline__ void philox_uniform2(uint64_t seed, uint32_t c0,
                                                uint32_t c1, uint32_t c2,
                                                uint32_t c3, double &u0, double &u1) {
    U4 r = philox4x32_10(seed, c0, c1, c2, c3);
    u0 = u53(r.x, r.y);
    u1 = u53(r.z, r.w);
}

__device__ __forceinline__ void box_muller(double u0, double u1, double &z0,
                                           double &z1) {
    double r = sqrt(-2.0 * log(u0));
    double a = 6.283185307179586476925286766559 * u1;
    double s, c;
    sincos(a, &s, &c);
    z0 = r * c;
    z1 = r * s;
}

// latent dimensions: every kernel is a template of n_features; 1..DLSM_D_MAX are built (the reference's examples
// and the paper use 2; k_pipe_step's register plan is tuned for 1..4 and spills above)
constexpr int DLSM_D_MAX = 8;
constexpr int DLSM_D_PIPE_MAX = 8;        // k_pipe_step (algo 4)
constexpr int DLSM_D_CCPIPE_MAX = 8;      // k_ccpipe_step (algo 5)

// ---- cross-workgroup hand-offs inside one launch -----------------------------------------
// A CU's vector L1 is never refreshed by another CU's stores, and a kernel boundary is the only
// implicit write-back / invalidate.  Bytes that one workgroup of a persistent launch hands to
// another are therefore stored write-through and loaded past the L1 (`sc1` on gfx950: agent
// scope), every one of them, on both sides; the flag or counter that announces them is written
// after every storing wavefront has drained its stores (s_waitcnt vmcnt(0)) and read by a
// relaxed sc1 poll (MI355X guide, "inter-workgroup visibility": the sc1 / sc1 hand-off).
// COH = false gives the plain accesses of the launch-per-batch kernels.
typedef unsigned int dlsm_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int dlsm_u2 __attribute__((ext_vector_type(2)));
constexpr int DLSM_AUX_SC1 = 16;                 // cache-policy bit of the buffer builtins
__device__ __forceinline__ __amdgpu_buffer_rsrc_t coh_rsrc(const void *base) {
    // wave-uniform base (forced into scalar registers), byte offsets from the lanes
    const uint64_t a = (uint64_t)base;
    const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    return __builtin_amdgcn_make_buffer_rsrc((void *)u, 0, 0x7fffffff, 0x00020000);
}
template <bool COH>
__device__ __forceinline__ double coh_load(const double *p) {
    if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool COH>
__device__ __forceinline__ void coh_store(double *p, double v) {
    if (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
__device__ __forceinline__ int32_t coh_load_i32(const int32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void coh_store_i32(int32_t *p, int32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 16 bytes at base + off (base wave-uniform, off a multiple of 16)
// (`soff`: a wave-uniform addend that travels in a scalar register, so that the loads of an
// unrolled loop share ONE per-lane offset register)
template <bool COH>
__device__ __forceinline__ double2 coh_load2(const void *base, uint32_t off, uint32_t soff = 0u) {
    if (COH) {
        const dlsm_u4 v = __builtin_amdgcn_raw_buffer_load_b128(coh_rsrc(base), (int)off, (int)soff, DLSM_AUX_SC1);
        return make_double2(__hiloint2double((int)v.y, (int)v.x), __hiloint2double((int)v.w, (int)v.z));
    }
    return *(const double2 *)((const char *)base + off + soff);
}
template <bool COH>
__device__ __forceinline__ void coh_store2(void *base, uint32_t off, double2 v) {
    if (COH) {
        dlsm_u4 w;
        w.x = (unsigned)__double2loint(v.x); w.y = (unsigned)__double2hiint(v.x);
        w.z = (unsigned)__double2loint(v.y); w.w = (unsigned)__double2hiint(v.y);
        __builtin_amdgcn_raw_buffer_store_b128(w, coh_rsrc(base), (int)off, 0, DLSM_AUX_SC1);
    } else {
        *(double2 *)((char *)base + off) = v;
    }
}
// D doubles of one row at base + off (one 16-byte access per pair)
template <int D, bool COH>
__device__ __forceinline__ void coh_load_row(const void *base, uint32_t off, double *out) {
    if (!COH) {
        const double *src = (const double *)((const char *)base + off);
#pragma unroll
        for (int d = 0; d < D; ++d) out[d] = src[d];
        return;
    }
    const __amdgpu_buffer_rsrc_t r = coh_rsrc(base);
    if (D % 2 == 0) {                   // rows of an even D are 16-byte aligned
#pragma unroll
        for (int d = 0; d + 1 < D; d += 2) {
            const dlsm_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(off + 8u * d), 0, DLSM_AUX_SC1);
            out[d] = __hiloint2double((int)v.y, (int)v.x);
            out[d + 1] = __hiloint2double((int)v.w, (int)v.z);
        }
    } else {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const dlsm_u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(off + 8u * d), 0, DLSM_AUX_SC1);
            out[d] = __hiloint2double((int)v.y, (int)v.x);
        }
    }
}
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- reductions ------------------------------------------------------------
// All-lanes reductions of a 64-wide wavefront without LDS traffic: four DPP steps inside
// each row of 16 lanes (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: after them
// every lane holds its row's total), then the four row totals through v_readlane.
// (every lane of these patterns has a valid source, so the move needs no "old" value: the
// update_dpp form cost an extra v_mov per word to provide one)
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_value(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum_all(double v) {
    v += dpp_move<0xB1>(v);        // quad_perm [1, 0, 3, 2]
    v += dpp_move<0x4E>(v);        // quad_perm [2, 3, 0, 1]
    v += dpp_move<0x141>(v);       // row_half_mirror
    v += dpp_move<0x140>(v);       // row_mirror
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
// The same sum, bit for bit (butterflies commute and the rows combine in the same order), with
// the rows exchanged through the LDS crossbar (ds_bpermute) instead of eight v_readlane: 14
// vector instructions against 27, for kernels bound by VALU issue; a lone wavefront waiting on
// its own result is better served by wave_sum_all (two ds_bpermute round trips are slower than
// the readlanes).
__device__ __forceinline__ double wave_sum_all_tp(double v, int lane) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    const int a16 = (lane ^ 16) << 2, a32 = (lane ^ 32) << 2;
    v += __hiloint2double(__builtin_amdgcn_ds_bpermute(a16, __double2hiint(v)),
                          __builtin_amdgcn_ds_bpermute(a16, __double2loint(v)));
    v += __hiloint2double(__builtin_amdgcn_ds_bpermute(a32, __double2hiint(v)),
                          __builtin_amdgcn_ds_bpermute(a32, __double2loint(v)));
    return v;
}
__device__ __forceinline__ double wave_prod_all(double v) {
    v *= dpp_move<0xB1>(v);
    v *= dpp_move<0x4E>(v);
    v *= dpp_move<0x141>(v);
    v *= dpp_move<0x140>(v);
    return (lane_value(v, 0) * lane_value(v, 16)) * (lane_value(v, 32) * lane_value(v, 48));
}

// Sum over the workgroup, result in every thread.  `buf` holds NW doubles of LDS
// that nobody rewrites before every wave has passed the barrier below (callers
// that loop alternate between two buffers).
template <int NW>
__device__ __forceinline__ double block_sum_all(double v, double *buf, int tid) {
    v = wave_sum_all(v);
    if ((tid & 63) == 0) buf[tid >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += buf[w];
    return s;
}

// ---- geometry --------------------------------------------------------------
template <int D>
__device__ __forceinline__ double dist_of(const double *a, const double *b,
                                          int squared) {
    double s = 0.0;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        double df = a[d] - b[d];
        s += df * df;
    }
    return squared ? s : sqrt(s);
}


// ---- lean float64 sqrt / exp for the hot loops ---------------------------------
// Same algorithms as the compiler's expansions (Goldschmidt refinement of
// v_rsq_f64; argument reduction + polynomial + ldexp) without the subnormal /
// huge-argument rescaling: squared distances between latent positions and
// exponents in [-745, 709] never need it.  Both are accurate to ~1 ulp.
// Measured on MI355X (profiles/micro/): v_rsq_f64 / v_sqrt_f64 are good to ~2^-25.6 and
// issue at a quarter of the fma rate; one Goldschmidt iteration + one residual
// correction already gives the correctly rounded root on 2^20 samples over
// [2e-9, 2e4], also with the unrefined h = rsq / 2 in the correction (the compiler's
// expansion refines h and spends a second correction).
// s == 0 (coincident points) would give rsq = inf: adding the smallest normal number
// leaves every s > 1e-292 unchanged and turns 0 into a root of 1.5e-154, which no sum or
// exponential downstream can tell from 0 (one add instead of a compare and two selects;
// dist_fast folds the add into its first fma)
constexpr double SQRT_GUARD = 2.2250738585072014e-308;
__device__ __forceinline__ double fast_sqrt_guarded(double s) {      // s >= SQRT_GUARD
    const double y = __builtin_amdgcn_rsq(s);
    double g = s * y;
    const double h = 0.5 * y;             // ~1 / (2 sqrt(s)): good enough for the correction
#ifdef DLSM_EXACT_SQRT
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
#endif
    // the residual correction alone (no Goldschmidt step before it): 35 ulp at most, 1.06 ulp on
    // average (profiles/r03_sqrt_acc.txt) - 8e-15 relative on a distance, against the 1e-6 the
    // path asks for - at 5 instructions instead of 7
    const double e = fma(-g, g, s);
    return fma(e, h, g);
}
__device__ __forceinline__ double fast_sqrt(double s) { return fast_sqrt_guarded(s + SQRT_GUARD); }

// 1 / r for a normal, positive r (radii): v_rcp_f64 (good to 2^-25.6) + two Newton steps,
// within an ulp of the division at a fifth of its instructions
__device__ __forceinline__ double fast_rcp(double r) {
    double x = __builtin_amdgcn_rcp(r);
    x = fma(fma(-r, x, 1.0), x, x);
    return fma(fma(-r, x, 1.0), x, x);
}

__device__ __forceinline__ double fast_exp(double x) {
    const double k = rint(x * 1.4426950408889634074);
    double r = fma(k, -6.93147180369123816490e-01, x);
    r = fma(k, -1.90821492927058770002e-10, r);      // |r| <= ln2 / 2
    // degree-11 interpolant of e^r at the Chebyshev nodes of |r| <= ln2 / 2 (computed in
    // 80-digit arithmetic, rounded to double): max relative error 1.7e-17 in exact
    // arithmetic, two fma fewer than the Taylor polynomial of the same accuracy
    double p = 0x1.af631d0059becp-26;
    p = fma(p, r, 0x1.28b4057f44145p-22);
    p = fma(p, r, 0x1.71ddf5749d126p-19);
    p = fma(p, r, 0x1.a01991ac8730ap-16);
    p = fma(p, r, 0x1.a01a01b14378fp-13);
    p = fma(p, r, 0x1.6c16c187fbe02p-10);
    p = fma(p, r, 0x1.111111110f225p-7);
    p = fma(p, r, 0x1.555555554f0cfp-5);
    p = fma(p, r, 0x1.555555555555ap-3);
    p = fma(p, r, 0x1.0000000000011p-1);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)k);
}

// log(x) for x in [1e-300, 1e300] (fdlibm's e_log.c scheme: x = 2^k (1 + f), s = f / (2 + f),
// log(1 + f) = 2 s + s R(s^2) arranged around f - f^2 / 2): < 1 ulp, a third of the compiler's
// expansion (no subnormal / special-value paths; the division is a reciprocal + Newton steps)
__device__ __forceinline__ double fast_log(double x) {
    int k = __builtin_amdgcn_frexp_exp(x);                 // x = m 2^k, m in [0.5, 1)
    double m = __builtin_amdgcn_frexp_mant(x);
    if (m < 0.70710678118654752440) { m *= 2.0; --k; }     // m in [sqrt(1/2), sqrt(2))
    const double f = m - 1.0;
    const double s = f * fast_rcp(2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01),
                              3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                     2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t1 + t2;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    return dk * 6.93147180369123816490e-01 -
           ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}

// e^x through a 256-entry table of 2^(j / 256) held in LDS: x = (256 n + j) ln2 / 256 + r with
// |r| <= ln2 / 512, e^x = 2^n * tab[j] * p(r), p the degree-4 Taylor polynomial (truncation
// 3.8e-17 relative).  13 vector instructions and one LDS read against fast_exp's 18 - the LDS
// pipe is idle in the kernels that are bound by float64 VALU issue.  The integer round(x 256 /
// ln2) is read from the low word of x 256 / ln2 + 1.5 * 2^52: VALID FOR |x| < 5.8e6 only (beyond
// it the word wraps) - minus a Euclidean distance between latent positions is inside by any
// margin; minus a SQUARED distance goes through tab_exp_clamped.  Against the correctly rounded
// exponential (tests/test_exp_table_cpu.py): 2 ulp + |x| / 2 ulp with the one-step argument
// reduction below (the two-step form, 1 ulp everywhere, is kept under DLSM_EXP_TWO_STEP).
constexpr int EXPTAB_N = 256;
// the workgroup's table, by its first 256 threads (callers put a barrier behind it)
__device__ __forceinline__ void exp_table_fill(double *tab, int tid) {
    if (tid < EXPTAB_N) tab[tid] = c_exp2_tab[tid];
}
__device__ __forceinline__ double tab_exp(double x, const double *tab) {
    const double magic = 6755399441055744.0;                     // 1.5 * 2^52
    const double t = fma(x, 369.3299304675746, magic);           // 256 / ln2
    const double kf = t - magic;
    const int ki = __double2loint(t);
#ifdef DLSM_EXP_TWO_STEP
    double r = fma(kf, -0x1.62e42fee00000p-9, x);                // ln2 / 256, high 32 bits
    r = fma(kf, -0x1.a39ef35793c76p-41, r);
#else
    // one step with ln2 / 256 rounded to double: the product is exact inside the fma, so the only
    // error is the constant's rounding, |x| 1.1e-16 on the reduced argument = on e^x relatively
    // (1e-15 at a distance of 10, 9e-15 at 80, where e^x is 1e-35) - one instruction of twelve
    const double r = fma(kf, -0x1.62e42fefa39efp-9, x);
#endif
    double p = fma(r, 1.0 / 24.0, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return __builtin_ldexp(tab[ki & (EXPTAB_N - 1)] * p, ki >> 8);
}

// any x <= 0: e^x is 0 in double below -745.2
__device__ __forceinline__ double tab_exp_clamped(double x, const double *tab) {
    return tab_exp(fmax(x, -1000.0), tab);
}

// The same scheme on a 2048-entry table of 2^(j / 2048) (16 KB of LDS): |r| <= ln2 / 4096 = 1.7e-4,
// so the degree-3 Taylor polynomial is enough (truncation r^4 / 24 = 3.4e-17 relative) and its first
// step, r / 6 + 1 / 2, has one literal and one inline constant - the degree-4 form's r / 24 + 1 / 6
// needs a register for its second literal: 11 vector instructions + one LDS read against 13.  For the
// kernels that own a CU's LDS (the pipelined sweeps' evaluators: one workgroup per CU); the
// log-likelihood passes keep the 2 KB table (six wavefronts per SIMD need the LDS).
// VALID FOR |x| < 7.2e5; error as tab_exp: 2 ulp + |x| / 2 ulp (tests/test_exp_table_cpu.py).
constexpr int EXPTAB11_N = 2048;
// the workgroup's table (callers put a barrier behind it): NT threads, two entries per load
template <int NT>
__device__ __forceinline__ void exp_table11_fill(double *tab, int tid) {
    for (int i = tid; i < EXPTAB11_N / 2; i += NT)
        ((double2 *)tab)[i] = ((const double2 *)c_exp2_tab11)[i];
}
__device__ __forceinline__ double tab_exp11(double x, const double *tab) {
    const double magic = 6755399441055744.0;                     // 1.5 * 2^52
    const double t = fma(x, 2954.639443740597, magic);           // 2048 / ln2
    const double kf = t - magic;
    const int ki = __double2loint(t);
    const double r = fma(kf, -0x1.62e42fefa39efp-12, x);         // ln2 / 2048 rounded to double
    double p = fma(r, 1.0 / 6.0, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return __builtin_ldexp(tab[ki & (EXPTAB11_N - 1)] * p, ki >> 11);
}
__device__ __forceinline__ double tab_exp11_clamped(double x, const double *tab) {
    return tab_exp11(fmax(x, -1000.0), tab);
}

template <int D>
__device__ __forceinline__ double dist_fast(const double *a, const double *b, int squared) {
    double s = squared ? 0.0 : SQRT_GUARD;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        double df = a[d] - b[d];
        s = fma(df, df, s);
    }
    return squared ? s : fast_sqrt_guarded(s);
}

__device__ __forceinline__ int bit_of(const uint32_t *row, int i) {
    return (row[i >> 5] >> (i & 31)) & 1;
}

// metropolis.py:5-20
__device__ __forceinline__ double tune_rw(double step, double rate) {
    if (rate < 0.001) step *= 0.1;
    else if (rate < 0.05) step *= 0.5;
    else if (rate < 0.25) step *= 0.9;
    else if (rate > 0.95) step *= 10.0;
    else if (rate > 0.75) step *= 2.0;
    else if (rate > 0.4) step *= 1.1;
    return step;
}

// metropolis.py:110-136 (incl. the tune_interval+1 window of the reference)
__device__ __forceinline__ void metropolis_bookkeeping(double &step, int32_t &n_acc,
                                                       int32_t &n_steps,
                                                       int32_t &until, int tune,
                                                       int tune_interval,
                                                       int accepted) {
    n_acc += accepted;
    n_steps += 1;
    if (tune >= 0) {
        if (n_steps < tune && until == 0) {
            double rate = (double)n_acc / (double)tune_interval;
            step = tune_rw(step, rate);
            n_acc = 0;
            until = tune_interval;
        } else {
            until -= 1;
        }
    }
}

}  // namespace dlsm
