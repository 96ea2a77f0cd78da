// Device-resident tail of the HDP-LPCM Gibbs iteration (hdp_lpcm.py:876-1023
// and the log-posterior of :1188-1280): everything between the label block update and the
// next sweep, with counter-based draws, so that an iteration never waits for the host.
//
// Draws: Philox stream HDP, counter (index, kind | attempt << 8, iteration); the kinds and the
// index conventions are shared with the CPU oracle (its PhiloxDraws class), which runs the
// reference's update code with the same draws.
//   tables     sample_auxillary.py:6-28   m[t,j,k] = sum_{i < n[t,j,k]} Bernoulli(p / (p + i));
//                                         trial i = uniform i & 1 of attempt i >> 1 of index
//                                         (t K + j) K + k
//   override   sample_auxillary.py:31-50  w[t,j] = Binomial(m[t+1,j,j], rho / (rho + beta_j (1 - rho)))
//                                         as a sum of Bernoullis, index t K + j
//   Dirichlet  hdp_lpcm.py:887-898        gamma variates (Marsaglia-Tsang, attempt a: normal from
//                                         attempt 2a, uniforms from 2a + 1) normalised by their sum
//   mu, sigma, lambda, hyper-parameters, concentration parameters: see each kernel.
// Launch order per iteration (capi_hdp.hpp): [labels], k_hdp_stage1 (label counts + tables | MEAN sums),
// k_hdp_stage2 (override variables, m_bar, beta, w0 | the alpha + kappa grid | mu + RESIDUAL sums),
// k_hdp_stage3 (w | eight gamma variates | sigma + LAMBDA sums), k_hdp_hypers (+ the sample's trace
// rows); the intercept's accept / reject is one more role of stage 1.  The log-posterior trace is
// computed afterwards, batched over the rows a run produced (k_hdp_logp_batch_sums / _finish).
#pragma once
#include "chain.hpp"
#include "device_common.hpp"
#include "kernels_hdp.hpp"

namespace dlsm {

constexpr uint32_t STREAM_HDP = 6;
enum : uint32_t {
    HK_TABLES = 0, HK_OVERRIDE, HK_BETA, HK_W0, HK_W, HK_MU, HK_SIGMA, HK_LAMBDA, HK_MVP, HK_B,
    HK_CONC_GAMMA, HK_CONC_ALPHA0, HK_AK_S, HK_AK_R, HK_AK, HK_RHO
};
constexpr double HDP_SMALL_EPS = 2.2250738585072014e-308;     // np.finfo('float64').tiny

struct HdpRng {
    uint64_t seed; uint32_t sw, iter;
    __device__ __forceinline__ void u2(uint32_t kind, uint32_t idx, uint32_t att, double &u0,
                                       double &u1) const {
        philox_uniform2(seed, idx, kind | (att << 8), iter, sw, u0, u1);
    }
};
__device__ __forceinline__ HdpRng hdp_rng(const ChainView &c, uint32_t iter) {
    return HdpRng{c.seed, stream_word(c.chain, STREAM_HDP), iter};
}

// Gamma(a, 1) by Marsaglia & Tsang (2000); a < 1 through Gamma(a + 1) U^(1 / a)
__device__ inline double hdp_gamma(const HdpRng &g, uint32_t kind, uint32_t idx, double a) {
    const double aa = a < 1.0 ? a + 1.0 : a;
    const double d = aa - 1.0 / 3.0, cc = 1.0 / sqrt(9.0 * d);
    for (uint32_t att = 0; att < 4096; ++att) {
        double u0, u1, z0, z1, w0, w1;
        g.u2(kind, idx, 2 * att, u0, u1);
        box_muller(u0, u1, z0, z1);
        g.u2(kind, idx, 2 * att + 1, w0, w1);
        const double t = 1.0 + cc * z0;
        if (t <= 0.0) continue;
        const double v = t * t * t;
        const double x2 = z0 * z0;
        if (w0 < 1.0 - 0.0331 * x2 * x2 || log(w0) < 0.5 * x2 + d * (1.0 - v + log(v))) {
            double out = d * v;
            if (a < 1.0) out *= pow(w1, 1.0 / a);
            return out;
        }
    }
    return 0.0;
}

// The first attempt's normal and uniforms do not depend on the shape: a lane can draw them while
// the shape is still being summed elsewhere, and finish with hdp_gamma_with (same arithmetic on
// the same numbers as hdp_gamma: the same variate, bit for bit).
struct HdpGammaPre { double z0, w0, w1, lw0; };
__device__ inline HdpGammaPre hdp_gamma_pre(const HdpRng &g, uint32_t kind, uint32_t idx) {
    HdpGammaPre p;
    double u0, u1, z1;
    g.u2(kind, idx, 0, u0, u1);
    box_muller(u0, u1, p.z0, z1);
    g.u2(kind, idx, 1, p.w0, p.w1);
    p.lw0 = log(p.w0);
    return p;
}
__device__ inline double hdp_gamma_with(const HdpRng &g, uint32_t kind, uint32_t idx, double a,
                                        const HdpGammaPre &p) {
    const double aa = a < 1.0 ? a + 1.0 : a;
    const double d = aa - 1.0 / 3.0, cc = 1.0 / sqrt(9.0 * d);
    for (uint32_t att = 0; att < 4096; ++att) {
        double u0, u1, z0, z1, w0, w1, lw0;
        if (att == 0) { z0 = p.z0; w0 = p.w0; w1 = p.w1; lw0 = p.lw0; }
        else {
            g.u2(kind, idx, 2 * att, u0, u1);
            box_muller(u0, u1, z0, z1);
            g.u2(kind, idx, 2 * att + 1, w0, w1);
            lw0 = log(w0);
        }
        const double t = 1.0 + cc * z0;
        if (t <= 0.0) continue;
        const double v = t * t * t;
        const double x2 = z0 * z0;
        if (w0 < 1.0 - 0.0331 * x2 * x2 || lw0 < 0.5 * x2 + d * (1.0 - v + log(v))) {
            double out = d * v;
            if (a < 1.0) out *= pow(w1, 1.0 / a);
            return out;
        }
    }
    return 0.0;
}

__device__ inline double hdp_beta(const HdpRng &g, uint32_t kind, uint32_t idx, double a, double b) {
    const double ga = hdp_gamma(g, kind, 2 * idx, a);
    const double gb = hdp_gamma(g, kind, 2 * idx + 1, b);
    return ga / (ga + gb);
}

#ifdef DLSM_PIPE_TIMING
// phases of the globals' workgroup (stage 2's long pole), rows [4][phase]: profiles/hdp_tail_timing.py
__device__ unsigned long long g_hdp_phase[16][2];
#define DLSM_HDP_PHASE(I_) { __syncthreads(); if (threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); g_hdp_phase[I_][0] = t_; g_hdp_phase[I_][1] = t_; } }
#else
#define DLSM_HDP_PHASE(I_)
#endif

// successes among n Bernoulli(p) trials; trial i = uniform i & 1 of attempt i >> 1
__device__ inline int hdp_binomial(const HdpRng &g, uint32_t kind, uint32_t idx, int n, double p) {
    int cnt = 0;
    for (int i = 0; i < n; i += 2) {
        double u0, u1;
        g.u2(kind, idx, (uint32_t)(i >> 1), u0, u1);
        cnt += (u0 <= p) + (i + 1 < n && u1 <= p);
    }
    return cnt;
}

// ---- truncated normal on [0, 1] in log space (scipy.stats.truncnorm's formulas) -----------
__device__ inline double dev_log_ndtr(double x) {
    if (x > 0.0) return log1p(-0.5 * erfc(x * 0.70710678118654752440));
    return log(0.5 * erfcx(-x * 0.70710678118654752440)) - 0.5 * x * x;
}
__device__ inline double dev_ndtr(double x) { return 0.5 * erfc(-x * 0.70710678118654752440); }
__device__ inline double dev_logaddexp(double x1, double x2) {
    if (x1 == x2) return x1 + 0.69314718055994530942;
    const double d = x1 - x2;
    return d > 0.0 ? x1 + log1p(exp(-d)) : x2 + log1p(exp(d));
}
__device__ inline double dev_log_gauss_mass(double a, double b) {
    if (a > 0.0) { const double t = a; a = -b; b = -t; }     // mass of [a, b] = mass of [-b, -a]
    if (b <= 0.0) {
        const double la = dev_log_ndtr(a), lb = dev_log_ndtr(b);
        return lb + log1p(-exp(la - lb));
    }
    return log1p(-dev_ndtr(a) - dev_ndtr(-b));
}
// x with log ndtr(x) = y (y <= 0): a closed-form start, Newton steps on log ndtr
__device__ inline double dev_ndtri_exp(double y) {
    double x;
    int polish = 2;
    if (y >= -0.69314718055994530942) {
        x = -normcdfinv(-expm1(y));
    } else if (y >= -30.0) {
        x = normcdfinv(exp(y));
    } else {                        // far tail: log ndtr(x) ~ -x^2/2 - log(-x) - log(2 pi)/2
        x = -sqrt(-2.0 * (y + 0.91893853320467274178));
        for (int i = 0; i < 3; ++i) x = -sqrt(-2.0 * (y + 0.91893853320467274178 + log(-x)));
        polish = 4;
    }
    for (int i = 0; i < polish; ++i) {
        const double l = dev_log_ndtr(x);
        // (converged: the remaining steps would move x by less than an ulp - each is three dependent
        // transcendental calls on the launch's longest chain, k_hdp_hypers)
        if (fabs(l - y) <= 4.4e-16 * fabs(y)) break;
        // d/dx log ndtr = phi / ndtr
        const double dl = x < 0.0 ? 0.79788456080286535588 / erfcx(-x * 0.70710678118654752440)
                                  : exp(-0.5 * x * x - 0.91893853320467274178 - l);
        if (!(dl > 0.0) || !isfinite(dl)) break;
        x -= (l - y) / dl;
    }
    return x;
}
__device__ inline double dev_truncnorm_quantile(double q, double mean, double var) {
    const double sd = sqrt(var);
    const double a = (0.0 - mean) / sd, b = (1.0 - mean) / sd;
    const double lm = dev_log_gauss_mass(a, b);
    double x;
    if (a < 0.0) x = dev_ndtri_exp(dev_logaddexp(dev_log_ndtr(a), log(q) + lm));
    else x = -dev_ndtri_exp(dev_logaddexp(dev_log_ndtr(-b), log1p(-q) + lm));
    return x * sd + mean;
}
__device__ inline double dev_truncnorm_logpdf(double x, double mean, double var) {
    const double sd = sqrt(var);
    const double a = (0.0 - mean) / sd, b = (1.0 - mean) / sd;
    const double y = (x - mean) / sd;
    if (y < a || y > b) return -INFINITY;
    return -0.5 * y * y - 0.91893853320467274178 - dev_log_gauss_mass(a, b) - log(sd);
}

// ---- buffers ------------------------------------------------------------------------------
struct HdpLoopBuf {
    double *beta;        // [K]
    double *w;           // [T][K][K]  (the label kernel's transition matrices)
    const int32_t *n;    // [T][K][K]  label transition counts (k_label_counts)
    const int32_t *nk;   // [T][K]
    int32_t *m;          // [T][K][K]  tables
    int32_t *wover;      // [T-1][K]   override counts
    double *mbar;        // [K]
    double *S;           // [T][K][D]  label sums, stage MEAN
    double *Q;           // [T][K]     stage RESIDUAL
    double *L;           // [T][K][2]  stage LAMBDA
    double *LP;          // [T][K]     stage LOGP
    double *LPD;         // [T][K]     Dirichlet log-densities of the rows (k_hdp_dirichlet_rows)
    double *mu, *sigma;  // the mixture prior's (chain->mu, chain->sigma)
    double *scr;         // [HS_COUNT] sums and draws handed from the multi-role launches to k_hdp_hypers
    int K;
};

struct HdpTrace {
    double *ic, *logp, *mu, *sigma, *beta, *w, *lambda, *hyper;
};

// ---- label counts + tables: workgroup (t, g) counts time step t, then draws HT_CELLS of its cells ------
// The counts the conjugate updates need (sample_labels.py:176-188: n[0][0][k] initial labels, n[t][j][k]
// transitions j -> k into time t, nk[t][k] labels in use) were a launch of their own between the label
// update and this one (k_label_counts: T workgroups, 4.6 us of which 2.6 are the launch).  The only
// role of THIS launch that reads them is the tables', and a time step's histogram is 2 N labels into
// K K + K LDS counters (integer atomics: order independent): every tables workgroup of a time step
// counts it for itself, group 0 files n / nk for the later launches, each group files its share of the
// sample's label trace row (bytes).  Then one wavefront per (t, j, k) cell, as before.
constexpr int HT_WAVES = 4;
constexpr int HT_CELLS = 8;                 // cells of a time step per workgroup: two per wavefront
__host__ __device__ inline int hdp_tab_groups(int K) { return (K * K + HT_CELLS - 1) / HT_CELLS; }
__device__ __forceinline__ void hdp_counts_tables_wg(const ChainView &c, const HdpLoopBuf &hb,
                                                     const HdpDeviceState *hs, uint32_t iter, int wg,
                                                     int32_t *hist /* K K + K */, uint8_t *trace_row) {
    const int K = hb.K, N = c.N, tid = threadIdx.x;
    const int G = hdp_tab_groups(K);
    const int t = wg / G, g = wg - t * G;
    const int32_t *zt = c.z + (size_t)t * N;
    const int32_t *zp = t > 0 ? c.z + (size_t)(t - 1) * N : nullptr;
    const int per = (N + G - 1) / G;        // nodes whose trace bytes this group files
    constexpr int PRE = 8;                  // labels requested together (one round trip at N = 2000)
    int a[PRE], b[PRE];
#pragma unroll
    for (int u = 0; u < PRE; ++u) {
        const int i = tid + u * HDP_THREADS;
        a[u] = i < N ? zt[i] : -1;
        b[u] = i < N && t > 0 ? zp[i] : 0;
    }
    for (int q = tid; q < K * K + K; q += HDP_THREADS) hist[q] = 0;
    __syncthreads();
    for (int i0 = tid;;) {
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int i = i0 + u * HDP_THREADS;
            if (a[u] < 0) continue;
            atomicAdd(&hist[b[u] * K + a[u]], 1);
            atomicAdd(&hist[K * K + a[u]], 1);
            if (trace_row && i / per == g) trace_row[(size_t)t * N + i] = (uint8_t)a[u];
        }
        i0 += PRE * HDP_THREADS;
        if (i0 - tid >= N) break;
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int i = i0 + u * HDP_THREADS;
            a[u] = i < N ? zt[i] : -1;
            b[u] = i < N && t > 0 ? zp[i] : 0;
        }
    }
    __syncthreads();
    if (g == 0) {
        int32_t *n_out = (int32_t *)hb.n, *nk_out = (int32_t *)hb.nk;
        for (int q = tid; q < K * K; q += HDP_THREADS) n_out[(size_t)t * K * K + q] = hist[q];
        for (int q = tid; q < K; q += HDP_THREADS) nk_out[t * K + q] = hist[K * K + q];
    }
    const int lane = tid & 63;
    const HdpRng rg = hdp_rng(c, iter);
    for (int cl = g * HT_CELLS + (tid >> 6); cl < min(K * K, (g + 1) * HT_CELLS); cl += HT_WAVES) {
        const int cell = t * K * K + cl, j = cl / K, k = cl - j * K;
        int cnt = 0;
        if (t > 0 || j == 0) {          // t = 0: only the initial distribution's row (0, 0, :)
            const int n = hist[cl];
            const double p = t == 0 ? hs->alpha_init * hb.beta[k]
                                    : hs->alpha * hb.beta[k] + (j == k ? hs->kappa : 0.0);
            for (int a0 = 0; 2 * a0 < n; a0 += 64) {
                const int att = a0 + lane, i0 = 2 * att;
                double u0 = 2.0, u1 = 2.0;
                if (i0 < n) rg.u2(HK_TABLES, (uint32_t)cell, (uint32_t)att, u0, u1);
                const bool s0 = i0 < n && u0 <= p / (p + (double)i0);
                const bool s1 = i0 + 1 < n && u1 <= p / (p + (double)(i0 + 1));
                cnt += __popcll(__ballot(s0)) + __popcll(__ballot(s1));
            }
        }
        if (lane == 0) hb.m[cell] = cnt;
        // The override variable of (t - 1, j) (sample_auxillary.py:37-42) is a binomial over exactly this
        // cell's tables, m[t, j, j], with a success probability made of LAST iteration's beta and
        // concentration parameters: drawn here, its trials on the lanes like the tables' (the same
        // uniforms as hdp_binomial's sequential loop - trial i = uniform i & 1 of attempt i >> 1 - and an
        // integer count).  In stage 2's globals workgroup, a lane per binomial, the longest of them
        // (hundreds of trials one after the other) was 10 of the launch's 16 us.
        if (t > 0 && j == k) {
            const double rho = hs->kappa / (hs->alpha + hs->kappa);
            const double p = rho / (rho + hb.beta[j] * (1 - rho));
            const uint32_t q = (uint32_t)((t - 1) * K + j);
            int w = 0;
            for (int a0 = 0; 2 * a0 < cnt; a0 += 64) {
                const int att = a0 + lane, i0 = 2 * att;
                double u0 = 2.0, u1 = 2.0;
                if (i0 < cnt) rg.u2(HK_OVERRIDE, q, (uint32_t)att, u0, u1);
                w += __popcll(__ballot(i0 < cnt && u0 <= p)) + __popcll(__ballot(i0 + 1 < cnt && u1 <= p));
            }
            if (lane == 0) hb.wover[q] = w;
        }
    }
}

// ---- override variables, m_bar, beta, w0: one workgroup -----------------------------------------
constexpr int HG_THREADS = 256;
constexpr int HG_MCAP = 8192;           // table cells the globals' workgroup stages in LDS (32 KB)
__device__ __forceinline__ void hdp_globals_wg(const ChainView &c, const HdpLoopBuf &hb,
                                               HdpDeviceState *hs, uint32_t iter) {
    // This workgroup is stage 2's long pole (profiles/hdp_tail_timing.py: 17 us against 9 for the
    // other roles), and its three chains ran one after the other: a lane per override binomial
    // (6.6 us), the tables' column sums a cell per trip through an LDS atomic (4.8 us), a lane per
    // gamma variate of beta (4.0 us).  Round 3: wavefronts 0-2 drew the binomials while wavefront 3
    // staged the tables in LDS, summed their columns and drew the shape-independent part of beta's
    // variates (17.4 -> 15 us: both sides ~10 us).  Round 4: the binomials are drawn in stage 1 beside
    // the tables they are binomials over (a wavefront's lanes = the trials), so wavefronts 0-2 sum the
    // override variables and the tables' columns (nine threads per column at config 3) while
    // wavefront 3 only draws the shape-independent part of the variates; what is left behind the
    // barrier is the variates' shape-dependent part.  Same draws, integer sums: the same values.
    // (Built and measured slower: the binomials' trials dealt out over all threads - 1800 pairs
    // at config 3 - through a scan and a pair -> cell search: 7.8 us; a column's rows read from
    // global memory by nine threads: 9.2 us.)
    __shared__ int sMsum[64], sWsum[64], sTot[4];
    __shared__ double sG[64];
    __shared__ int sMt[HG_MCAP];
    __shared__ int sReady;
    const int K = hb.K, T = c.T, tid = threadIdx.x, wave = tid >> 6;
    const int TKK = T * K * K;
    const bool mlds = TKK <= HG_MCAP;
    const HdpRng g = hdp_rng(c, iter);
    if (tid < 64) { sMsum[tid] = 0; sWsum[tid] = 0; }
    if (tid < 4) sTot[tid] = 0;
    if (tid == 0) sReady = 0;
    __syncthreads();
    DLSM_HDP_PHASE(0)
    // every thread stages its share of the tables (one burst of loads); the wavefronts that sum the
    // columns wait for the shares on a counter in LDS, wavefront 3 goes on at once
    const int wfirst = wave < 3 && tid < (T - 1) * K ? hb.wover[tid] : 0;
    if (mlds) {
        constexpr int NB = 16;                      // (config 3: the 4000 cells in ONE burst)
        for (int q0 = tid; q0 < TKK; q0 += NB * HG_THREADS) {
            int v[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) v[u] = q0 + u * HG_THREADS < TKK ? hb.m[q0 + u * HG_THREADS] : 0;
#pragma unroll
            for (int u = 0; u < NB; ++u) if (q0 + u * HG_THREADS < TKK) sMt[q0 + u * HG_THREADS] = v[u];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);         // lgkmcnt(0): this wavefront's LDS writes landed
        __builtin_amdgcn_wave_barrier();
        if ((tid & 63) == 0) atomicAdd(&sReady, 1);
    }
    HdpGammaPre gp{0.0, 0.0, 0.0, 0.0};
    const int l3 = tid - 3 * 64;                    // lane of wavefront 3 = component
    if (wave == 3) {
        // the shape-independent part of beta's variates: needs nothing summed here
        if (l3 < K) gp = hdp_gamma_pre(g, HK_BETA, (uint32_t)l3);
    } else {
        // sums of the override variables (sample_auxillary.py:37-42; drawn in stage 1 beside the tables
        // they are binomials over: hdp_counts_tables_wg; the first 192 were requested with the tables)
        int wsum = wfirst;
        if (tid < (T - 1) * K) atomicAdd(&sWsum[tid % K], wfirst);
        for (int q = tid + 3 * 64; q < (T - 1) * K; q += 3 * 64) {
            const int j = q % K;
            const int w = hb.wover[q];
            atomicAdd(&sWsum[j], w);
            wsum += w;
        }
        atomicAdd(&sTot[0], wsum);
        // column sums of the tables: m_bar[k] = sum_{t >= 1, j} m[t, j, k] - sum_t w[t, k] + m[0, 0, k]
        // (nsub threads per column, four rows requested per trip: a row per trip is an LDS round trip
        // per row, 200 in a row at config 3) - by the three wavefronts that used to draw the binomials
        if (mlds)
            while (*(volatile int *)&sReady < HG_THREADS / 64) __builtin_amdgcn_s_sleep(1);
        const int KK = min(K, 3 * 64);
        const int nsub = max(1, (3 * 64) / KK);
        for (int k0 = 0; k0 < K; k0 += 3 * 64) {
            const int k = k0 + tid % KK, sub = tid / KK;
            if (k >= K || sub >= nsub) continue;
            int acc = 0, rest = 0;
            for (int r0 = sub; r0 < T * K; r0 += 4 * nsub) {        // row r = (t, j)
                int v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = min(r0 + u * nsub, T * K - 1);
                    v[u] = mlds ? sMt[r * K + k] : hb.m[(size_t)r * K + k];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = r0 + u * nsub;
                    if (r >= T * K || (r < K && r != 0)) continue;  // t = 0: only j = 0
                    acc += v[u];
                    if (r >= K) rest += v[u];
                }
            }
            atomicAdd(&sMsum[k], acc);
            atomicAdd(&sTot[1], rest);
        }
    }
    __syncthreads();
    DLSM_HDP_PHASE(2)
    // global transition distribution beta ~ Dirichlet(gamma / K + m_bar) (hdp_lpcm.py:887)
    if (l3 >= 0 && l3 < K) {
        const double mb = (double)(sMsum[l3] - sWsum[l3]);
        hb.mbar[l3] = mb;
        sG[l3] = hdp_gamma_with(g, HK_BETA, (uint32_t)l3, hs->gamma / K + mb, gp);
    }
    __syncthreads();
    DLSM_HDP_PHASE(3)
    double tot = 0.0;
    for (int k = 0; k < K; ++k) tot += sG[k];
    const double bnew = tid < K ? sG[tid] * (1.0 / tot) : 0.0;
    // (the initial distribution w0, whose gamma draws wait for beta, is drawn in stage 3 beside
    // the other rows of w)
    if (tid < K) hb.beta[tid] = bnew;
    if (tid == 0) {
        double mbt = 0.0, mbp = 0.0, m00 = 0.0;
        for (int k = 0; k < K; ++k) {
            const double v = (double)(sMsum[k] - sWsum[k]);
            mbt += v; mbp += v > 0.0 ? 1.0 : 0.0;
            m00 += (double)hb.m[k];
        }
        hs->mbar_total = mbt; hs->mbar_positive = mbp; hs->m00_total = m00;
        hs->m_rest_total = (double)sTot[1]; hs->override_total = (double)sTot[0];
    }
    DLSM_HDP_PHASE(4)
}

// ---- transition distributions w[t, j, :] ~ Dirichlet(alpha beta + kappa e_j + n[t, j, :]) ---------
// (hdp_lpcm.py:894-898); workgroup t - 1 draws the K x K gamma variates of time t
// HW_SPLIT workgroups per time step, each a band of rows (a row is normalised by itself): one
// gamma variate per thread at K = 20 instead of two in a row (the role was stage 3's long pole)
constexpr int HW_SPLIT = 2;
__device__ __forceinline__ void hdp_weights_wg(const ChainView &c, const HdpLoopBuf &hb,
                                               const HdpDeviceState *hs, uint32_t iter, int t, int part,
                                               double *sGam /* K * K + K */) {
    const int K = hb.K, tid = threadIdx.x;
    const int jb = (K + HW_SPLIT - 1) / HW_SPLIT, j0 = part * jb, j1 = min(K, j0 + jb);
    double *sInv = sGam + K * K;
    const HdpRng g = hdp_rng(c, iter);
    const double alpha = hs->alpha, kappa = hs->kappa;
    for (int q = j0 * K + tid; q < j1 * K; q += 256) {
        const int j = q / K, k = q - j * K;
        double al = (alpha * hb.beta[k] + (j == k ? kappa : 0.0)) + (double)hb.n[(size_t)t * K * K + q];
        if (al <= 0.0) al = HDP_SMALL_EPS;
        sGam[q] = hdp_gamma(g, HK_W, (uint32_t)(t * K * K + q), al);
    }
    __syncthreads();
    if (tid >= j0 && tid < j1) {
        double tot = 0.0;
        for (int k = 0; k < K; ++k) tot += sGam[tid * K + k];
        sInv[tid] = 1.0 / tot;
    }
    __syncthreads();
    for (int q = j0 * K + tid; q < j1 * K; q += 256) hb.w[(size_t)t * K * K + q] = sGam[q] * sInv[q / K];
}

// ---- cluster means (hdp_lpcm.py:901-921) and variances (:924-938) -----------------------------------
// The draw of cluster k needs T label sums and one Philox call: every workgroup (k, t) of the
// label-sum pass that follows redoes it for its own k (same counters, same value) instead of
// waiting for a launch of its own; the workgroup t = 0 files the value.
template <int D>
__device__ __forceinline__ void hdp_mu_of(const ChainView &c, const HdpLoopBuf &hb,
                                          const HdpDeviceState *hs, uint32_t iter, int k,
                                          double (&mu)[D]) {
    const int K = hb.K, T = c.T;
    const HdpRng g = hdp_rng(c, iter);
    const double lm = hs->lmbda, sk = hb.sigma[k];
    double pk = 1.0 / hs->mvp, mk[D];
#pragma unroll
    for (int d = 0; d < D; ++d) mk[d] = 0.0;
    // (eight time steps' counts and sums requested together, added in the same order)
    constexpr int NT = 8;
    for (int t0 = 0; t0 < T; t0 += NT) {
        int nv[NT];
        double sv[NT][D];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int t = min(t0 + u, T - 1);
            nv[u] = t0 + u < T ? hb.nk[t * K + k] : 0;
            const double *S = hb.S + ((size_t)t * K + k) * D;
#pragma unroll
            for (int d = 0; d < D; ++d) sv[u][d] = S[d];
        }
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int cnt = nv[u];
            if (cnt <= 0) continue;
            if (t0 + u == 0) {
                pk += (double)cnt / sk;
#pragma unroll
                for (int d = 0; d < D; ++d) mk[d] += (1.0 / sk) * sv[u][d];
            } else {
                pk += (lm * lm / sk) * (double)cnt;
#pragma unroll
                for (int d = 0; d < D; ++d) mk[d] += (lm / sk) * sv[u][d];
            }
        }
    }
    pk = 1.0 / pk;
    const double sd = sqrt(pk);
#pragma unroll
    for (int d = 0; d < D; d += 2) {
        double u0, u1, z0, z1;
        g.u2(HK_MU, (uint32_t)k, (uint32_t)(d >> 1), u0, u1);
        box_muller(u0, u1, z0, z1);
        mu[d] = mk[d] * pk + sd * z0;
        if (d + 1 < D) mu[d + 1] = mk[d + 1] * pk + sd * z1;
    }
}

__device__ __forceinline__ double hdp_sigma_of(const ChainView &c, const HdpLoopBuf &hb,
                                               const HdpDeviceState *hs, uint32_t iter, int k) {
    const int K = hb.K, T = c.T;
    const HdpRng g = hdp_rng(c, iter);
    long cnt = 0;
    double bk = 0.5 * hs->b;
    // (eight time steps' counts and sums requested together - a step at a time was two dependent round
    // trips per step on stage 3's longest chain -, added in the same order)
    constexpr int NT = 8;
    for (int t0 = 0; t0 < T; t0 += NT) {
        int nv[NT];
        double qv[NT];
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int t = min(t0 + u, T - 1);
            nv[u] = t0 + u < T ? hb.nk[t * K + k] : 0;
            qv[u] = hb.Q[(size_t)t * K + k];
        }
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            cnt += nv[u];
            if (nv[u] > 0) bk += 0.5 * qv[u];
        }
    }
    const double ak = 0.5 * ((double)cnt * c.D + hs->a);
    return 1.0 / (hdp_gamma(g, HK_SIGMA, (uint32_t)k, ak) * (1.0 / bk));
}

// mu_k drawn in the prologue, then the squared residuals about it (stage RESIDUAL)
template <int D>
__device__ __forceinline__ void hdp_mu_residual_wg(const ChainView &c, const HdpLoopBuf &hb,
                                                   const HdpDeviceState *hs, uint32_t iter, int k,
                                                   int t) {
    double mk[D];
    hdp_mu_of<D>(c, hb, hs, iter, k, mk);
    if (t == 0 && threadIdx.x == 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) hb.mu[(size_t)k * D + d] = mk[d];
    }
    hdp_label_sums_wg<D, HDP_SUMS_RESIDUAL>(c, k, t, mk, 1.0, hs->lmbda, 0.0, 0.0, nullptr, hb.Q);
}

// sigma_k drawn in the prologue (one lane per workgroup: a gamma variate), then the sums of the
// blending coefficient's conditional (stage LAMBDA)
template <int D>
__device__ __forceinline__ void hdp_sigma_lambda_wg(const ChainView &c, const HdpLoopBuf &hb,
                                                    const HdpDeviceState *hs, uint32_t iter, int k,
                                                    int t) {
    __shared__ double sSig;
    if (threadIdx.x == 0) {
        sSig = hdp_sigma_of(c, hb, hs, iter, k);
        if (t == 0) hb.sigma[k] = sSig;
    }
    __syncthreads();
    double mk[D];
#pragma unroll
    for (int d = 0; d < D; ++d) mk[d] = hb.mu[(size_t)k * D + d];
    hdp_label_sums_wg<D, HDP_SUMS_LAMBDA>(c, k, t, mk, sSig, hs->lmbda, 0.0, 0.0, nullptr, hb.L);
}

// ---- blending coefficient, variance hyper-parameters, concentration parameters: one workgroup ------
// (hdp_lpcm.py:941-1023).  A float64 gamma variate costs a lone lane ~2 us of dependent
// transcendental latency, so the draws that do not depend on each other run side by side: the
// (t, j) grid of the alpha + kappa update over all threads, eight gamma variates on the lanes of
// the last wavefront and the truncated-normal quantile on another one at the same time, then the
// three gamma variates whose shapes depend on those (Escobar & West's mixture indicator, the
// alpha + kappa shape) on three lanes.  Counters are fixed per draw, so the values do not depend
// on which lane makes them.
constexpr int HH_THREADS = 256;
// scratch slots of HdpLoopBuf::scr
enum : int { HS_SSUM = 0, HS_LOGR = 1, HS_MVAL = 2, HS_GAM8 = 3 /* .. 10 */, HS_COUNT = 16 };

// alpha + kappa's grid: s ~ Bernoulli(n. / (n. + alpha + kappa)), r ~ Beta(alpha + kappa + 1, n.)
// for every (t >= 1, j) with n.[t, j] = sum_k n[t, j, k] > 0 (hdp_lpcm.py:999-1011): their sums.
// Needs the label counts and the tables only, so it rides in the launch of the globals.
__device__ __forceinline__ void hdp_akgrid_wg(const ChainView &c, const HdpLoopBuf &hb,
                                              const HdpDeviceState *hs, uint32_t iter) {
    __shared__ double red[3][HH_THREADS / 64];
    const int K = hb.K, T = c.T, tid = threadIdx.x;
    const HdpRng g = hdp_rng(c, iter);
    const double ak0 = hs->alpha + hs->kappa;
    double s_sum = 0.0, logr_sum = 0.0, mval_sum = 0.0;
    for (int q = tid; q < (T - 1) * K; q += HH_THREADS) {
        const int t = q / K + 1, j = q - (t - 1) * K;
        long ndot = 0, mrow = 0;
        for (int k = 0; k < K; ++k) {
            ndot += hb.n[((size_t)t * K + j) * K + k];
            mrow += hb.m[((size_t)t * K + j) * K + k];
        }
        if (ndot <= 0) continue;
        double u0, u1;
        g.u2(HK_AK_S, (uint32_t)q, 0, u0, u1);
        s_sum += u0 <= (double)ndot / ((double)ndot + ak0) ? 1.0 : 0.0;
        logr_sum += log(hdp_beta(g, HK_AK_R, (uint32_t)q, ak0 + 1.0, (double)ndot));
        mval_sum += (double)mrow;
    }
    s_sum = block_sum_all<HH_THREADS / 64>(s_sum, red[0], tid);
    logr_sum = block_sum_all<HH_THREADS / 64>(logr_sum, red[1], tid);
    mval_sum = block_sum_all<HH_THREADS / 64>(mval_sum, red[2], tid);
    if (tid == 0) { hb.scr[HS_SSUM] = s_sum; hb.scr[HS_LOGR] = logr_sum; hb.scr[HS_MVAL] = mval_sum; }
}

// the eight gamma variates whose shapes are known once the globals are: the two Beta draws of
// Escobar & West's eta (sample_concentration.py:11), the two variance hyper-parameters
// (hdp_lpcm.py:957-972), rho's Beta draw (:1017-1020); one lane each
__device__ __forceinline__ void hdp_gam8_wg(const ChainView &c, const HdpLoopBuf &hb,
                                            const HdpDeviceState *hs, uint32_t iter) {
    const int K = hb.K, lane = threadIdx.x;
    const HdpRng g = hdp_rng(c, iter);
    {   // second wavefront: initial distribution w0 ~ Dirichlet(alpha_init beta + nk[0])
        // (hdp_lpcm.py:890), beta being stage 2's
        __shared__ double sG0[64];
        const int j = lane - 64;
        if (j >= 0 && j < K) {
            double al = hs->alpha_init * hb.beta[j] + (double)hb.nk[j];
            if (al <= 0.0) al = HDP_SMALL_EPS;
            sG0[j] = hdp_gamma(g, HK_W0, (uint32_t)j, al);
        }
        __syncthreads();
        if (j >= 0 && j < K) {
            double tot = 0.0;
            for (int k = 0; k < K; ++k) tot += sG0[k];
            hb.w[j] = sG0[j] * (1.0 / tot);
        }
    }
    if (lane >= 8) return;
    const double nsucc = hs->override_total;
    uint32_t kind = HK_CONC_GAMMA, idx = (uint32_t)(lane & 1);
    double shape = lane == 0 ? hs->gamma + 1.0 : hs->mbar_total;
    if (lane == 2 || lane == 3) { kind = HK_CONC_ALPHA0; shape = lane == 2 ? hs->alpha_init + 1.0 : (double)c.N; }
    if (lane == 4) { kind = HK_MVP; idx = 0; shape = 0.5 * (hs->a0 + K); }
    if (lane == 5) { kind = HK_B; idx = 0; shape = 0.5 * (hs->c0 + K * hs->a); }
    if (lane == 6 || lane == 7) { kind = HK_RHO; shape = lane == 6 ? 8.0 + nsucc : hs->m_rest_total - nsucc + 2.0; }
    hb.scr[HS_GAM8 + lane] = hdp_gamma(g, kind, idx, shape);
}

#ifdef DLSM_PIPE_TIMING
// entry / exit stamps (100 MHz) of every workgroup of the six launches behind the label update
// (profiles/hdp_tail_timing.py)
__device__ unsigned long long g_hdp_t[6][512][2];
struct HdpStamp {
    int kid, blk;
    __device__ HdpStamp(int kid_) : kid(kid_), blk((int)(blockIdx.x + gridDim.x * blockIdx.y)) {
        if (threadIdx.x == 0 && blk < 512) {
            unsigned long long t;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
            g_hdp_t[kid][blk][0] = t;
        }
    }
    __device__ ~HdpStamp() {
        if (threadIdx.x == 0 && blk < 512) {
            unsigned long long t;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
            g_hdp_t[kid][blk][1] = t;
        }
    }
};
#define DLSM_HDP_STAMP(K_) HdpStamp hdp_stamp_(K_);
#else
#define DLSM_HDP_STAMP(K_)
#endif

// what is left for the launch of its own: the blending coefficient (needs the LAMBDA sums), the
// three gamma variates whose shapes depend on the draws above, the new values
// ... and the sample's trace rows (mu, sigma, beta, w are final by now; lambda and the six
// hyper-parameters as thread 0 sets them).  The log-posterior of the sample is NOT computed here:
// dlsm_hdp_run evaluates it for all the rows it produced in one batched pass over the trace
// (k_hdp_logp_batch_*), off the iteration's critical path.
__device__ __forceinline__ void hdp_hypers_wg(const ChainView &c, const HdpLoopBuf &hb,
                                              HdpDeviceState *hs, const HdpTrace &tr, IterRef ir) {
    DLSM_HDP_STAMP(3)
    __shared__ double sC[3], sLam, sEp[2];
    const int K = hb.K, T = c.T, D = c.D, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const HdpRng g = hdp_rng(c, ir.get());
    const double *sB = hb.scr + HS_GAM8;
    // Three chains that do not wait for each other, each on wavefronts of its own and with one
    // barrier behind all of them: wavefront 1 the blending coefficient (its sums by the wavefront
    // alone: no workgroup barrier in front of the quantile), wavefront 0 the gamma variates (from
    // the first instruction: they need nothing summed here), wavefronts 2-3 the trace rows and
    // the two sums of the variance hyper-parameters.
    if (wave == 1) {
        // sums of the lambda update over the cells (t >= 1, k) with members (hdp_lpcm.py:941-950)
        // (counts and sums of four trips requested together: a trip at a time was two dependent round
        // trips per trip, count then sums, on the launch's longest chain; a lane still adds its cells in
        // ascending order)
        double a0 = 0.0, a1 = 0.0;
        constexpr int NQ = 4;
        for (int q0 = K + lane; q0 < T * K; q0 += NQ * 64) {
            int nkv[NQ];
            double l0[NQ], l1[NQ];
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const int q = min(q0 + u * 64, T * K - 1);
                nkv[u] = q0 + u * 64 < T * K ? hb.nk[q] : 0;
                l0[u] = hb.L[2 * (size_t)q];
                l1[u] = hb.L[2 * (size_t)q + 1];
            }
#pragma unroll
            for (int u = 0; u < NQ; ++u)
                if (nkv[u] > 0) { a0 += l0[u]; a1 += l1[u]; }
        }
        const double ml_sum = wave_sum_all(a0);
        const double sl_sum = wave_sum_all(a1);
        if (lane == 0) {
            // blending coefficient (hdp_lpcm.py:951-954)
            double sl = 1.0 / hs->lambda_var + sl_sum;
            sl = 1.0 / sl;
            double ml = ml_sum + hs->lambda_prior / hs->lambda_var;
            ml *= sl;
            double u0, u1;
            g.u2(HK_LAMBDA, 0, 0, u0, u1);
            sLam = dev_truncnorm_quantile(u0, ml, sl);
        }
    }
    // Escobar & West's gamma draws (sample_concentration.py:13-21) and alpha + kappa
    if (tid < 3) {
        uint32_t kind = HK_AK, idx = 0;
        double shape = hs->ak_shape + hb.scr[HS_MVAL] - hb.scr[HS_SSUM];
        double m_scale = hs->ak_rate - hb.scr[HS_LOGR];
        if (tid < 2) {
            kind = tid == 0 ? HK_CONC_GAMMA : HK_CONC_ALPHA0;
            idx = 3;
            const double eta = sB[2 * tid] / (sB[2 * tid] + sB[2 * tid + 1]);
            const double n_clusters = tid == 0 ? hs->mbar_positive : hs->m00_total;
            const double n_samples = tid == 0 ? hs->mbar_total : (double)c.N;
            shape = (tid == 0 ? hs->gamma_shape : hs->alpha0_shape) + n_clusters - 1.0;
            m_scale = (tid == 0 ? hs->gamma_rate : hs->alpha0_rate) - log(eta);
            const double log_odds = (shape / m_scale) * (1.0 / n_samples);
            double u0, u1;
            g.u2(kind, 2, 0, u0, u1);
            if (u0 <= log_odds / (1.0 + log_odds)) shape += 1.0;
        }
        sC[tid] = hdp_gamma(g, kind, idx, shape) * (1.0 / m_scale);
    }
    if (wave >= 2) {    // the sample's trace rows, by the WAVEFRONTS that have no draw to make (lanes of
                        // the drawing wavefronts would run their stores in turn with the draws)
        constexpr int NS = HH_THREADS - 128;
        const int st = tid - 128;
        const size_t it = ir.get();
        for (int q = st; q < K * D; q += NS) tr.mu[it * K * D + q] = hb.mu[q];
        for (int q = st; q < K; q += NS) {
            tr.sigma[it * K + q] = hb.sigma[q];
            tr.beta[it * K + q] = hb.beta[q];
        }
        for (int q = st; q < T * K * K; q += NS) tr.w[it * T * K * K + q] = hb.w[q];
        if (tid == 128 && hs->has_a0) {             // (the order of the sums is the reference's)
            double b = 0.5 * hs->b0;
            for (int k = 0; k < K; ++k) {
                double ss = 0.0;
                for (int d = 0; d < D; ++d) ss += hb.mu[(size_t)k * D + d] * hb.mu[(size_t)k * D + d];
                b += 0.5 * ss;
            }
            sEp[0] = b;
        }
        if (tid == 192 && hs->has_c0) {
            double scale = 0.5 * hs->d0;
            for (int k = 0; k < K; ++k) scale += 0.5 * (1.0 / hb.sigma[k]);
            sEp[1] = scale;
        }
    }
    __syncthreads();
    if (tid != 0) return;
    hs->lmbda = sLam;
    if (hs->has_a0) hs->mvp = 1.0 / (sB[4] * (1.0 / sEp[0]));
    if (hs->has_c0) hs->b = sB[5] * (1.0 / sEp[1]);
    hs->gamma = sC[0];
    hs->alpha_init = sC[1];
    const double ak = sC[2];
    const double rho = sB[6] / (sB[6] + sB[7]);
    hs->kappa = ak * rho;
    hs->alpha = ak - hs->kappa;
    const size_t it = ir.get();
    tr.lambda[it] = hs->lmbda;
    double *hy = tr.hyper + it * 6;
    hy[0] = hs->gamma; hy[1] = hs->alpha_init; hy[2] = hs->alpha; hy[3] = hs->kappa;
    hy[4] = hs->mvp; hy[5] = hs->b;
}

__global__ __launch_bounds__(HH_THREADS) void k_hdp_hypers(ChainView c, HdpLoopBuf hb,
                                                           HdpDeviceState *hs, HdpTrace tr, IterRef ir) {
    hdp_hypers_wg(c, hb, hs, tr, ir);
}

// intercept step (sample_coefficients.py:76-86 around the fused two-candidate pass whose records
// are in `partials`): one workgroup; nothing inside the iteration reads its results
template <int NF = 12>
__device__ __forceinline__ void hdp_intercept_wg(const double *__restrict__ partials, int nrec,
                                                 LsmDeviceState *lsm, HdpDeviceState *hs,
                                                 double *__restrict__ intercept,
                                                 double *__restrict__ trace_ic, int it) {
    __shared__ double scratch[4 * 256];
    __shared__ double sums[4];
    reduce_records<NF>(partials, nrec, 4, sums, scratch, threadIdx.x);
    if (threadIdx.x == 0) {
        const double b0 = lsm->cand[0], b1 = lsm->cand[1];
        const double ll0 = b0 * sums[0] - sums[1] - sums[2];
        const double ll1 = b1 * sums[0] - sums[1] - sums[3];
        const double pm = lsm->intercept_prior[0], v = lsm->intercept_var;
        const double lp0 = ll0 - (b0 - pm) * (b0 - pm) / (2 * v);
        const double lp1 = ll1 - (b1 - pm) * (b1 - pm) / (2 * v);
        const int accepted = !(lsm->logu >= lp1 - lp0);
        intercept[0] = accepted ? b1 : b0;
        hs->ll = accepted ? ll1 : ll0;
        double st = lsm->i_step[0];
        int32_t na = lsm->i_nacc[0], ns = lsm->i_nsteps[0], un = lsm->i_until[0];
        metropolis_bookkeeping(st, na, ns, un, lsm->i_tune, lsm->i_tune_interval, accepted);
        lsm->i_step[0] = st; lsm->i_nacc[0] = na; lsm->i_nsteps[0] = ns; lsm->i_until[0] = un;
        trace_ic[(size_t)it * 2] = intercept[0];
        // the network log-likelihood of the stored state rides in the undirected model's unused
        // second intercept slot: the batched log-posterior pass reads it there, and the tie-break
        // of posterior_vi.py:62-80 (often thousands of identical partitions) needs no pass over
        // the stored positions
        trace_ic[(size_t)it * 2 + 1] = hs->ll;
    }
}

// ---- the intercept's likelihood pass on a queue of its own (undirected model; round 4) ----------------
// The intercept step (a chip-filling 31 us pass + one workgroup) reads the centred positions and
// nothing else of the iteration; the label update and the conjugate draws behind it (six launches of
// a few busy workgroups, ~60 us) do not read what it produces.  Two queues with event hand-overs
// lost more than they hid (profiles/r03_labels_notes.md); here the hand-overs are words in device
// memory and no event is recorded inside an iteration:
//   flags[HF_CENTRED]  ticket of the last iteration whose centred positions are final - stored by the
//                      label kernel at its entry (it follows the centring launch on the chain's queue,
//                      whose end-of-kernel release has made the positions visible to the device);
//                      k_hdp_gate, ahead of the pass on the second queue, polls it
//   flags[HF_SETTLED]  ticket of the last iteration whose intercept step is done (k_hdp_intercept_fork:
//                      one thread stores the results, fences and stores the flag); polled by the
//                      iteration's last launch before it takes the intercept for the next sweep
//   *err               sticky: a bounded wait ran out of its budget.  A word of HOST memory mapped into the
//                      device's address space: the host reads it behind a synchronisation without a copy
//                      (a blocking 4-byte hipMemcpy per dlsm_synchronize was ~25 us of every call)
// Every waiter is enqueued (host order) after the launch that stores what it waits for, so a wait ends
// even when the runtime has mapped both queues onto one hardware queue, where launches start in that
// order; the poll budget and the error word are the net under that argument.
enum : int { HF_CENTRED = 0, HF_SETTLED = 1 };
struct HdpFork { int32_t *flags; int32_t ticket; int32_t budget; int32_t *err; };

__device__ __forceinline__ void hdp_fork_wait(const HdpFork &f, int which) {
    for (int n = 0; n < f.budget; ++n) {
        if (coh_load_i32(f.flags + which) - f.ticket >= 0) return;
        __builtin_amdgcn_s_sleep(4);
    }
    __hip_atomic_fetch_or(f.err, 1 << which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(64) void k_hdp_gate(HdpFork f, int which) {
    if (threadIdx.x == 0) hdp_fork_wait(f, which);
}


// Two chains of work follow the label update and meet only in k_hdp_hypers: tables -> override
// variables, m_bar, beta, w0 -> w  and  MEAN sums -> mu + RESIDUAL sums -> sigma + LAMBDA sums.
// Their k-th links share a launch (workgroup ranges = roles), together with the hyper-parameter
// draws that are ready by then: 3 launches of ~5 / 20 / 15 us instead of 6.
template <int D>
__global__ __launch_bounds__(HDP_THREADS) void k_hdp_stage1(ChainView c, HdpLoopBuf hb,
                                                            HdpDeviceState *hs, LsmDeviceState *lsm,
                                                            const double *__restrict__ partials,
                                                            int nrec, double *__restrict__ intercept,
                                                            double *__restrict__ trace_ic, IterRef ir,
                                                            uint8_t *__restrict__ trace_row) {
    DLSM_HDP_STAMP(0)
    __builtin_amdgcn_s_setprio(3);          // ahead of the second queue's likelihood pass on a shared SIMD
    extern __shared__ int32_t sHist[];          // K K + K (the counts + tables role)
    const int K = hb.K, T = c.T;
    const int n_tab = T * hdp_tab_groups(K);
    if ((int)blockIdx.x < n_tab) { hdp_counts_tables_wg(c, hb, hs, ir.get(), blockIdx.x, sHist, trace_row); return; }
    if ((int)blockIdx.x == n_tab + K * T) {
        hdp_intercept_wg<4>(partials, nrec, lsm, hs, intercept, trace_ic, (int)ir.get());
        return;
    }
    const int q = (int)blockIdx.x - n_tab, k = q % K, t = q / K;
    double mk[D];
#pragma unroll
    for (int d = 0; d < D; ++d) mk[d] = 0.0;
    hdp_label_sums_wg<D, HDP_SUMS_MEAN>(c, k, t, mk, 1.0, hs->lmbda, 0.0, 0.0, nullptr, hb.S);
}

// the intercept step behind its pass on the second queue: the role of stage 1's last workgroup, then
// the hand-over (the stores above are thread 0's: drained, released to the device, then the flag)
__global__ __launch_bounds__(256) void k_hdp_intercept_fork(const double *__restrict__ partials, int nrec,
                                                            LsmDeviceState *lsm, HdpDeviceState *hs,
                                                            double *__restrict__ intercept,
                                                            double *__restrict__ trace_ic, int it, HdpFork f,
                                                            int which_flag) {
    hdp_intercept_wg(partials, nrec, lsm, hs, intercept, trace_ic, it);
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (which_flag >= 0) coh_store_i32(f.flags + which_flag, f.ticket);
    }
}

// the word that says "everything ahead of this launch on its queue has ended" (a kernel boundary lies
// between: what those launches stored is visible to whoever starts behind the flag)
__global__ __launch_bounds__(64) void k_fork_set(HdpFork f, int which) {
    if (threadIdx.x == 0) coh_store_i32(f.flags + which, f.ticket);
}

// consumer side, one lane: the settled intercept (poll, acquire, then a plain load)
__device__ __forceinline__ void hdp_fork_acquire_settled(const HdpFork &f) {
    hdp_fork_wait(f, HF_SETTLED);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int D>
__global__ __launch_bounds__(HDP_THREADS) void k_hdp_stage2(ChainView c, HdpLoopBuf hb,
                                                            HdpDeviceState *hs, IterRef ir) {
    DLSM_HDP_STAMP(1)
    __builtin_amdgcn_s_setprio(3);          // ahead of the second queue's likelihood pass on a shared SIMD
    const int K = hb.K;
    if (blockIdx.x == 0) { hdp_globals_wg(c, hb, hs, ir.get()); return; }
    if (blockIdx.x == 1) { hdp_akgrid_wg(c, hb, hs, ir.get()); return; }
    const int q = (int)blockIdx.x - 2;
    hdp_mu_residual_wg<D>(c, hb, hs, ir.get(), q % K, q / K);
}

template <int D>
__global__ __launch_bounds__(HDP_THREADS) void k_hdp_stage3(ChainView c, HdpLoopBuf hb,
                                                            const HdpDeviceState *hs, IterRef ir) {
    DLSM_HDP_STAMP(2)
    __builtin_amdgcn_s_setprio(3);          // ahead of the second queue's likelihood pass on a shared SIMD
    extern __shared__ double sGam[];            // K * K + K (the weights' role)
    const int K = hb.K, T = c.T;
    const int nw = HW_SPLIT * (T - 1);
    if ((int)blockIdx.x < nw) {
        hdp_weights_wg(c, hb, hs, ir.get(), (int)blockIdx.x / HW_SPLIT + 1, (int)blockIdx.x % HW_SPLIT, sGam);
        return;
    }
    if ((int)blockIdx.x == nw) { hdp_gam8_wg(c, hb, hs, ir.get()); return; }
    const int q = (int)blockIdx.x - nw - 1;
    hdp_sigma_lambda_wg<D>(c, hb, hs, ir.get(), q % K, q / K);
}

// one term of a Dirichlet log-density (hdp_lpcm.py:1193-1203; distributions.py:95-100: alphas and
// x are clipped at the smallest normal number)
__device__ __forceinline__ double dirichlet_term(double al, double x) {
    if (al <= 0.0) al = HDP_SMALL_EPS;
    if (x <= 0.0) x = HDP_SMALL_EPS;
    return (al - 1.0 == 0.0 ? 0.0 : (al - 1.0) * log(x)) - lgamma(al);
}

// ---- log-posterior trace (hdp_lpcm.py:1188-1280), batched over the stored samples ---------------
// Nothing in an iteration needs the sample's log-posterior, and everything it is computed from is
// in the trace (positions, labels as bytes, mu, sigma, beta, w, lambda, the six resampled
// hyper-parameters, the intercept and - second intercept slot - the network log-likelihood).  So
// dlsm_hdp_run computes it afterwards for all the rows it produced: the two launches that did it
// inside the iteration (label sums + Dirichlet rows, then a single-workgroup reduction: 23 us of
// the 114 us tail at config 3) become chip-wide passes over the trace.  Same terms, same order of
// summation per sample as the per-iteration kernels had (hdp_label_sums_wg<LOGP>, the reduction
// of k_hdp_finalize): the values are bit for bit the old ones.
struct HdpTraceView {
    const double *X;       // [n][T][N][D]
    const uint8_t *z;      // [n][T][N]
    const double *ic, *mu, *sigma, *beta, *w, *lambda, *hyper;
    double *logp;
};

__device__ __forceinline__ double hdp_dirichlet_row_at(const double *beta, const double *w,
                                                       const double *hy, int K, int j, int t, int lane);

// grid (K, T, samples): LP[s][t][k] = the node terms of cluster k at time t of sample s0 + s;
// cnt = its members; LPD[s][t][k] = the Dirichlet log-density of row (k, t) of the sample's transition
// weights, by the workgroup's last wavefront (round 6: the finish kernel computed a sample's T K rows on
// eight wavefronts one after the other - 92 us for one sample or five hundred, the fixed cost of every call
// of dlsm_hdp_run; here they are 4000 wavefronts' worth of a launch that has as many workgroups)
template <int D>
__global__ __launch_bounds__(HDP_THREADS) void k_hdp_logp_batch_sums(ChainView c, HdpTraceView tv,
                                                                     int s0, double a_,
                                                                     double *__restrict__ LP,
                                                                     int32_t *__restrict__ cnt,
                                                                     double *__restrict__ LPD) {
    __shared__ double buf[HDP_THREADS / 64];
    __shared__ int sC[HDP_THREADS / 64];
    const int k = blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
    const size_t s = (size_t)s0 + blockIdx.z;
    const int N = c.N, K = c.K, T = c.T;
    if ((tid >> 6) == HDP_THREADS / 64 - 1) {
        const double v = hdp_dirichlet_row_at(tv.beta + s * K, tv.w + s * T * K * K, tv.hyper + s * 6, K, k, t,
                                              tid & 63);
        if ((tid & 63) == 0) LPD[((size_t)blockIdx.z * T + t) * K + k] = v;
    }
    const uint8_t *zt = tv.z + (s * T + t) * N;
    const uint8_t *zp = t > 0 ? tv.z + (s * T + t - 1) * N : nullptr;
    const double *Xt = tv.X + (s * T + t) * (size_t)N * D;
    const double *Xp = t > 0 ? tv.X + (s * T + t - 1) * (size_t)N * D : nullptr;
    const double *w = tv.w + s * T * K * K;
    double mk[D];
#pragma unroll
    for (int d = 0; d < D; ++d) mk[d] = tv.mu[(s * K + k) * D + d];
    const double sk = tv.sigma[s * K + k], lm = tv.lambda[s], hb_ = tv.hyper[s * 6 + 5];
    const double lsk = log(sk);
    double acc = 0.0;
    int members = 0;
    // (four nodes per thread and trip, labels then positions requested together; a thread adds its
    // nodes in ascending order as before)
    constexpr int NU = 4;
    for (int i0 = tid; i0 < N; i0 += NU * HDP_THREADS) {
        int zz[NU], zq[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int i = i0 + u * HDP_THREADS;
            zz[u] = i < N ? (int)zt[i] : -1;
            zq[u] = i < N && t > 0 ? (int)zp[i] : 0;
        }
        double xs[NU][D], xps[NU][D], lw[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int i = i0 + u * HDP_THREADS;
            if (zz[u] == k) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    xs[u][d] = Xt[(size_t)i * D + d];
                    xps[u][d] = t > 0 ? Xp[(size_t)i * D + d] : 0.0;
                }
                lw[u] = w[((size_t)t * K + zq[u]) * K + k];
            }
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (zz[u] != k) continue;
            ++members;
            double ss = 0.0;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const double x = xs[u][d], xp = xps[u][d];
                const double r = t > 0 ? x - (1 - lm) * xp - lm * mk[d] : x - mk[d];
                ss += r * r;
            }
            acc += log(lw[u]) - 0.5 * lsk - 0.5 * ss / sk - (0.5 * a_ + 1.0) * lsk - 0.5 * hb_ / sk;
        }
    }
    const double v = block_sum_all<HDP_THREADS / 64>(acc, buf, tid);
    const double m = wave_sum_all((double)members);
    if ((tid & 63) == 0) sC[tid >> 6] = (int)m;
    __syncthreads();
    if (tid == 0) {
        const size_t q = ((size_t)blockIdx.z * T + t) * K + k;
        LP[q] = v;
        cnt[q] = (sC[0] + sC[1]) + (sC[2] + sC[3]);
    }
}

// Dirichlet log-density of row (j, t) of sample s (hdp_lpcm.py:1193-1203; distributions.py:95-100:
// alphas and x clipped at the smallest normal number): one wavefront, lanes = components
__device__ __forceinline__ double hdp_dirichlet_row_at(const double *beta, const double *w,
                                                       const double *hy, int K, int j, int t, int lane) {
    double total = 0.0;
    if (t >= 1) {
        double al = 0.0, v = 0.0;
        if (lane < K) {
            al = hy[2] * beta[lane] + (j == lane ? hy[3] : 0.0);       // alpha beta + kappa delta
            if (al <= 0.0) al = HDP_SMALL_EPS;
            v = dirichlet_term(al, w[((size_t)t * K + j) * K + lane]);
        }
        total = lgamma(wave_sum_all(al)) + wave_sum_all(v);
    } else if (j == 0) {
        double al0 = 0.0, al1 = 0.0, v0 = 0.0, v1 = 0.0;
        if (lane < K) {
            al0 = hy[0] / K;                                           // gamma / K
            al1 = hy[1] * beta[lane];                                  // alpha_init beta
            if (al0 <= 0.0) al0 = HDP_SMALL_EPS;
            if (al1 <= 0.0) al1 = HDP_SMALL_EPS;
            v0 = dirichlet_term(al0, beta[lane]);
            v1 = dirichlet_term(al1, w[lane]);
        }
        total = (lgamma(wave_sum_all(al0)) + wave_sum_all(v0)) +
                (lgamma(wave_sum_all(al1)) + wave_sum_all(v1));
    }
    return total;
}

// grid (samples): k_hdp_finalize's reduction and scalars over the sample's node terms and Dirichlet rows
// (k_hdp_logp_batch_sums; the sums keep their order: thread q adds what it added when the rows were computed here)
constexpr int HF_THREADS = 256;
template <int D>
__global__ __launch_bounds__(HF_THREADS) void k_hdp_logp_batch_finish(ChainView c, HdpTraceView tv,
                                                                      int s0, const HdpDeviceState *hs,
                                                                      const LsmDeviceState *lsm,
                                                                      const double *__restrict__ LP,
                                                                      const int32_t *__restrict__ cnt,
                                                                      const double *__restrict__ LPD) {
    __shared__ double red[HF_THREADS / 64];
    const int K = c.K, T = c.T, tid = threadIdx.x;
    const size_t s = (size_t)s0 + blockIdx.x;
    const double *hy = tv.hyper + s * 6;
    double acc = 0.0;
    for (int q = tid; q < T * K; q += 256) {
        const size_t g = (size_t)blockIdx.x * T * K + q;
        acc += (cnt[g] > 0 ? LP[g] : 0.0) + LPD[g];
    }
    const double body = block_sum_all<HF_THREADS / 64>(acc, red, tid);
    if (tid != 0) return;
    // + the network log-likelihood: the undirected model's second intercept slot; the directed
    // models use both slots and park it in the row's log-posterior slot (k_dir_tail)
    const bool directed = c.model != DLSM_UNDIRECTED;
    double lp = body + (directed ? tv.logp[s] : tv.ic[s * 2 + 1]);
    {   // intercept prior, cluster means, blending coefficient, hyper-priors
        const double b = tv.ic[s * 2], diff = b - lsm->intercept_prior[0];
        if (directed) {                 // hdp_lpcm.py:1234-1237, :1268-1269 (flat Dirichlet: log Gamma(N))
            const double diff1 = tv.ic[s * 2 + 1] - lsm->intercept_prior[1];
            lp -= (0.5 * (diff * diff) / lsm->intercept_var + 0.5 * (diff1 * diff1) / lsm->intercept_var);
            lp += lgamma((double)c.N);
        } else
        lp -= 0.5 * (diff * diff) / lsm->intercept_var;
        double ss = 0.0;
        for (int q = 0; q < K * D; ++q) ss += tv.mu[s * K * D + q] * tv.mu[s * K * D + q];
        const double mvp = hy[4], hb_ = hy[5];
        lp -= 0.5 * ss / mvp;
        lp += dev_truncnorm_logpdf(tv.lambda[s], hs->lambda_prior, hs->lambda_var);
        if (hs->has_a0) lp += -(0.5 * hs->a0 + 1.0) * log(mvp) - (0.5 * hs->b0 / mvp);
        if (hs->has_c0) lp += (hs->c0 - 1.0) * log(hb_) - hs->d0 * hb_;
    }
    tv.logp[s] = lp;
}

// labels of the sample -> its trace row as bytes (K <= 64)
__global__ __launch_bounds__(256) void k_hdp_trace_labels(const int32_t *__restrict__ z, long n,
                                                          uint8_t *__restrict__ row) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) row[i] = (uint8_t)z[i];
}

}  // namespace dlsm
