// Label block update: per node, backward messages
// over K states and T steps, then forward categorical sampling.
// One wave per node, lanes over the K mixture components (K <= 64); the T x K
// tables of a node live in that wave's slice of LDS.
#pragma once
#include "chain.hpp"
#include "device_common.hpp"

namespace dlsm {

// gaussian_likelihood_fast.pyx:17-27
template <int D>
__device__ __forceinline__ double spherical_normal_log_pdf(const double *x,
                                                           const double *mean,
                                                           double var) {
    double ss = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) ss += (x[k] - mean[k]) * (x[k] - mean[k]);
    ss *= 0.5 * (1. / var);
    return -0.5 * D * log(2 * 3.14159265358979323846 * var) - ss;
}

// log N(X[t, i]; m_tk, sigma_k) for component k (gaussian_likelihood_fast.pyx:44-49)
template <int D>
__device__ __forceinline__ double gauss_loglik_tk(const ChainView &c, int t, int i,
                                                  int k) {
    const double *x = c.X + ((size_t)t * c.N + i) * D;
    const double *m = c.mu + (size_t)k * D;
    if (t == 0) return spherical_normal_log_pdf<D>(x, m, c.sigma[k]);
    const double *xp = c.X + ((size_t)(t - 1) * c.N + i) * D;
    const double lm = c.lmbda_p[0];
    double muk[D];
#pragma unroll
    for (int j = 0; j < D; ++j) muk[j] = lm * m[j] + (1 - lm) * xp[j];
    return spherical_normal_log_pdf<D>(x, muk, c.sigma[k]);
}

// a11 seam: T x K table of one node (one wave)
template <int D>
__global__ __launch_bounds__(64) void k_gauss_table(ChainView c, int node,
                                                    int normalize,
                                                    double *__restrict__ out) {
    const int lane = threadIdx.x;
    for (int t = 0; t < c.T; ++t) {
        double v = lane < c.K ? gauss_loglik_tk<D>(c, t, node, lane) : -INFINITY;
        if (normalize) {
            double m = v;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
            v -= m;
        }
        if (lane < c.K) out[t * c.K + lane] = exp(v);
    }
}

// a12: sample_labels.py:134-190.  Dynamic LDS: the T transition matrices (rows padded
// to an odd length, shared by the workgroup; WLDS = false reads them from global memory
// when they do not fit) and per wave 2*T*K doubles (likelihood, partial marginal).  Sums that decide a draw run in the reference's index order so that the
// oracle (same Philox uniform) gets the same label.
constexpr int LAB_WAVES = 8;

__host__ __device__ inline int lab_row_pad(int K) { return K | 1; }

#ifdef DLSM_PIPE_TIMING
__device__ unsigned long long g_lab_t[4096][6];    // per wavefront: phase stamps (profiles/labels_phases.py)
#define DLSM_LAB_STAMP(I_, DEP_) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(DEP_)); lts[I_] = t_; }
#else
#define DLSM_LAB_STAMP(I_, DEP_)
#endif
template <int D, bool WLDS>
__global__ __launch_bounds__(64 * LAB_WAVES) void k_sample_labels(
    ChainView c, const double *__restrict__ w, uint32_t iter,
    int32_t *__restrict__ z_out) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int T = c.T, K = c.K, N = c.N;
    const int KP = WLDS ? lab_row_pad(K) : K;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * LAB_WAVES + wave;
#ifdef DLSM_PIPE_TIMING
    unsigned long long lts[6] = {0, 0, 0, 0, 0, 0};
#endif
    DLSM_LAB_STAMP(0, (double)lane)
    const double *wt = w;                     // row (t, j) at wt + (t K + j) KP
    double *tables = smem;
    if (WLDS) {
        for (int q = threadIdx.x; q < T * K * K; q += 64 * LAB_WAVES) {
            const int r = q / K;
            smem[(size_t)r * KP + (q - r * K)] = w[q];
        }
        wt = smem;
        tables = smem + (size_t)T * K * KP;
    }
    DLSM_LAB_STAMP(1, (double)lane)
    double *L = tables + (size_t)wave * 2 * T * K;
    double *pm = L + T * K;
    const bool live = i < N;                  // a whole wave is live or not
    if (live)                                 // the T x K table, 64 entries at a time
        for (int q = lane; q < T * K; q += 64) {
            const int t = q / K;
            L[q] = exp(gauss_loglik_tk<D>(c, t, i, q - t * K));
        }
    __syncthreads();
    if (!live) return;
    DLSM_LAB_STAMP(2, (double)lane)
    // the uniform of time t is drawn by lane t (T <= 64 per pass), all times at once
    double u_all = 0.0;
    // Lane k owns component k.  The sums over the components run in index order (as the
    // reference's do) with the k-th term fetched from lane k's register, not through LDS.
    // backward messages :164-170
    double bmk = 1.0;                         // message of component `lane` at time t
    for (int t = T - 1; t > 0; --t) {
        const double pmk = lane < K ? L[t * K + lane] * bmk : 0.0;
        if (lane < K) pm[t * K + lane] = pmk;          // the forward pass needs it again
        double s = 0.0;
        const double *wr = wt + ((size_t)t * K + min(lane, K - 1)) * KP;
        // the lane's row of w, eight LDS reads in flight per trip (one read per term of the
        // chain would pay the LDS latency K times per step); same terms, same order
        // (no test of k0 + u < K inside a trip: lanes >= K hold pmk = 0, the clamped read is a
        // finite weight, and adding their product, +0, changes nothing.  A step is 1.4 us of one
        // wavefront's chain - profiles/labels_phases.py; requesting the next step's row and table
        // entry a step ahead changed nothing: 38.7 against 39.3 us for the two label kernels)
        for (int k0 = 0; k0 < K; k0 += 8) {
            double wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = wr[min(k0 + u, K - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += wv[u] * lane_value(pmk, (k0 + u) & 63);
        }
        double tot = 0.0;
        for (int r0 = 0; r0 < K; r0 += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double v = lane_value(s, (r0 + u) & 63);
                tot += r0 + u < K ? v : 0.0;
            }
        }
        bmk = s / tot;
    }
    if (lane < K) pm[lane] = L[lane] * bmk;
    DLSM_LAB_STAMP(3, bmk)
    __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): LDS writes landed
    __builtin_amdgcn_wave_barrier();
    // forward sampling :173-188: the cumulative sum runs in index order and lane k keeps its
    // k-th value; the label is the number of cumulative values below u * total
    int zprev = 0;
    for (int t = 0; t < T; ++t) {
        const double *wrow = wt + (t == 0 ? (size_t)0 : ((size_t)t * K + zprev) * KP);
        if ((t & 63) == 0) {
            double u1;
            philox_uniform2(c.seed, (uint32_t)i, (uint32_t)(t + lane), iter,
                            stream_word(c.chain, STREAM_LABELS), u_all, u1);
        }
        const double u0 = lane_value(u_all, t & 63);
        const double term = lane < K ? wrow[lane] * pm[t * K + lane] : 0.0;
        double cdf = 0.0, mine = 0.0;
        for (int k0 = 0; k0 < K; k0 += 8) {            // lanes >= K hold term = 0: no test per k
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                cdf += lane_value(term, (k0 + u) & 63);
                mine = k0 + u == lane ? cdf : mine;
            }
        }
        const double u = u0 * cdf;
        const int zt = __popcll(__ballot(lane < K && u > mine));
        if (lane == 0) z_out[(size_t)t * N + i] = zt;
        zprev = zt;
    }
#ifdef DLSM_PIPE_TIMING
    DLSM_LAB_STAMP(4, (double)zprev)
    if (lane == 0 && i < 4096) for (int q = 0; q < 5; ++q) g_lab_t[i][q] = lts[q];
#endif
}

// ---- a12 on the f64 matrix cores: 16 nodes per workgroup ----
// The backward recursion of one node is a chain of T - 1 matrix-vector products with the same
// matrices for every node: for 16 nodes at once a step is  S[n][j] = sum_k P[n][k] w[t][j][k],
// a (16 x K)(K x K) product, ceil(K/4) x ceil(K/16) v_mfma_f64_16x16x4_f64 per step (operand
// placement probed on the hardware, profiles/micro/mfma_f64_layout.hip: A[i = lane % 16][k =
// lane / 16], B[k = lane / 16][j = lane % 16], D[i = 4 r + lane / 16][j = lane % 16]).  The
// wavefront-per-node kernel above spends 1.4 us per step on a K-term chain of register fetches;
// here a step of 16 nodes is one burst of LDS reads, ten matrix instructions, a rescale and one
// trip through LDS that turns the result's placement into the next step's A operand.
//   * KS = ceil(K / 4) is a template parameter: every loop over components is unrolled without a
//     guard, the rows in LDS are padded with zeros up to 4 KS components instead;
//   * the whole workgroup builds the 16 nodes' T x K likelihood table (same expression as
//     gauss_loglik_tk, the -d/2 log(2 pi sigma_k) term computed once per component) and the
//     Philox uniforms (counter = (node, t, iteration), as above); wavefront 0 then runs the
//     recursion and the draws;
//   * the messages are rescaled by a power of two per node and step (the largest exponent of the
//     row) instead of divided by the row's sum: the scale cancels in the draw (u * total against
//     the cumulative sums), a power of two rounds nothing, and the row maximum - unlike a sum -
//     is the same number in every lane whatever the order of the reduction;
//   * forward draws: four lanes per node, each the cumulative sums of KS components in index
//     order; the quarters' offsets are (c0), (c0 + c1), ((c0 + c1) + c2).
// The sums over k run inside the matrix instruction and over quarters here, the reference's and
// the kernel above's in one chain: the messages and cumulative sums differ from theirs in the
// last bits, so a label can differ when u * total lies within a few ulp of a cumulative sum
// (about K 1e-16 per draw; the parity tests compare labels exactly over thousands of draws).
constexpr int LM_THREADS = 1024;
constexpr int LM_NODES = 8;             // nodes per workgroup: rows 8-15 of the A operand repeat rows 0-7
typedef double lm_v4d __attribute__((ext_vector_type(4)));
constexpr int LM_MAX_KS = 8;          // K <= 32: the unrolled forms that stay in registers (capi.hip)
__host__ __device__ inline int lm_ksteps(int K) { return (K + 3) / 4; }
__host__ __device__ inline int lm_wstride(int K) { return (4 * lm_ksteps(K)) | 1; }
__host__ __device__ inline int lm_stride(int K) { return 16 * ((lm_ksteps(K) + 3) / 4) + 1; }
__host__ __device__ inline size_t lm_lds_bytes(int T, int K) {
    return ((size_t)T * K * lm_wstride(K) + (size_t)(T + 1) * LM_NODES * lm_stride(K) + LM_NODES * (size_t)T + 2 * K) *
           sizeof(double);
}
__host__ __device__ inline size_t lm_lds_bytes(int T, int K, int D) {
    return lm_lds_bytes(T, K) + ((size_t)T * LM_NODES + K) * D * sizeof(double);
}

template <int KS>
__global__ __launch_bounds__(LM_THREADS) void k_sample_labels_mfma(
    ChainView c, const double *__restrict__ w, uint32_t iter, int32_t *__restrict__ z_out,
    int32_t *flag, int32_t flag_val) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // (HDP-LPCM loop with the likelihood pass on a second queue: this launch follows the centring
    // launch on the chain's queue - the centred positions are final, the pass may start)
    if (flag && blockIdx.x == 0 && threadIdx.x == 0) coh_store_i32(flag, flag_val);
    __builtin_amdgcn_s_setprio(3);          // ahead of the second queue's likelihood pass on a shared SIMD
    constexpr int KT = (KS + 3) / 4, S = 16 * KT + 1, WS = (4 * KS) | 1;
    const int T = c.T, K = c.K, N = c.N, D = c.D;
    double *wt = smem;                                  // [T][K][WS] transition matrices, zero padded
    constexpr int NN = LM_NODES;
    double *tab = wt + (size_t)T * K * WS;              // [T][NN][S] likelihood, then partial marginal
    double *Mb = tab + (size_t)T * NN * S;              // [NN][S] the step's messages
    double *U = Mb + NN * S;                            // [NN][T] uniforms
    double *lognorm = U + NN * T;                       // [K]
    double *hiv = lognorm + K;                          // [K] 0.5 / sigma_k
    double *sx = hiv + K;                               // [T][NN][D] the nodes' positions
    double *smu = sx + (size_t)T * NN * D;              // [K][D]
    const int tid = threadIdx.x, lane = tid & 63;
    const int node0 = blockIdx.x * NN;
#ifdef DLSM_PIPE_TIMING
    unsigned long long lts[6] = {0, 0, 0, 0, 0, 0};
#endif
    DLSM_LAB_STAMP(0, (double)lane)
    // the transition matrices' first passes are requested before anything waits on a load
    constexpr int WPRE = 6;
    const int TKW = T * K * WS;
    double wv[WPRE];
#pragma unroll
    for (int u = 0; u < WPRE; ++u) {
        const int q = tid + u * LM_THREADS, r = q / WS, k = q - r * WS;
        wv[u] = q < TKW && k < K ? w[(size_t)r * K + k] : 0.0;
    }
    for (int q = tid; q < T * NN * D; q += LM_THREADS) {
        const int tn = q / D, j = q - tn * D;
        sx[q] = c.X[((size_t)(tn / NN) * N + min(node0 + (tn % NN), N - 1)) * D + j];
    }
    for (int q = tid; q < K * D; q += LM_THREADS) smu[q] = c.mu[q];
#pragma unroll
    for (int u = 0; u < WPRE; ++u) {
        const int q = tid + u * LM_THREADS;
        if (q < TKW) wt[q] = wv[u];
    }
    for (int q = tid + WPRE * LM_THREADS; q < TKW; q += LM_THREADS) {
        const int r = q / WS, k = q - r * WS;
        wt[q] = k < K ? w[(size_t)r * K + k] : 0.0;
    }
    for (int k = tid; k < K; k += LM_THREADS) {
        const double var = c.sigma[k];
        lognorm[k] = -0.5 * D * log(2 * 3.14159265358979323846 * var);
        hiv[k] = 0.5 * (1. / var);
    }
    for (int q = tid; q < NN * T; q += LM_THREADS) {
        const int n = q / T, t = q - n * T;
        double u0, u1;
        philox_uniform2(c.seed, (uint32_t)min(node0 + n, N - 1), (uint32_t)t, iter,
                        stream_word(c.chain, STREAM_LABELS), u0, u1);
        U[q] = u0;
    }
    for (int q = tid; q < NN * S; q += LM_THREADS) Mb[q] = 1.0;     // the message of time T - 1
    __syncthreads();
    DLSM_LAB_STAMP(1, (double)lane)
    {
        const double lm = c.lmbda_p[0];
        for (int q = tid; q < T * NN * 4 * KS; q += LM_THREADS) {
            const int tn = q / (4 * KS), k = q - tn * (4 * KS), t = tn / NN;
            double v = 0.0;                             // components K .. 4 KS - 1: padding
            if (k < K) {
                const double *x = sx + (size_t)tn * D;
                const double *m = smu + (size_t)k * D;
                double ss = 0.0;
                if (t == 0) {
                    for (int j = 0; j < D; ++j) ss += (x[j] - m[j]) * (x[j] - m[j]);
                } else {
                    const double *xp = x - (size_t)NN * D;
                    for (int j = 0; j < D; ++j) {
                        const double mk = lm * m[j] + (1 - lm) * xp[j];
                        ss += (x[j] - mk) * (x[j] - mk);
                    }
                }
                ss *= hiv[k];
                v = exp(lognorm[k] - ss);
            }
            tab[(size_t)tn * S + k] = v;
        }
    }
    __syncthreads();
    if (tid >= 64) return;                              // wavefront 0 carries on alone
    DLSM_LAB_STAMP(2, (double)lane)
    // backward messages (sample_labels.py:164-170)
    const int an = lane & 15, g = lane >> 4;            // A operand: node an % NN, component 4 s + g
    const int ar = an % NN;
    const bool own = an < NN;                           // the lanes that file what they compute
    // operands of a step that do not wait for the step before it (likelihood column, B operand)
    // are requested a step ahead, while the matrix instructions run
    double lk[KS], b[KS][KT];
    auto request = [&](int t) {
        const double *trow = tab + ((size_t)t * NN + ar) * S + g;
#pragma unroll
        for (int s = 0; s < KS; ++s) lk[s] = trow[4 * s];
#pragma unroll
        for (int ct = 0; ct < KT; ++ct) {               // B operand: w[t][j][k], rows past K are zero
            const int j = 16 * ct + an;
            const double *wrow = wt + ((size_t)t * K + min(j, K - 1)) * WS + g;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const double wv = wrow[4 * s];
                b[s][ct] = j < K ? wv : 0.0;
            }
        }
    };
    if (T > 1) request(T - 1);
    for (int t = T - 1; t > 0; --t) {
        double *trow = tab + ((size_t)t * NN + ar) * S + g;
        const double *mrow = Mb + ar * S + g;
        double a[KS], bb[KS][KT];
#pragma unroll
        for (int s = 0; s < KS; ++s) a[s] = lk[s] * mrow[4 * s];
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int ct = 0; ct < KT; ++ct) bb[s][ct] = b[s][ct];
#pragma unroll
        for (int s = 0; s < KS; ++s)
            if (own) trow[4 * s] = a[s];                            // the forward pass needs it again
        if (t > 1) request(t - 1);
        lm_v4d acc[KT];
#pragma unroll
        for (int ct = 0; ct < KT; ++ct) acc[ct] = lm_v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int ct = 0; ct < KT; ++ct)
                acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], bb[s][ct], acc[ct], 0, 0, 0);
        // lane holds nodes 4 r + g, components 16 ct + an: one power of two per node, from the
        // largest biased exponent of the row (the sums are >= 0: the high word orders them; the
        // four nodes' reductions are written side by side so that their DPP waits interleave)
        constexpr int NR = NN / 4;                      // result registers that hold real nodes
        int e[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            e[r] = __double2hiint(acc[0][r]);
#pragma unroll
            for (int ct = 1; ct < KT; ++ct) e[r] = max(e[r], __double2hiint(acc[ct][r]));
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) e[r] = max(e[r], __builtin_amdgcn_mov_dpp(e[r], 0x128, 0xF, 0xF, false));  // row_ror:8
#pragma unroll
        for (int r = 0; r < NR; ++r) e[r] = max(e[r], __builtin_amdgcn_mov_dpp(e[r], 0x124, 0xF, 0xF, false));  // row_ror:4
#pragma unroll
        for (int r = 0; r < NR; ++r) e[r] = max(e[r], __builtin_amdgcn_mov_dpp(e[r], 0x122, 0xF, 0xF, false));  // row_ror:2
#pragma unroll
        for (int r = 0; r < NR; ++r) e[r] = max(e[r], __builtin_amdgcn_mov_dpp(e[r], 0x121, 0xF, 0xF, false));  // row_ror:1
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int sh = 1023 - (e[r] >> 20);         // the row's largest entry lands in [1, 2)
#pragma unroll
            for (int ct = 0; ct < KT; ++ct)
                Mb[(4 * r + g) * S + 16 * ct + an] = ldexp(acc[ct][r], sh);
        }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s)
        if (own) tab[(size_t)ar * S + g + 4 * s] *= Mb[ar * S + g + 4 * s];
    DLSM_LAB_STAMP(3, (double)lane)
    __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): LDS writes landed
    __builtin_amdgcn_wave_barrier();
    // forward sampling (:173-188): four lanes per node, KS components each
    const int fl = lane >> 2, fn = fl % NN, q = lane & 3, node = node0 + fn;   // lanes 4 NN .. 63 shadow the first
    int zprev = 0;
    double pv[KS];
#pragma unroll
    for (int e = 0; e < KS; ++e) pv[e] = tab[(size_t)fn * S + q * KS + e];
    double un = U[fn * T];
    for (int t = 0; t < T; ++t) {
        const double *wrow = wt + (t == 0 ? (size_t)0 : ((size_t)t * K + zprev) * WS) + q * KS;
        double pl[KS], run = 0.0;
#pragma unroll
        for (int e = 0; e < KS; ++e) pl[e] = wrow[e];
#pragma unroll
        for (int e = 0; e < KS; ++e) {
            run += pl[e] * pv[e];
            pl[e] = run;
        }
        const double ut0 = un;
        if (t + 1 < T) {                                // the next step's operands that do not wait for the label
#pragma unroll
            for (int e = 0; e < KS; ++e) pv[e] = tab[((size_t)(t + 1) * NN + fn) * S + q * KS + e];
            un = U[fn * T + t + 1];
        }
        const double c0 = dpp_move<0x00>(run), c1 = dpp_move<0x55>(run), c2 = dpp_move<0xAA>(run),
                     c3 = dpp_move<0xFF>(run);
        const double o2 = c0 + c1, o3 = o2 + c2, total = o3 + c3;
        const double off = q == 0 ? 0.0 : q == 1 ? c0 : q == 2 ? o2 : o3;
        const double ut = ut0 * total;
        int cnt = 0;                                    // padding repeats the quarter's last sum: never below u * total
#pragma unroll
        for (int e = 0; e < KS; ++e) cnt += ut > off + pl[e] ? 1 : 0;
        cnt += __builtin_amdgcn_mov_dpp(cnt, 0xB1, 0xF, 0xF, false);             // quad_perm [1, 0, 3, 2]
        cnt += __builtin_amdgcn_mov_dpp(cnt, 0x4E, 0xF, 0xF, false);             // quad_perm [2, 3, 0, 1]
        const int zt = min(cnt, K - 1);
        if (q == 0 && fl < NN && node < N) z_out[(size_t)t * N + node] = zt;
        zprev = zt;
    }
#ifdef DLSM_PIPE_TIMING
    DLSM_LAB_STAMP(4, (double)zprev)
    if (lane == 0 && blockIdx.x < 4096) for (int qq = 0; qq < 5; ++qq) g_lab_t[blockIdx.x][qq] = lts[qq];
#endif
}

// The counts the conjugate updates need (sample_labels.py:176-188): n[0][0][k] initial labels,
// n[t][j][k] transitions j -> k into time t, nk[t][k] labels in use.  One workgroup per time
// step, histogram in LDS (same-address global atomics from 2000 wavefronts cost ~170 ns each).
// `trace_row` (may be NULL): the labels are also filed as bytes, [T][N], the device-resident
// HDP-LPCM loop's trace row of this sample.
constexpr int LC_THREADS = 1024;
__global__ __launch_bounds__(LC_THREADS) void k_label_counts(const int32_t *__restrict__ z, int N, int K,
                                                             int32_t *__restrict__ n_cnt,
                                                             int32_t *__restrict__ nk_cnt,
                                                             uint8_t *__restrict__ trace_row) {
    extern __shared__ int32_t hist[];         // K * K + K
    const int t = blockIdx.x;
    // the first two passes' labels are requested before the histogram is cleared (a pass is one
    // memory round trip: 2 of them at N = 2000 instead of 8 with 256 threads)
    constexpr int PRE = 2;
    int zt[PRE], zp[PRE];
#pragma unroll
    for (int u = 0; u < PRE; ++u) {
        const int i = threadIdx.x + u * LC_THREADS;
        zt[u] = i < N ? z[(size_t)t * N + i] : 0;
        zp[u] = i < N && t > 0 ? z[(size_t)(t - 1) * N + i] : 0;
    }
    for (int q = threadIdx.x; q < K * K + K; q += LC_THREADS) hist[q] = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PRE; ++u) {
        const int i = threadIdx.x + u * LC_THREADS;
        if (i < N) {
            atomicAdd(&hist[zp[u] * K + zt[u]], 1);
            atomicAdd(&hist[K * K + zt[u]], 1);
            if (trace_row) trace_row[(size_t)t * N + i] = (uint8_t)zt[u];
        }
    }
    for (int i = threadIdx.x + PRE * LC_THREADS; i < N; i += LC_THREADS) {
        const int a = z[(size_t)t * N + i];
        const int b = t == 0 ? 0 : z[(size_t)(t - 1) * N + i];
        atomicAdd(&hist[b * K + a], 1);
        atomicAdd(&hist[K * K + a], 1);
        if (trace_row) trace_row[(size_t)t * N + i] = (uint8_t)a;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < K * K; q += LC_THREADS) n_cnt[(size_t)t * K * K + q] = hist[q];
    for (int q = threadIdx.x; q < K; q += LC_THREADS) nk_cnt[t * K + q] = hist[K * K + q];
}

}  // namespace dlsm
