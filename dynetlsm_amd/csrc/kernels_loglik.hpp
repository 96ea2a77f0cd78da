// Network packing and log-likelihood kernels.
#pragma once
#include "chain.hpp"
#include "device_common.hpp"

namespace dlsm {

// ---------------------------------------------------------------------------
// pack: float64 adjacency -> 1 bit / dyad.  One wave reads 64 consecutive
// doubles of a row (coalesced 512 B) and ballots them into two uint32 words.
// transpose != 0 builds bit i of row j = Y[t, i, j].
// flag[0] |= 1 when an entry is neither 0.0 nor 1.0.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_bits(const double *__restrict__ Y,
                                                   int N, int W, int transpose,
                                                   uint32_t *__restrict__ bits,
                                                   int *__restrict__ flag) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int row = blockIdx.x;            // over T*N rows
    const int t = row / N, j = row % N;
    const int nseg = (W * 32) / 64;        // W is a multiple of 4 -> whole segments
    for (int seg = wave; seg < nseg; seg += 4) {
        int i = seg * 64 + lane;
        double y = 0.0;
        if (i < N)
            y = transpose ? Y[((size_t)t * N + i) * N + j]
                          : Y[((size_t)t * N + j) * N + i];
        if (y != 0.0 && y != 1.0) atomicOr(flag, 1);
        unsigned long long m = __ballot(y != 0.0);
        if (lane == 0) {
            bits[(size_t)row * W + seg * 2] = (uint32_t)m;
            bits[(size_t)row * W + seg * 2 + 1] = (uint32_t)(m >> 32);
        }
    }
}

// The undirected network's words column-block-major (ChainView::ycm): thread = (row i, 64-column
// block cb); a wavefront reads 64 rows' words (stride W) and writes 512 contiguous bytes.
__global__ __launch_bounds__(256) void k_pack_colmajor(const uint32_t *__restrict__ bits, int T, int N,
                                                       int W, int Ncm,
                                                       unsigned long long *__restrict__ ycm) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long per_t = (long)(W / 2) * Ncm;
    if (idx >= (long)T * per_t) return;
    const int t = (int)(idx / per_t);
    const long r = idx - (long)t * per_t;
    const int cb = (int)(r / Ncm), i = (int)(r - (long)cb * Ncm);
    unsigned long long v = 0ull;
    if (i < N) v = *(const unsigned long long *)(bits + ((size_t)t * N + i) * W + 2 * cb);
    ycm[idx] = v;
}

// Invariants of a packed network handed in ready-made (dlsm_set_network_packed): bits beyond
// N and the diagonal are zero, bit i of row j of `bits` equals bit j of row i of `tbits`
// (tbits = bits for the undirected model: symmetry).  One wavefront per row.
__global__ __launch_bounds__(64) void k_check_packed(const uint32_t *__restrict__ bits,
                                                     const uint32_t *__restrict__ tbits, int T,
                                                     int N, int W, int *__restrict__ flag) {
    const int lane = threadIdx.x;
    const size_t row = blockIdx.x;
    const int t = (int)(row / N), j = (int)(row % N);
    const uint32_t *r = bits + row * W;
    int bad = 0;
    for (int w = lane; w < W; w += 64) {
        const uint32_t v = r[w];
        const int lo = 32 * w;
        if (lo + 32 > N) {
            const uint32_t keep = lo >= N ? 0u : ((1u << (N - lo)) - 1u);
            if (v & ~keep) bad |= 1;
        }
        if (j >= lo && j < lo + 32 && ((v >> (j - lo)) & 1u)) bad |= 2;
    }
    for (int i = lane; i < N; i += 64) {
        const uint32_t a = (r[i >> 5] >> (i & 31)) & 1u;
        const uint32_t b = (tbits[((size_t)t * N + i) * W + (j >> 5)] >> (j & 31)) & 1u;
        if (a != b) bad |= 4;
    }
    if (bad) atomicOr(flag, bad);
}

// ---------------------------------------------------------------------------
// Full log-likelihood, undirected (a4) and directed (a5).
//
// Grid: T x (upper-triangular 128x128 tiles) [x 2 half tiles, undirected]; thread = one
// column j of the tile x one half (64) of its rows.  X tiles (and, directed, the tile's Y
// bits) are staged in LDS; X_i is an LDS broadcast, X_j stays in registers.
// Per-thread accumulators in fp64, wave shuffle -> LDS -> one partial record per
// workgroup; k_reduce_partials sums the records in a fixed order (deterministic).
//
// Undirected, candidate k:  ll_k = b_k*SY - SYd - sum log(1 + exp(b_k) e^{-d})
//   (SY = sum Y, SYd = sum Y d are intercept-independent: every candidate of the
//    intercept MH step and the log-posterior trace come out of ONE pass.)
// Directed, candidate k = (b_in, b_out, radii_k): both directions of a pair
//   share the distance.
// Record layout: [SY, SYd, S_0 .. S_{M-1}] (undirected), [L_0 .. L_{M-1}] (directed)
// ---------------------------------------------------------------------------
constexpr int LL_TILE = 128;
constexpr int LL_THREADS = 256;

struct LoglikCand {
    const double *intercepts;   // device, M x (1|2)
    const double *radii[2];     // device N, per candidate (directed)
};

__device__ __forceinline__ void tile_decode(int r, int nt, int &ti, int &tj) {
    // r-th tile of the upper triangle (ti <= tj), row-major
    ti = 0;
    int rowlen = nt;
    while (r >= rowlen) { r -= rowlen; ++ti; --rowlen; }
    tj = ti + r;
}

// Undirected: one workgroup (2 wavefronts) per HALF tile, 64 rows x 128 columns; thread =
// one column j (X_j in registers), the rows' X_i an LDS broadcast.  The 64 bits of row i under
// a wavefront's columns are one aligned 8-byte word of the packed network: read as a SCALAR
// and used as the lane mask of "y = 1" directly, and counted with a scalar popcount.
constexpr int LLU_THREADS = 128;
#ifndef LLU_ROWS_V
#define LLU_ROWS_V 64
#endif
constexpr int LLU_ROWS = LLU_ROWS_V;
constexpr int LLU_SPLIT = 128 / LLU_ROWS;         // workgroups per tile

#ifdef DLSM_PIPE_TIMING
__device__ unsigned long long g_ll_t[8192][3];     // per wavefront: entry, exit (100 MHz), HW_ID
#endif
// workgroups of the undirected pass per time step: two (row halves) per tile above the diagonal, ONE per
// diagonal tile (below) - nt (nt - 1) + nt = nt^2
__host__ __device__ inline int llu_blocks_per_slice(int nt) { return nt * nt; }
// Diagonal tiles.  Of a diagonal tile's four 64 x 64 blocks one is full (rows 0-63 x columns 64-127), two are
// triangles (i < j inside rows / columns 0-63 and inside 64-127) and one is empty; as two half-tile workgroups
// that was four wavefronts of 64 row steps each for 1.5 wavefronts of work - and 2720 workgroups at config 2,
// 5.3 wavefronts per SIMD: some SIMDs carried six, some four, and the launch lasted as long as the sixes
// (profiles/r04_loglik_timing.json).  Now a diagonal tile is one workgroup: wavefront 1 takes the full block,
// wavefront 0 BOTH triangles - at row step r its lanes j > r pair row r with column j (the first triangle), its
// lanes j < r pair row 127 - r with column 127 - j (the second one, mirrored: 127 - r < 127 - j), every lane but
// one busy at every step.  2560 workgroups at config 2: five wavefronts on every SIMD, all of the same length.
template <int D, int M>
__global__ __launch_bounds__(LLU_THREADS) __attribute__((amdgpu_waves_per_eu(D <= 3 ? 5 : (D == 4 ? 4 : 3), 8))) void k_loglik_undirected(
    ChainView c, LoglikCand cand, double *__restrict__ partials, int prio) {
    __shared__ double sXi[LL_TILE * D];
    __shared__ double sRed[2 * (2 + M)];
    __shared__ __attribute__((aligned(16))) double sTab[EXPTAB_N];      // tab_exp (device_common.hpp)
    const int tid = threadIdx.x;
    const int N = c.N;
#ifdef DLSM_PIPE_TIMING
    unsigned long long tl0;
    unsigned int hwid, xccid;
    asm volatile("s_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl0), "=s"(hwid), "=s"(xccid));
#endif
    static_assert(2 * LLU_THREADS == EXPTAB_N, "two table entries per thread");
    sTab[tid] = c_exp2_tab[tid];                   // visible after the staging barrier below
    sTab[tid + LLU_THREADS] = c_exp2_tab[tid + LLU_THREADS];
    const int nt = (N + LL_TILE - 1) / LL_TILE;
    static_assert(LLU_SPLIT == 2 && LLU_ROWS == 64 && LLU_THREADS == 128, "the diagonal tiles' pairing");
    const int per_t = llu_blocks_per_slice(nt), noff = nt * (nt - 1);     // (half tiles above the diagonal)
    const int t = blockIdx.x / per_t, wq = blockIdx.x - t * per_t;
    const bool diag = wq >= noff;
    int ti, tj, half = 0;
    if (diag) { ti = tj = wq - noff; }
    else {                                          // the (wq / 2)-th tile with ti < tj, row-major
        int r = wq >> 1, rowlen = nt - 1;
        ti = 0;
        while (r >= rowlen) { r -= rowlen; ++ti; --rowlen; }
        tj = ti + 1 + r;
        half = wq & 1;
    }
    const int i0 = ti * LL_TILE + half * LLU_ROWS, j0 = tj * LL_TILE;
    const double *Xt = c.X + (size_t)t * N * D;
    for (int k = tid; k < (diag ? LL_TILE : LLU_ROWS) * D; k += LLU_THREADS) {
        const int gi = i0 * D + k;
        sXi[k] = gi < N * D ? Xt[gi] : 0.0;
    }
    const int j = j0 + tid;
    double xj[D];
#pragma unroll
    for (int d = 0; d < D; ++d) xj[d] = j < N ? Xt[(size_t)j * D + d] : 0.0;
    double E[M];
#pragma unroll
    for (int k = 0; k < M; ++k) E[k] = fast_exp(cand.intercepts[k]);
    __syncthreads();

    // sum_i log(1 + E e^{-d_i}) = log prod_i (1 + E e^{-d_i}): one log per
    // `nflush` dyads (the running product stays far inside the double range)
    double Emax = E[0];
#pragma unroll
    for (int k = 1; k < M; ++k) Emax = fmax(Emax, E[k]);
    // (float precision is plenty for a flush interval with a margin of e^109)
    const float l1p = __logf(1.0f + (float)Emax);
    const int nflush = !(l1p > 0.0f) ? 64 : (600.0f / l1p < 1.0f ? 1 : (600.0f / l1p > 64.0f ? 64 : (int)(600.0f / l1p)));
    constexpr int U = 4;          // rows per trip, each with its own product chain
    double syd = 0.0, S[M], P[U][M];
    int sy = 0;                   // wave-uniform count of the edges seen
#pragma unroll
    for (int k = 0; k < M; ++k) {
        S[k] = 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) P[u][k] = 1.0;
    }
    // rows r < rend of this half tile pair with column j (i < j < N)
    const int rend = j < N ? min(LLU_ROWS, j - i0) : 0;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the row words of this wavefront's 64 columns, rows i0 .. : contiguous in the column-block-major
    // copy (row-major they are W words apart: every 8-byte scalar load dragged a 128-byte line in,
    // 25 MB of traffic per pass for 5 MB of bits; rows beyond N are zero there)
    const unsigned long long *ywords = c.ycm + ((size_t)t * (c.W >> 1) + (j0 >> 6) + wave) * c.Ncm + i0;
    // a half tile strictly above the diagonal with all its columns inside the network
    // needs no per-dyad validity test
    // (wavefront 1 of a diagonal tile - rows 0-63, columns 64-127 - is such a block too)
    const bool whole = (i0 + LLU_ROWS <= j0 || (diag && wave == 1)) && j0 + LL_TILE <= N;
    int cnt = 0;
    // issue priority by progress (as the sweep's items, kernels_spec_pipe.hpp): the wavefronts of a SIMD end together
#ifdef DLSM_LLU_NOPRIO
#define DLSM_LLU_PRIO(R_)
#else
    // (prio == 0: the pass stays at the default priority - the HDP-LPCM loop's second queue, where priority 3
    // belongs to the chain's small launches beside it: capi_hdp.hpp)
#define DLSM_LLU_PRIO(R_)                                                                      \
        if (prio) {                                                                            \
            if ((R_) == 0) __builtin_amdgcn_s_setprio(3);                                      \
            else if ((R_) == LLU_ROWS / 4) __builtin_amdgcn_s_setprio(2);                      \
            else if ((R_) == LLU_ROWS / 2) __builtin_amdgcn_s_setprio(1);                      \
            else if ((R_) == 3 * LLU_ROWS / 4) __builtin_amdgcn_s_setprio(0);                  \
        }
#endif
#define DLSM_LLU_TRIPS(WHOLE_, SQ_)                                                            \
    for (int r = 0; r < LLU_ROWS; r += U) {                                                    \
        DLSM_LLU_PRIO(r)                                                                       \
        unsigned long long ym[U];                                                              \
        _Pragma("unroll")                                                                      \
        for (int u = 0; u < U; ++u)                                                            \
            ym[u] = ywords[r + u];                                                             \
        double dd[U], e[U];                                                                    \
        _Pragma("unroll")                                                                      \
        for (int u = 0; u < U; ++u) dd[u] = dist_fast<D>(&sXi[(r + u) * D], xj, SQ_);          \
        _Pragma("unroll")                                                                      \
        for (int u = 0; u < U; ++u)                                                            \
            e[u] = (SQ_) ? tab_exp_clamped(-dd[u], sTab) : tab_exp(-dd[u], sTab);              \
        _Pragma("unroll")                                                                      \
        for (int u = 0; u < U; ++u) {                                                          \
            const bool ok = WHOLE_ || r + u < rend;                                            \
            const unsigned long long live = WHOLE_ ? ym[u] : (ym[u] & __ballot(ok));           \
            sy += __popcll(live);                                                              \
            syd = fma(__builtin_amdgcn_inverse_ballot_w64(live) ? 1.0 : 0.0, dd[u], syd);      \
            if (ok) {                                                                          \
                _Pragma("unroll")                                                              \
                for (int k = 0; k < M; ++k) P[u][k] *= fma(E[k], e[u], 1.0);                   \
            }                                                                                  \
        }                                                                                      \
        if (++cnt >= nflush) {                                                                 \
            _Pragma("unroll")                                                                  \
            for (int k = 0; k < M; ++k) {                                                      \
                _Pragma("unroll")                                                              \
                for (int u = 0; u < U; ++u) { S[k] += fast_log(P[u][k]); P[u][k] = 1.0; }           \
            }                                                                                  \
            cnt = 0;                                                                           \
        }                                                                                      \
    }
    // wavefront 0 of a diagonal tile: both triangles (header).  Lane l holds column j0 + l (xj) for the first
    // and column j0 + 127 - l (xjb) for the mirrored second; row words: the first triangle's are this
    // wavefront's (ywords), the second's those of column block + 1 at rows i0 + 64 .. (bit 63 - l of the
    // word of row 127 - r = bit l of its bit reversal)
#define DLSM_LLU_TRIPS_DIAG(SQ_)                                                               \
    {                                                                                          \
        const int lane_ = tid;                                                                 \
        const int jb_ = j0 + (LL_TILE - 1) - lane_;                                            \
        double xjb[D];                                                                         \
        _Pragma("unroll")                                                                      \
        for (int d = 0; d < D; ++d) xjb[d] = jb_ < N ? Xt[(size_t)jb_ * D + d] : 0.0;          \
        const unsigned long long va_ = __ballot(j < N), vb_ = __ballot(jb_ < N);               \
        const unsigned long long *ywb = ywords + c.Ncm + LLU_ROWS;                             \
        for (int r = 0; r < LLU_ROWS; r += U) {                                                \
            DLSM_LLU_PRIO(r)                                                                   \
            unsigned long long ya[U], yb[U];                                                   \
            _Pragma("unroll")                                                                  \
            for (int u = 0; u < U; ++u) { ya[u] = ywords[r + u]; yb[u] = ywb[LLU_ROWS - 1 - (r + u)]; } \
            double dd[U], e[U];                                                                \
            unsigned long long okm[U];                                                         \
            _Pragma("unroll")                                                                  \
            for (int u = 0; u < U; ++u) {                                                      \
                const int rr = r + u;                                                          \
                const unsigned long long lo_ = (1ull << rr) - 1ull;                            \
                const unsigned long long hi_ = rr == 63 ? 0ull : ~((2ull << rr) - 1ull);       \
                const bool first_ = lane_ > rr;                                                \
                const double *xi_ = sXi + (first_ ? rr : LL_TILE - 1 - rr) * D;                \
                double xs_[D];                                                                 \
                _Pragma("unroll")                                                              \
                for (int d = 0; d < D; ++d) xs_[d] = first_ ? xj[d] : xjb[d];                  \
                dd[u] = dist_fast<D>(xi_, xs_, SQ_);                                           \
                okm[u] = (hi_ & va_) | (lo_ & vb_);                                            \
                ya[u] = (ya[u] & hi_) | (__builtin_bitreverse64(yb[u]) & lo_);                 \
            }                                                                                  \
            _Pragma("unroll")                                                                  \
            for (int u = 0; u < U; ++u)                                                        \
                e[u] = (SQ_) ? tab_exp_clamped(-dd[u], sTab) : tab_exp(-dd[u], sTab);          \
            _Pragma("unroll")                                                                  \
            for (int u = 0; u < U; ++u) {                                                      \
                const unsigned long long live = ya[u] & okm[u];                                \
                sy += __popcll(live);                                                          \
                syd = fma(__builtin_amdgcn_inverse_ballot_w64(live) ? 1.0 : 0.0, dd[u], syd);  \
                if (__builtin_amdgcn_inverse_ballot_w64(okm[u])) {                             \
                    _Pragma("unroll")                                                          \
                    for (int k = 0; k < M; ++k) P[u][k] *= fma(E[k], e[u], 1.0);               \
                }                                                                              \
            }                                                                                  \
            if (++cnt >= nflush) {                                                             \
                _Pragma("unroll")                                                              \
                for (int k = 0; k < M; ++k) {                                                  \
                    _Pragma("unroll")                                                          \
                    for (int u = 0; u < U; ++u) { S[k] += fast_log(P[u][k]); P[u][k] = 1.0; }  \
                }                                                                              \
                cnt = 0;                                                                       \
            }                                                                                  \
        }                                                                                      \
    }
    if (diag && wave == 0) {
        if (c.squared) { DLSM_LLU_TRIPS_DIAG(1) } else { DLSM_LLU_TRIPS_DIAG(0) }
    }
    else if (c.squared) { DLSM_LLU_TRIPS(false, 1) }
    else if (whole) { DLSM_LLU_TRIPS(true, 0) }
    else { DLSM_LLU_TRIPS(false, 0) }
#undef DLSM_LLU_TRIPS
#undef DLSM_LLU_TRIPS_DIAG
#undef DLSM_LLU_PRIO
    // nflush == 64: all chains together hold <= 64 factors (1 + E) <= e^(600 / 64) each
    if (nflush >= LLU_ROWS) {
#pragma unroll
        for (int k = 0; k < M; ++k) {
            double q = P[0][k];
#pragma unroll
            for (int u = 1; u < U; ++u) q *= P[u][k];
            S[k] += fast_log(q);
        }
    } else {
#pragma unroll
        for (int k = 0; k < M; ++k)
#pragma unroll
            for (int u = 0; u < U; ++u) S[k] += fast_log(P[u][k]);
    }
    double acc[2 + M];
    acc[0] = (tid & 63) == 0 ? (double)sy : 0.0;
    acc[1] = syd;
#pragma unroll
    for (int k = 0; k < M; ++k) acc[2 + k] = S[k];
#pragma unroll
    for (int q = 0; q < 2 + M; ++q) {
        double v = wave_sum_all_tp(acc[q], tid & 63);
        if ((tid & 63) == 0) sRed[(tid >> 6) * (2 + M) + q] = v;
    }
    __syncthreads();
    if (tid < 2 + M)
        partials[(size_t)blockIdx.x * (2 + M) + tid] = sRed[tid] + sRed[(2 + M) + tid];
#ifdef DLSM_PIPE_TIMING
    {
        unsigned long long tl1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl1) : "v"(acc[1]));
        const unsigned w = blockIdx.x * 2 + (tid >> 6);
        if ((tid & 63) == 0 && w < 8192) { g_ll_t[w][0] = tl0; g_ll_t[w][1] = tl1; g_ll_t[w][2] = (unsigned long long)hwid | ((unsigned long long)xccid << 32); }
    }
#endif
}

template <int D, int M>
__global__ __launch_bounds__(LL_THREADS) void k_loglik_directed(
    ChainView c, LoglikCand cand, double *__restrict__ partials) {
    __shared__ double sXi[LL_TILE * D];
    __shared__ double sXj[LL_TILE * D];
    __shared__ double sRi[M][LL_TILE];      // 1 / radii of the row nodes
    __shared__ double sRj[M][LL_TILE];
    __shared__ uint32_t sY[LL_TILE * 4];    // Y[i, j]
    __shared__ uint32_t sYT[LL_TILE * 4];   // Y[j, i]
    __shared__ double sRed[4 * M];
    const int tid = threadIdx.x;
    const int nt = (c.N + LL_TILE - 1) / LL_TILE;
    const int ntri = nt * (nt + 1) / 2;
    const int t = blockIdx.x / ntri;
    int ti, tj;
    tile_decode(blockIdx.x % ntri, nt, ti, tj);
    const int i0 = ti * LL_TILE, j0 = tj * LL_TILE;
    const double *Xt = c.X + (size_t)t * c.N * D;
    for (int k = tid; k < LL_TILE * D; k += LL_THREADS) {
        int gi = i0 * D + k, gj = j0 * D + k;
        sXi[k] = gi < c.N * D ? Xt[gi] : 0.0;
        sXj[k] = gj < c.N * D ? Xt[gj] : 0.0;
    }
    for (int k = tid; k < LL_TILE; k += LL_THREADS) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            sRi[m][k] = (i0 + k) < c.N ? 1.0 / cand.radii[m][i0 + k] : 1.0;
            sRj[m][k] = (j0 + k) < c.N ? 1.0 / cand.radii[m][j0 + k] : 1.0;
        }
    }
    for (int k = tid; k < LL_TILE * 4; k += LL_THREADS) {
        int r = k >> 2, w = k & 3, gi = i0 + r;
        size_t off = ((size_t)t * c.N + gi) * c.W + (j0 >> 5) + w;
        sY[k] = gi < c.N ? c.ybits[off] : 0u;
        sYT[k] = gi < c.N ? c.ytbits[off] : 0u;
    }
    double bin[M], bout[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        bin[m] = cand.intercepts[2 * m];
        bout[m] = cand.intercepts[2 * m + 1];
    }
    __syncthreads();

    const int cj = tid & (LL_TILE - 1);
    const int half = tid >> 7;
    const int j = j0 + cj;
    double xj[D];
#pragma unroll
    for (int d = 0; d < D; ++d) xj[d] = sXj[cj * D + d];
    // eta_ij = b_in (1 - d / r_j) + b_out (1 - d / r_i) = B - d a,  a = b_in / r_j + b_out / r_i
    // eta_ji =                                            B - d g,  g = b_in / r_i + b_out / r_j
    // sum log((1 + e^eta_ij)(1 + e^eta_ji)) = log of a running product of (1 + E e^{-d a}) factors,
    // E = e^B, flushed through one log before it could leave the double range; an exponent
    // above 40 (a negative a at a large distance: the degenerate corner of the parameter space)
    // takes the term-by-term form instead.
    double L[M], P[M], Es[M], Bs[M], binrj[M], boutrj[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        L[m] = 0.0; P[m] = 1.0;
        Bs[m] = bin[m] + bout[m];
        Es[m] = exp(Bs[m]);
        binrj[m] = bin[m] * sRj[m][cj];
        boutrj[m] = bout[m] * sRj[m][cj];
    }
    if (j < c.N) {
        const int rbeg = half * 64;
        const int rend = min(rbeg + 64, j - i0);
        for (int r = rbeg; r < rend; ++r) {
            const double dd = dist_fast<D>(&sXi[r * D], xj, c.squared);
            const double yij = (double)((sY[r * 4 + (cj >> 5)] >> (cj & 31)) & 1);
            const double yji = (double)((sYT[r * 4 + (cj >> 5)] >> (cj & 31)) & 1);
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const double ri = sRi[m][r];
                // i -> j : directed_likelihoods_fast.pyx:199-203
                const double a = fma(bout[m], ri, binrj[m]), g = fma(bin[m], ri, boutrj[m]);
                const double xa = -dd * a, xg = -dd * g;
                L[m] += yij * (Bs[m] + xa) + yji * (Bs[m] + xg);
                if (fmax(Bs[m] + xa, Bs[m] + xg) > 40.0 || !(Es[m] < 1e17)) {
                    L[m] -= log((1.0 + exp(Bs[m] + xa)) * (1.0 + exp(Bs[m] + xg)));
                } else {
                    if (P[m] > 1e200) { L[m] -= log(P[m]); P[m] = 1.0; }
                    P[m] *= fma(Es[m], fast_exp(xa), 1.0) * fma(Es[m], fast_exp(xg), 1.0);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < M; ++m) L[m] -= log(P[m]);
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {
        double v = wave_sum_all(L[m]);
        if ((tid & 63) == 0) sRed[(tid >> 6) * M + m] = v;
    }
    __syncthreads();
    if (tid < M) {
        double s = 0.0;
        for (int w = 0; w < 4; ++w) s += sRed[w * M + tid];
        partials[(size_t)blockIdx.x * M + tid] = s;
    }
}

// ---------------------------------------------------------------------------
// Case-control full log-likelihood (a6, directed_likelihoods_fast.pyx:208-270): k_loglik_casecontrol_rows
// below.  LLCC_NODES nodes per workgroup and one record [L_0 .. L_{M-1}] per workgroup: fewer records for
// the single-workgroup reduction that follows.  (Rounds 1-4 had a wave-per-node form with the loads where
// they were used and a prefetch form with three fixed 64-lane slots per node; the rows form replaced both.)
// ---------------------------------------------------------------------------
constexpr int LLCC_NODES = 16;

// Packed records for the gathers of the case-control log-likelihood: [T][N][RW] doubles
// (x[D], r, r') with r / r' the radii of the two candidates.  A term gathers the position and
// the radius of a random node: from separate arrays those are two cache-line requests for 16 +
// 8 useful bytes, and the kernel is bound by exactly that request rate (6 M terms per pass at
// C4); from one 32-byte record (64 at d > 2) it is one.
// (d = 7, 8: 96 bytes)
__host__ __device__ constexpr int llcc_record_width(int D) { return D + 2 <= 4 ? 4 : (D + 2 <= 8 ? 8 : 12); }
template <int D>
__global__ __launch_bounds__(256) void k_pack_xr(const double *__restrict__ X,
                                                 const double *__restrict__ r0,
                                                 const double *__restrict__ r1, long nodes, int N,
                                                 double *__restrict__ XR) {
    constexpr int RW = llcc_record_width(D);
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nodes) return;
    const int i = (int)(q % N);
    double rec[RW];
#pragma unroll
    for (int d = 0; d < RW; ++d) rec[d] = 0.0;
#pragma unroll
    for (int d = 0; d < D; ++d) rec[d] = X[q * D + d];
    // RECIPROCAL radii (round 5): every gathered term needs b / r, and a reciprocal per term (v_rcp_f64 is a
    // quarter-rate instruction, + two Newton steps) was a tenth of the pass's arithmetic
    rec[D] = 1.0 / r0[i];
    rec[D + 1] = 1.0 / r1[i];
#pragma unroll
    for (int d = 0; d < RW; d += 2)
        *(double2 *)(XR + q * RW + d) = make_double2(rec[d], rec[d + 1]);
}

// The pass from the nodes' term ROWS (cc_rows.hpp; round 5).  Its predecessor gave a node three fixed
// 64-lane slots - its out-edges (20 of 64 lanes busy at config 4), controls 0-63, controls 64-127 (36 of
// 64) - so 38 % of the lanes it computed and gathered for were idle.  A row lists the node's out-edges
// and out-controls back to back at a known offset: ~120 terms fill TWO dense trips, their indices leave with
// the row's header (no degree round trip), and the control weight (N - deg - 1) / n_controls comes with the
// header instead of a float64 division per node.  Any out-degree and any number of controls: terms beyond
// the first 128 of a node take further trips.  A lane tells edges from controls by its position in the row;
// two nodes per wavefront and eight wavefronts per workgroup (94 / 114 registers: five / four wavefronts per
// SIMD - with four nodes per wavefront, 150 / 182 registers, the pass was slower than the slots it replaced).
// rslot: which of the record's two (reciprocal) radii a single candidate (M == 1) reads.
// 47.7 -> 41.9 us per pass at config 4 (average of the iteration's three passes, round 5).
constexpr int LLCR_THREADS = 512;        // 8 wavefronts x 2 nodes: the same LLCC_NODES nodes (and one record) per workgroup
template <int D, int M>
__global__ __launch_bounds__(LLCR_THREADS) void k_loglik_casecontrol_rows(
    ChainView c, LoglikCand cand, const double *__restrict__ XR, const int32_t *__restrict__ terms, int tw,
    double *__restrict__ partials, int rslot) {
    constexpr int NWV = LLCR_THREADS / 64, NPW = LLCC_NODES / NWV;   // two nodes per wavefront: ~90 registers,
    constexpr int RW = llcc_record_width(D);                         // five wavefronts per SIMD (four nodes: 150 - 180, two or three)
    constexpr int NS = 2;                      // 64-term trips requested up front per node
    __shared__ double sRed[NWV * M];
    __shared__ __attribute__((aligned(16))) double sTab[EXPTAB_N];       // tab_exp (device_common.hpp)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    exp_table_fill(sTab, tid);                 // the first 256 threads: one entry each; barrier below
    const long nodes = (long)c.T * c.N;
    // (M = 2: the second candidate may differ in its radii; M = 4 - the case-control loop's two intercept
    // steps at once - differs in intercepts only)
    const bool two_radii = M == 2 && cand.radii[1] != cand.radii[0];
    // (round 6: `node` is a ROW of the slice - rows are stored by descending term count inside every 512 nodes,
    // cc_rows.hpp - and a wavefront takes the rows `pi` from the head and `pi` from the tail of its batch: a long
    // node beside a short one, every wavefront about the same number of trips; `who` is the node a row belongs to.
    // Workgroup -> (slice, batch, 8 such pairs): T ceil(N / 16) workgroups, as ll_blocks() counts them)
    long node[NPW], who[NPW];
    static_assert(NPW == 2 && CC_SORT_B % LLCC_NODES == 0, "pairs of rows, whole workgroups per batch");
    // (grid (ceil(N / 16), T): a quotient by a launch constant in the prologue of every wavefront was part of what
    // the first sorted-row form lost - 73 against 69 us for the four-candidate pass on the degree-regular network)
    const int wps = (int)gridDim.x;
    const int sl = (int)blockIdx.y, ws = (int)blockIdx.x;
    const int lin = sl * wps + ws;                      // the workgroup's record
    const int bj0 = (ws / (CC_SORT_B / LLCC_NODES)) * CC_SORT_B, bnb = min(CC_SORT_B, c.N - bj0);
    const int pi = (ws % (CC_SORT_B / LLCC_NODES)) * NWV + wave;
    int outdeg[NPW], nt[NPW], e[NPW][NS];
    double adj[NPW];
    const char *rows[NPW];
    constexpr uint32_t HB = CP_HDR * sizeof(int32_t);
#pragma unroll
    for (int r = 0; r < NPW; ++r) {
        const bool have = r == 0 ? pi < (bnb + 1) / 2 : pi < bnb / 2;
        node[r] = have ? (long)sl * c.N + bj0 + (r == 0 ? pi : bnb - 1 - pi) : nodes;
        const long nn = node[r] < nodes ? node[r] : 0;
        rows[r] = (const char *)(terms + nn * tw);
        const int4 hdr = *(const int4 *)rows[r];
        who[r] = nn / c.N * c.N + *(const int32_t *)(rows[r] + 32);
        adj[r] = *(const double *)(rows[r] + 24);                       // adj_out
        outdeg[r] = node[r] < nodes ? hdr.y : 0;
        nt[r] = node[r] < nodes ? hdr.y + hdr.w : 0;                    // out-edges + out-controls
#pragma unroll
        for (int s = 0; s < NS; ++s)
            e[r][s] = *(const int32_t *)(rows[r] + (HB + 4u * (uint32_t)min(64 * s + lane, tw - CP_HDR - 1)));
    }
#pragma unroll
    for (int r = 0; r < NPW; ++r)
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (64 * s + lane >= nt[r]) e[r][s] = -1;
    __syncthreads();                           // sTab visible
    double xi[NPW][D], xe[NPW][NS][D], re0[NPW][NS], re1[NPW][NS], ri0[NPW], ri1[NPW];
#pragma unroll
    for (int r = 0; r < NPW; ++r) {
        const long nn = node[r] < nodes ? who[r] : 0;
        const int t = (int)(nn / c.N);
        const char *Rt = (const char *)(XR + (size_t)t * c.N * RW);
        {
            const double *rec = XR + (size_t)nn * RW;
#pragma unroll
            for (int d = 0; d < D; ++d) xi[r][d] = rec[d];
            ri0[r] = rec[D + (M == 1 ? rslot : 0)];
            ri1[r] = rec[D + 1];
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const double *rec = (const double *)(Rt + __umul24((uint32_t)max(e[r][s], 0), (uint32_t)(RW * sizeof(double))));
#pragma unroll
            for (int d = 0; d < D; ++d) xe[r][s][d] = rec[d];
            re0[r][s] = rec[D + (M == 1 ? rslot : 0)];
            re1[r][s] = rec[D + 1];
        }
    }
    double bin[M], bout[M];
#pragma unroll
    for (int m = 0; m < M; ++m) { bin[m] = cand.intercepts[2 * m]; bout[m] = cand.intercepts[2 * m + 1]; }
    double L[M], Pe[M];
#pragma unroll
    for (int m = 0; m < M; ++m) { L[m] = 0.0; Pe[m] = 1.0; }
    // one term: the partner's position xq and radii (rq0 / rq1) against node r's; edge: an out-edge
    // (directed_likelihoods_fast.pyx:236-247), else an out-control (:250-268)
#define DLSM_CCR_TERM(XQ_, RQ0_, RQ1_, EDGE_)                                                          \
    {                                                                                                  \
        const double dd_ = dist_fast<D>(XQ_, xi[r], c.squared);                                        \
        const double ire0_ = (RQ0_);               /* the records hold reciprocal radii */            \
        const double ire1_ = two_radii ? (RQ1_) : ire0_;                                               \
        _Pragma("unroll")                                                                              \
        for (int m = 0; m < M; ++m) {                                                                  \
            const double ire_ = m == 0 ? ire0_ : ire1_;                                                \
            const double eta_ = bin[m] * (1.0 - dd_ * ire_) + bout[m] * (1.0 - dd_ * iri[m]);          \
            /* log(1 + e^eta) = eta beyond 130: such a term adds nothing to an edge's sum and eta to a \
               control's (selects, not branches) */                                                    \
            const bool big_ = eta_ > 130.0;                                                            \
            const double f_ = big_ ? 1.0 : 1.0 + tab_exp(fmin(fmax(eta_, -700.0), 130.0), sTab);      \
            L[m] += ((EDGE_) && !big_) ? eta_ : 0.0;                                                   \
            ctl[m] += (!(EDGE_) && big_) ? eta_ : 0.0;                                                 \
            if (__builtin_amdgcn_ballot_w64(Pe[m] > 1e250))                                            \
                if (Pe[m] > 1e250) { L[m] -= fast_log(Pe[m]); Pe[m] = 1.0; }                           \
            if (__builtin_amdgcn_ballot_w64(Pc[m] > 1e250))                                            \
                if (Pc[m] > 1e250) { ctl[m] += fast_log(Pc[m]); Pc[m] = 1.0; }                         \
            Pe[m] *= (EDGE_) ? f_ : 1.0;                                                               \
            Pc[m] *= (EDGE_) ? 1.0 : f_;                                                               \
        }                                                                                              \
    }
#pragma unroll
    for (int r = 0; r < NPW; ++r) {
        if (node[r] >= nodes) continue;                   // wave-uniform
        double iri[M], Pc[M], ctl[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            iri[m] = m == 0 || !two_radii ? ri0[r] : ri1[r];   // (reciprocals; same radii: slot 1 may hold a proposal)
            Pc[m] = 1.0; ctl[m] = 0.0;
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (64 * s >= nt[r]) continue;                // wave-uniform
            if (e[r][s] >= 0) DLSM_CCR_TERM(xe[r][s], re0[r][s], re1[r][s], 64 * s + lane < outdeg[r])
        }
        // a node with more than 64 NS out-edges and out-controls: its further terms trip by trip
        const long nn = node[r];
        const char *Rt = (const char *)(XR + (size_t)(nn / c.N) * c.N * RW);
        for (int q0 = 64 * NS; q0 < nt[r]; q0 += 64) {
            const int q = q0 + lane;
            const int eq = q < nt[r] ? *(const int32_t *)(rows[r] + (HB + 4u * (uint32_t)q)) : -1;
            if (eq >= 0) {
                const double *rec = (const double *)(Rt + __umul24((uint32_t)eq, (uint32_t)(RW * sizeof(double))));
                double xq[D];
#pragma unroll
                for (int d = 0; d < D; ++d) xq[d] = rec[d];
                const double rq0 = rec[D + (M == 1 ? rslot : 0)], rq1 = rec[D + 1];
                DLSM_CCR_TERM(xq, rq0, rq1, q < outdeg[r])
            }
        }
#pragma unroll
        for (int m = 0; m < M; ++m) L[m] -= adj[r] * (ctl[m] + fast_log(Pc[m]));
    }
#undef DLSM_CCR_TERM
#pragma unroll
    for (int m = 0; m < M; ++m) {
        L[m] -= fast_log(Pe[m]);
        double v = wave_sum_all(L[m]);
        if (lane == 0) sRed[wave * M + m] = v;
    }
    __syncthreads();
    if (tid < M) {
        double s = 0.0;
        for (int w = 0; w < NWV; ++w) s += sRed[w * M + tid];
        partials[(size_t)lin * M + tid] = s;
    }
}

// Deterministic final reduction of `nrec` records of `width` doubles: one
// workgroup, fixed strided order + fixed tree.  out[q] = sum_r rec[r][q].
// NF: records a thread has in flight at a time (12: its whole share in one round trip at configs 2 to 4; 4 where
// the registers matter more - stage 1 of the HDP-LPCM loop, whose other roles run beside the likelihood pass)
template <int NF = 12>
__device__ __forceinline__ void reduce_records(const double *__restrict__ rec,
                                               int nrec, int width, double *sums,
                                               double *scratch /*4 * 256*/, int tid) {
    // up to 4 columns at a time share one tree (and its barriers); in a wider workgroup the
    // threads past 255 only keep the barriers company (same order of the sums for any width)
    for (int q0 = 0; q0 < width; q0 += 4) {
        const int nq = min(4, width - q0);
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        if (tid < 256) {
            // (twelve, then four records per thread requested together - the thread's whole share in one
            // round trip at configs 2 to 4; they are added in the same order)
            int r = tid;
            for (; NF > 4 && r + (NF - 1) * 256 < nrec; r += NF * 256) {
                double v[NF][4];
#pragma unroll
                for (int u = 0; u < NF; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        v[u][q] = q < nq ? rec[(size_t)(r + u * 256) * width + q0 + q] : 0.0;
#pragma unroll
                for (int u = 0; u < NF; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (q < nq) s[q] += v[u][q];
            }
            for (; r + 3 * 256 < nrec; r += 4 * 256) {
                double v[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        v[u][q] = q < nq ? rec[(size_t)(r + u * 256) * width + q0 + q] : 0.0;
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (q < nq) s[q] += v[u][q];
            }
            for (; r < nrec; r += 256)
                for (int q = 0; q < nq; ++q) s[q] += rec[(size_t)r * width + q0 + q];
            for (int q = 0; q < 4; ++q) scratch[q * 256 + tid] = s[q];
        }
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off)
                for (int q = 0; q < 4; ++q) scratch[q * 256 + tid] += scratch[q * 256 + tid + off];
            __syncthreads();
        }
        if (tid < nq) sums[q0 + tid] = scratch[tid * 256];
        __syncthreads();
    }
}

// finish dlsm_loglik_full: turn the summed records into m log-likelihoods
__global__ __launch_bounds__(256) void k_reduce_loglik(
    const double *__restrict__ partials, int nrec, int model, int M,
    const double *__restrict__ intercepts, double *__restrict__ out) {
    __shared__ double scratch[4 * 256];
    __shared__ double sums[8];
    const int width = model == DLSM_UNDIRECTED ? 2 + M : M;
    reduce_records(partials, nrec, width, sums, scratch, threadIdx.x);
    if (threadIdx.x < M) {
        const int k = threadIdx.x;
        out[k] = model == DLSM_UNDIRECTED
                     ? intercepts[k] * sums[0] - sums[1] - sums[2 + k]
                     : sums[k];
    }
}

// ---------------------------------------------------------------------------
// Prior terms of the sweep's logp closure for node (t, j) at position x.
//   random walk : sample_latent_positions.py:132-140
//   AR mixture  : sample_latent_positions.py:187-199
// ---------------------------------------------------------------------------
// COH: the neighbouring slices' positions were written by other workgroups of the SAME launch
// (persistent sweep): loaded past the L1 (device_common.hpp)
template <int D, bool COH = false>
__device__ __forceinline__ double node_log_prior(const ChainView &c, int t, int j,
                                                 const double *x) {
    const int N = c.N;
    double lp = 0.0;
    if (c.prior_kind == DLSM_PRIOR_RANDOM_WALK) {
        double s = 0.0;
        if (t == 0) {
#pragma unroll
            for (int d = 0; d < D; ++d) s += x[d] * x[d];
            lp -= 0.5 * s / c.tau_sq;
        } else {
            const double *xp = c.X + ((size_t)(t - 1) * N + j) * D;
#pragma unroll
            for (int d = 0; d < D; ++d) { const double df = x[d] - coh_load<COH>(xp + d); s += df * df; }
            lp -= 0.5 * s / c.sigma_sq;
        }
        if (t < c.T - 1) {
            const double *xn = c.X + ((size_t)(t + 1) * N + j) * D;
            s = 0.0;
#pragma unroll
            for (int d = 0; d < D; ++d) { const double df = coh_load<COH>(xn + d) - x[d]; s += df * df; }
            lp -= 0.5 * s / c.sigma_sq;
        }
    } else {
        const double lm = c.lmbda_p[0];
        const int zt = c.z[(size_t)t * N + j];
        const double *m = c.mu + (size_t)zt * D;
        double s = 0.0;
        if (t == 0) {
#pragma unroll
            for (int d = 0; d < D; ++d) s += (x[d] - m[d]) * (x[d] - m[d]);
        } else {
            const double *xp = c.X + ((size_t)(t - 1) * N + j) * D;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                double df = x[d] - (1 - lm) * coh_load<COH>(xp + d) - lm * m[d];
                s += df * df;
            }
        }
        lp -= 0.5 * s / c.sigma[zt];
        if (t < c.T - 1) {
            const int zn = c.z[(size_t)(t + 1) * N + j];
            const double *mn = c.mu + (size_t)zn * D;
            const double *xn = c.X + ((size_t)(t + 1) * N + j) * D;
            s = 0.0;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                double df = coh_load<COH>(xn + d) - (1 - lm) * x[d] - lm * mn[d];
                s += df * df;
            }
            lp -= 0.5 * s / c.sigma[zn];
        }
    }
    return lp;
}

// node_log_prior with its operands requested ahead of their use (the launch-per-batch resolvers: the
// terms' loads were a round trip of their own - two with the mixture prior, label then component -
// behind everything else the resolver had asked for).  request1: the neighbouring slices' rows, the
// labels, the blending coefficient; request2, once the labels are in: the components' means and
// variances; value(x): node_log_prior's arithmetic, term for term, on the operands at hand.
template <int D>
struct NodePriorPre {
    double xp[D], xn[D], m[D], mn[D], st, sn, lm;
    int zt, zn;
    bool has_p, has_n, mixture;
    double tau_sq, sigma_sq;
    __device__ __forceinline__ void request1(const ChainView &c, int t, int j) {
        const int N = c.N;
        mixture = c.prior_kind != DLSM_PRIOR_RANDOM_WALK;
        has_p = t > 0; has_n = t < c.T - 1;
        tau_sq = c.tau_sq; sigma_sq = c.sigma_sq;
        const double *rp = c.X + ((size_t)(has_p ? t - 1 : t) * N + j) * D;
        const double *rn = c.X + ((size_t)(has_n ? t + 1 : t) * N + j) * D;
#pragma unroll
        for (int d = 0; d < D; ++d) { xp[d] = rp[d]; xn[d] = rn[d]; }
        lm = 0.0; zt = 0; zn = 0;
        if (mixture) {
            lm = c.lmbda_p[0];
            zt = c.z[(size_t)t * N + j];
            zn = c.z[(size_t)(has_n ? t + 1 : t) * N + j];
        }
    }
    __device__ __forceinline__ void request2(const ChainView &c) {
        st = 1.0; sn = 1.0;
#pragma unroll
        for (int d = 0; d < D; ++d) { m[d] = 0.0; mn[d] = 0.0; }
        if (mixture) {
#pragma unroll
            for (int d = 0; d < D; ++d) { m[d] = c.mu[(size_t)zt * D + d]; mn[d] = c.mu[(size_t)zn * D + d]; }
            st = c.sigma[zt]; sn = c.sigma[zn];
        }
    }
    __device__ __forceinline__ double value(const double *x) const {
        double lp = 0.0;
        if (!mixture) {
            double s = 0.0;
            if (!has_p) {
#pragma unroll
                for (int d = 0; d < D; ++d) s += x[d] * x[d];
                lp -= 0.5 * s / tau_sq;
            } else {
#pragma unroll
                for (int d = 0; d < D; ++d) { const double df = x[d] - xp[d]; s += df * df; }
                lp -= 0.5 * s / sigma_sq;
            }
            if (has_n) {
                s = 0.0;
#pragma unroll
                for (int d = 0; d < D; ++d) { const double df = xn[d] - x[d]; s += df * df; }
                lp -= 0.5 * s / sigma_sq;
            }
        } else {
            double s = 0.0;
            if (!has_p) {
#pragma unroll
                for (int d = 0; d < D; ++d) s += (x[d] - m[d]) * (x[d] - m[d]);
            } else {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    double df = x[d] - (1 - lm) * xp[d] - lm * m[d];
                    s += df * df;
                }
            }
            lp -= 0.5 * s / st;
            if (has_n) {
                s = 0.0;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    double df = xn[d] - (1 - lm) * x[d] - lm * mn[d];
                    s += df * df;
                }
                lp -= 0.5 * s / sn;
            }
        }
        return lp;
    }
};

// ---------------------------------------------------------------------------
// Per-node partial log-likelihood (a1/a2/a3), the function seam used by the
// parity tests.  Grid (N, T), 256 threads; one node per workgroup.  Written in
// the reference's literal form  y*eta - log(1 + exp(eta)).
// ovr_* : replace X[ovr_t, ovr_j] by ovr_x for that one node (x argument of the
// closure); ovr_t < 0 : none.
// ---------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_partial_all(ChainView c, int with_prior,
                                                     int only_t, int only_j,
                                                     const double *__restrict__ ovr_x,
                                                     double *__restrict__ out) {
    __shared__ double sRed[4];
    const int tid = threadIdx.x;
    const int t = only_t >= 0 ? only_t : blockIdx.y;
    const int j = only_j >= 0 ? only_j : blockIdx.x;
    const int N = c.N;
    const double *Xt = c.X + (size_t)t * N * D;
    double x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = ovr_x ? ovr_x[d] : Xt[(size_t)j * D + d];
    double acc = 0.0;
    if (c.model == DLSM_UNDIRECTED) {
        const double b = c.intercept[0];
        const uint32_t *row = c.ybits + ((size_t)t * N + j) * c.W;
        for (int i = tid; i < N; i += 256) {
            if (i == j) continue;
            double eta = b - dist_of<D>(&Xt[(size_t)i * D], x, c.squared);
            acc += (double)bit_of(row, i) * eta - log(1.0 + exp(eta));
        }
    } else if (c.model == DLSM_DIRECTED) {
        const double bin = c.intercept[0], bout = c.intercept[1];
        const double rj = c.radii[j];
        const uint32_t *row = c.ybits + ((size_t)t * N + j) * c.W;
        const uint32_t *col = c.ytbits + ((size_t)t * N + j) * c.W;
        for (int i = tid; i < N; i += 256) {
            if (i == j) continue;
            const double dd = dist_of<D>(&Xt[(size_t)i * D], x, c.squared);
            const double ri = c.radii[i];
            double eta = bin * (1 - dd / ri) + bout * (1 - dd / rj);
            acc += (double)bit_of(row, i) * eta - log(1.0 + exp(eta));
            eta = bin * (1 - dd / rj) + bout * (1 - dd / ri);
            acc += (double)bit_of(col, i) * eta - log(1.0 + exp(eta));
        }
    } else {
        const double bin = c.intercept[0], bout = c.intercept[1];
        const double rj = c.radii[j];
        const size_t node = (size_t)t * N + j;
        const int in_deg = c.degree[node * 2], out_deg = c.degree[node * 2 + 1];
        const int32_t *ie = c.in_edges + node * c.Din;
        const int32_t *oe = c.out_edges + node * c.Dout;
        const int32_t *ci = c.ctrl_in + node * c.C;
        const int32_t *co = c.ctrl_out + node * c.C;
        for (int k = tid; k < in_deg; k += 256) {
            const int e = ie[k];
            const double dd = dist_of<D>(e == j ? x : &Xt[(size_t)e * D], x, c.squared);
            double eta = bin * (1 - dd / rj) + bout * (1 - dd / c.radii[e]);
            acc += eta - log(1.0 + exp(eta));
        }
        for (int k = tid; k < out_deg; k += 256) {
            const int e = oe[k];
            const double dd = dist_of<D>(e == j ? x : &Xt[(size_t)e * D], x, c.squared);
            double eta = bin * (1 - dd / c.radii[e]) + bout * (1 - dd / rj);
            acc += eta - log(1.0 + exp(eta));
        }
        // number of valid controls (lists are -1 terminated)
        int nci = 0, nco = 0;
        for (int k = 0; k < c.C && ci[k] >= 0; ++k) ++nci;
        for (int k = 0; k < c.C && co[k] >= 0; ++k) ++nco;
        double ctl = 0.0;
        for (int k = tid; k < nci; k += 256) {
            const int e = ci[k];
            const double dd = dist_of<D>(e == j ? x : &Xt[(size_t)e * D], x, c.squared);
            double eta = bin * (1 - dd / rj) + bout * (1 - dd / c.radii[e]);
            ctl += log(1.0 + exp(eta));
        }
        acc -= ((double)(N - in_deg - 1) / (double)nci) * ctl;
        ctl = 0.0;
        for (int k = tid; k < nco; k += 256) {
            const int e = co[k];
            const double dd = dist_of<D>(e == j ? x : &Xt[(size_t)e * D], x, c.squared);
            double eta = bin * (1 - dd / c.radii[e]) + bout * (1 - dd / rj);
            ctl += log(1.0 + exp(eta));
        }
        acc -= ((double)(N - out_deg - 1) / (double)nco) * ctl;
    }
    double tot = block_sum_all<4>(acc, sRed, tid);
    if (tid == 0) {
        if (with_prior) tot += node_log_prior<D>(c, t, j, x);
        out[only_t >= 0 ? 0 : (size_t)t * N + j] = tot;
    }
}

}  // namespace dlsm
