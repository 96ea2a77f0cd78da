// Starting values of a chain, on the device: shortest-path hop
// matrices from the bit-packed network, SMACOF metric MDS for the first slice,
// the Lanczos pieces of the Sarkar-Moore eigen step for the following ones, and
// the sums behind the conditional-MLE objectives.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "chain.hpp"
#include "device_common.hpp"
#include "kernels_sweep.hpp"

namespace dlsm {

constexpr uint16_t HOP_UNREACHED = 0xFFFF;
constexpr int BFS_THREADS = 256;

// ---------------------------------------------------------------------------
// latent_space.py:36-38: csgraph.shortest_path(Y, directed=False,
// unweighted=True) = hop counts of the symmetrised graph.  One workgroup per
// source node runs a level-synchronous BFS on bitsets in LDS: expanding a
// frontier node ORs its W-word adjacency row into the next-level set, so every
// (source, reached node) pair costs one coalesced row read from L2.
// LDS: visited | frontier | next (W words each) + the hop row (N uint16).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BFS_THREADS) void k_hops_bfs(ChainView c,
                                                          uint16_t *__restrict__ hops,
                                                          int *__restrict__ slice_max) {
    extern __shared__ uint32_t bfs_lds[];
    const int N = c.N, W = c.W;
    uint32_t *V = bfs_lds, *F = bfs_lds + W, *Nx = bfs_lds + 2 * W;
    uint16_t *lvl = (uint16_t *)(bfs_lds + 3 * W);
    const int t = blockIdx.y, src = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t *A = c.ybits + (size_t)t * N * W;
    const uint32_t *At = c.ytbits ? c.ytbits + (size_t)t * N * W : nullptr;

    for (int w = tid; w < W; w += BFS_THREADS) {
        const uint32_t b = (w == (src >> 5)) ? (1u << (src & 31)) : 0u;
        V[w] = b; F[w] = b; Nx[w] = 0u;
    }
    for (int j = tid; j < N; j += BFS_THREADS) lvl[j] = j == src ? 0 : HOP_UNREACHED;
    __syncthreads();

    int level = 0, reached_level = 0;
    while (true) {
        ++level;
        for (int wbase = 0; wbase < W; wbase += 64) {
            const int wo = min(wbase + lane, W - 1);
            uint32_t acc = 0u;
            for (int fw = wave; fw < W; fw += BFS_THREADS / 64) {
                uint32_t bits = F[fw];                      // wave-uniform
                while (bits) {
                    const int j = fw * 32 + __ffs(bits) - 1;
                    bits &= bits - 1;
                    uint32_t a = A[(size_t)j * W + wo];
                    if (At) a |= At[(size_t)j * W + wo];
                    acc |= a;
                }
            }
            if (wbase + lane < W && acc) atomicOr(&Nx[wo], acc);
        }
        __syncthreads();
        int any = 0;
        for (int w = tid; w < W; w += BFS_THREADS) {
            const uint32_t nw = Nx[w] & ~V[w];
            F[w] = nw; V[w] |= nw; Nx[w] = 0u;
            any |= nw != 0u;
        }
        any = __syncthreads_or(any);
        if (!any) break;
        for (int j = tid; j < N; j += BFS_THREADS)
            if ((F[j >> 5] >> (j & 31)) & 1u) lvl[j] = (uint16_t)level;
        reached_level = level;
    }
    __syncthreads();
    uint16_t *row = hops + ((size_t)t * N + src) * N;
    for (int j = tid; j < N; j += BFS_THREADS) row[j] = lvl[j];
    if (tid == 0) atomicMax(&slice_max[t], reached_level);
}

// latent_space.py:41-42: unconnected pairs get the largest finite distance + 1
__global__ __launch_bounds__(256) void k_hops_fill(uint16_t *__restrict__ hops, int N,
                                                   const int *__restrict__ slice_max) {
    const int t = blockIdx.y;
    const size_t n2 = (size_t)N * N;
    const uint16_t fill = (uint16_t)(slice_max[t] + 1);
    uint16_t *p = hops + (size_t)t * n2;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n2; k += (size_t)gridDim.x * 256)
        if (p[k] == HOP_UNREACHED) p[k] = fill;
}

__global__ __launch_bounds__(256) void k_hops_to_double(const uint16_t *__restrict__ hops,
                                                        size_t n, double *__restrict__ out) {
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256)
        out[k] = (double)hops[k];
}

// ---------------------------------------------------------------------------
// SMACOF (sklearn.manifold._mds._smacof_single, metric=True; called from
// latent_space.py:66-68).  One pass reads configuration X_p and produces, in the
// same sweep over the N x N dissimilarities, the stress of X_p and the Guttman
// transform X_{p+1} = (1/N) B(X_p) X_p, using (B X)_i = sum_j ratio_ij (x_i - x_j).
// One wavefront per row i, lanes over j.
// ---------------------------------------------------------------------------
struct SmacofState {
    double old_stress, stress, sumsq;
    int32_t n_iter, done, answer, pad_;
};
constexpr int SM_THREADS = 256;
constexpr int SM_ROWS = SM_THREADS / 64;

template <int D>
__global__ __launch_bounds__(SM_THREADS) void k_smacof_pass(
    const uint16_t *__restrict__ hops, int N, const double *__restrict__ Xin,
    double *__restrict__ Xout, double *__restrict__ rec,
    const SmacofState *__restrict__ st) {
    const int run = blockIdx.y;
    if (st[run].done) return;
    __shared__ double sred[2][SM_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * SM_ROWS + wave;
    const double *X = Xin + (size_t)run * N * D;
    double stress = 0.0, sumsq = 0.0;
    if (i < N) {
        double xi[D], acc[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { xi[d] = X[(size_t)i * D + d]; acc[d] = 0.0; }
        const uint16_t *hrow = hops + (size_t)i * N;
        for (int j = lane; j < N; j += 64) {
            double df[D], s = 0.0;
#pragma unroll
            for (int d = 0; d < D; ++d) { df[d] = xi[d] - X[(size_t)j * D + d]; s += df[d] * df[d]; }
            const double dis = sqrt(s);
            const double delta = (double)hrow[j];
            const double e = dis - delta;
            stress += e * e;
            sumsq += s;
            const double ratio = delta / (dis == 0.0 ? 1e-5 : dis);
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] += ratio * df[d];
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const double a = wave_sum_all(acc[d]);
            if (lane == 0) Xout[((size_t)run * N + i) * D + d] = a / (double)N;
        }
    }
    stress = wave_sum_all(stress);
    sumsq = wave_sum_all(sumsq);
    if (lane == 0) { sred[0][wave] = stress; sred[1][wave] = sumsq; }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < SM_ROWS; ++w) { a += sred[0][w]; b += sred[1][w]; }
        rec[((size_t)run * gridDim.x + blockIdx.x) * 2 + 0] = a;
        rec[((size_t)run * gridDim.x + blockIdx.x) * 2 + 1] = b;
    }
}

// Convergence test of _smacof_single after pass p: stress(X_p) against
// stress(X_{p-1}); the answer of a finished run is X_p = buffer p % 2.
__global__ __launch_bounds__(256) void k_smacof_check(const double *__restrict__ rec,
                                                      int nblk, int p, int max_iter,
                                                      double eps, SmacofState *st) {
    const int run = blockIdx.x;
    if (st[run].done) return;
    __shared__ double buf[2][4];
    const int tid = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int k = tid; k < nblk; k += 256) {
        a += rec[((size_t)run * nblk + k) * 2 + 0];
        b += rec[((size_t)run * nblk + k) * 2 + 1];
    }
    a = block_sum_all<4>(a, buf[0], tid);
    b = block_sum_all<4>(b, buf[1], tid);
    if (tid == 0) {
        const double stress = 0.5 * a;
        SmacofState s = st[run];
        bool stop = false;
        if (p >= 2 && (s.old_stress - stress) / (0.5 * b) < eps) stop = true;
        if (p >= max_iter) stop = true;
        s.old_stress = stress;
        if (stop) { s.done = 1; s.n_iter = p; s.answer = p & 1; s.stress = stress; s.sumsq = b; }
        st[run] = s;
    }
}

// ---------------------------------------------------------------------------
// Sarkar-Moore step (latent_space.py:71-89): top-D eigenpairs of
//   G = alpha * H (-D_t^2 / 2) H + beta * X_{t-1} X_{t-1}^T,  H = I - 11^T / N,
// by Lanczos with full reorthogonalisation; G is never formed.
// small[]: [0] mean of the current Lanczos vector q, [1..D] X_{t-1}^T q.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(SM_THREADS) void k_gmds_matvec(
    const uint16_t *__restrict__ hops, int N, const double *__restrict__ q,
    const double *__restrict__ small, double *__restrict__ wB) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * SM_ROWS + wave;
    if (i >= N) return;
    const double qbar = small[0];
    const uint16_t *hrow = hops + (size_t)i * N;
    double acc = 0.0;
    for (int j = lane; j < N; j += 64) {
        const double h = (double)hrow[j];
        acc += (-0.5 * h * h) * (q[j] - qbar);
    }
    acc = wave_sum_all(acc);
    if (lane == 0) wB[i] = acc;
}

constexpr int LZ_THREADS = 1024;
constexpr int LZ_WAVES = LZ_THREADS / 64;

template <int D>
__device__ __forceinline__ void lanczos_publish(int N, const double *__restrict__ qn,
                                                const double *__restrict__ Xprev,
                                                double *small, double (*buf)[LZ_WAVES],
                                                int tid) {
    double m = 0.0, cx[D];
#pragma unroll
    for (int d = 0; d < D; ++d) cx[d] = 0.0;
    for (int i = tid; i < N; i += LZ_THREADS) {
        const double v = qn[i];
        m += v;
#pragma unroll
        for (int d = 0; d < D; ++d) cx[d] += Xprev[(size_t)i * D + d] * v;
    }
    m = block_sum_all<LZ_WAVES>(m, buf[0], tid);
#pragma unroll
    for (int d = 0; d < D; ++d) cx[d] = block_sum_all<LZ_WAVES>(cx[d], buf[1 + d], tid);
    if (tid == 0) {
        small[0] = m / (double)N;
#pragma unroll
        for (int d = 0; d < D; ++d) small[1 + d] = cx[d];
    }
}

// q_0: a fixed pseudo-random direction (the eigenpairs do not depend on it)
template <int D>
__global__ __launch_bounds__(LZ_THREADS) void k_lanczos_init(int N, int t, uint64_t seed,
                                                             const double *__restrict__ Xprev,
                                                             double *__restrict__ Q,
                                                             double *__restrict__ small) {
    __shared__ double buf[2 + D][LZ_WAVES];
    const int tid = threadIdx.x;
    double ss = 0.0;
    for (int i = tid; i < N; i += LZ_THREADS) {
        const U4 r = philox4x32_10(seed, (uint32_t)i, (uint32_t)t, 0u, 0xFFu);
        const double v = u53(r.x, r.y) - 0.5;
        Q[i] = v;
        ss += v * v;
    }
    ss = block_sum_all<LZ_WAVES>(ss, buf[1 + D], tid);
    const double inv = 1.0 / sqrt(ss);
    for (int i = tid; i < N; i += LZ_THREADS) Q[i] *= inv;
    __syncthreads();
    lanczos_publish<D>(N, Q, Xprev, small, buf, tid);
}

// One Lanczos step k: w = G q_k from the matvec's H-less B part, the three-term
// recurrence, two passes of classical Gram-Schmidt against q_0..q_k, then q_{k+1}.
// ab[k] = alpha_k, ab[kmax + k] = beta_k (norm of the residual).
template <int D>
__global__ __launch_bounds__(LZ_THREADS) void k_lanczos_step(
    int N, int k, int kmax, double alpha_w, double beta_w,
    const double *__restrict__ Xprev, double *__restrict__ Q,
    const double *__restrict__ wB, double *__restrict__ w, double *__restrict__ small,
    double *__restrict__ ab) {
    extern __shared__ double lz_coef[];                 // kmax + 1
    __shared__ double buf[2 + D][LZ_WAVES];
    __shared__ double buf2[2][LZ_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double *qk = Q + (size_t)k * N;
    double c[D];
#pragma unroll
    for (int d = 0; d < D; ++d) c[d] = small[1 + d];

    double m = 0.0;
    for (int i = tid; i < N; i += LZ_THREADS) m += wB[i];
    m = block_sum_all<LZ_WAVES>(m, buf2[0], tid) / (double)N;
    double a = 0.0;
    for (int i = tid; i < N; i += LZ_THREADS) {
        double lr = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) lr += Xprev[(size_t)i * D + d] * c[d];
        const double wi = alpha_w * (wB[i] - m) + beta_w * lr;
        w[i] = wi;
        a += wi * qk[i];
    }
    a = block_sum_all<LZ_WAVES>(a, buf2[1], tid);
    const double bprev = k > 0 ? ab[kmax + k - 1] : 0.0;
    const double *qp = k > 0 ? Q + (size_t)(k - 1) * N : qk;
    for (int i = tid; i < N; i += LZ_THREADS)
        w[i] -= a * qk[i] + (k > 0 ? bprev * qp[i] : 0.0);
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
        for (int mcol = wave; mcol <= k; mcol += LZ_WAVES) {
            const double *qm = Q + (size_t)mcol * N;
            double s = 0.0;
            for (int i = lane; i < N; i += 64) s += qm[i] * w[i];
            s = wave_sum_all(s);
            if (lane == 0) lz_coef[mcol] = s;
        }
        __syncthreads();
        for (int i = tid; i < N; i += LZ_THREADS) {
            double s = 0.0;
            for (int mcol = 0; mcol <= k; ++mcol) s += lz_coef[mcol] * Q[(size_t)mcol * N + i];
            w[i] -= s;
        }
        __syncthreads();
    }
    double ss = 0.0;
    for (int i = tid; i < N; i += LZ_THREADS) ss += w[i] * w[i];
    ss = block_sum_all<LZ_WAVES>(ss, buf[1 + D], tid);
    const double bk = sqrt(ss);
    if (tid == 0) { ab[k] = a; ab[kmax + k] = bk; }
    double *qn = Q + (size_t)(k + 1) * N;
    const double inv = bk > 0.0 ? 1.0 / bk : 0.0;
    for (int i = tid; i < N; i += LZ_THREADS) qn[i] = w[i] * inv;
    __syncthreads();
    lanczos_publish<D>(N, qn, Xprev, small, buf, tid);
}

// X_t = (Q S) sqrt(theta), then the Procrustes rotation onto X_{t-1}
// (latent_space.py:84-89; procrustes.py:20-25).  S: [D][k] Ritz coefficients.
template <int D>
__global__ __launch_bounds__(LZ_THREADS) void k_gmds_finish(
    int N, int k, const double *__restrict__ S, const double *__restrict__ theta,
    const double *__restrict__ Q, const double *__restrict__ Xprev,
    double *__restrict__ Xout) {
    __shared__ double buf[D * D][LZ_WAVES];
    __shared__ double sR[D * D];
    const int tid = threadIdx.x;
    double M[D][D];
    for (int a = 0; a < D; ++a)
        for (int b = 0; b < D; ++b) M[a][b] = 0.0;
    double sc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) sc[d] = sqrt(theta[d]);
    for (int i = tid; i < N; i += LZ_THREADS) {
        double x[D];
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = 0.0;
        for (int mcol = 0; mcol < k; ++mcol) {
            const double qv = Q[(size_t)mcol * N + i];
#pragma unroll
            for (int d = 0; d < D; ++d) x[d] += qv * S[(size_t)d * k + mcol];
        }
#pragma unroll
        for (int d = 0; d < D; ++d) { x[d] *= sc[d]; Xout[(size_t)i * D + d] = x[d]; }
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
            for (int b = 0; b < D; ++b) M[a][b] += x[a] * Xprev[(size_t)i * D + b];
    }
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b)
            M[a][b] = block_sum_all<LZ_WAVES>(M[a][b], buf[a * D + b], tid);
    if (tid == 0) {
        double R[D][D];
        jacobi_polar<D>(M, R);
        for (int a = 0; a < D; ++a)
            for (int b = 0; b < D; ++b) sR[a * D + b] = R[a][b];
    }
    __syncthreads();
    for (int i = tid; i < N; i += LZ_THREADS) {
        double x[D], y[D];
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = Xout[(size_t)i * D + d];
#pragma unroll
        for (int b = 0; b < D; ++b) {
            double s = 0.0;
#pragma unroll
            for (int a = 0; a < D; ++a) s += x[a] * sR[a * D + b];
            y[b] = s;
        }
#pragma unroll
        for (int d = 0; d < D; ++d) Xout[(size_t)i * D + d] = y[d];
    }
}

// ---------------------------------------------------------------------------
// Sums behind the conditional MLEs of lsm.py:32-97 at the chain's positions:
//   undirected (p0 = log scale, p1 = intercept), eta = p1 - exp(p0) d_ij:
//     rec = [ loglik over i<j (network_likelihoods.py:26-33),
//             scale_grad (lsm.py:39-44: over both orders of a dyad),
//             undirected_intercept_grad (lsm.py:32-36) ]
//   directed (p0 = b_in, p1 = b_out), eta = p0 (1 - d/r_j) + p1 (1 - d/r_i):
//     rec = [ loglik over i != j (directed_likelihoods_fast.pyx:185-205),
//             directed_intercept_grad in / out (:20-43) ]
// One wavefront per (t, i) row over all j != i.
// ---------------------------------------------------------------------------
template <int D, int MODEL>
__global__ __launch_bounds__(SM_THREADS) void k_mle_sums(ChainView c, double p0, double p1,
                                                         double *__restrict__ rec) {
    __shared__ double sred[3][SM_ROWS];
    const int N = c.N, W = c.W;
    const int t = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * SM_ROWS + wave;
    const double *X = c.X + (size_t)t * N * D;
    const double scale = MODEL == DLSM_UNDIRECTED ? exp(p0) : 0.0;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    if (i < N) {
        double xi[D];
#pragma unroll
        for (int d = 0; d < D; ++d) xi[d] = X[(size_t)i * D + d];
        const uint32_t *yrow = c.ybits + ((size_t)t * N + i) * W;
        const double ri = MODEL == DLSM_UNDIRECTED ? 1.0 : c.radii[i];
        // undirected: each dyad once (the upper triangle j > i, as the reference's
        // log-likelihood), weighted as its gradient sums over both orders; one exp
        // per dyad: e = exp(-|eta|) gives both softplus and expit
        const int jbeg = MODEL == DLSM_UNDIRECTED ? i + 1 : 0;
        for (int j = jbeg + lane; j < N; j += 64) {
            if (j == i) continue;
            double xj[D];
#pragma unroll
            for (int d = 0; d < D; ++d) xj[d] = X[(size_t)j * D + d];
            const double dist = dist_of<D>(xi, xj, c.squared);
            const double y = (double)bit_of(yrow, j);
            double eta, sd = 0.0, d_in = 0.0, d_out = 0.0;
            if (MODEL == DLSM_UNDIRECTED) {
                sd = scale * dist;
                eta = p1 - sd;
            } else {
                d_in = 1.0 - dist / c.radii[j];
                d_out = 1.0 - dist / ri;
                eta = p0 * d_in + p1 * d_out;
            }
            const double e = exp(-fabs(eta));
            const double sp = fmax(eta, 0.0) + log1p(e);
            const double step = y - (eta >= 0.0 ? 1.0 : e) / (1.0 + e);
            s0 += y * eta - sp;
            if (MODEL == DLSM_UNDIRECTED) {
                s1 += -2.0 * sd * step;
                s2 += step;
            } else {
                s1 += d_in * step;
                s2 += d_out * step;
            }
        }
    }
    s0 = wave_sum_all(s0); s1 = wave_sum_all(s1); s2 = wave_sum_all(s2);
    if (lane == 0) { sred[0][wave] = s0; sred[1][wave] = s1; sred[2][wave] = s2; }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0, d = 0.0;
        for (int w = 0; w < SM_ROWS; ++w) { a += sred[0][w]; b += sred[1][w]; d += sred[2][w]; }
        double *r = rec + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3;
        r[0] = a; r[1] = b; r[2] = d;
    }
}

// out[c] = sum over n records of C columns (fixed order: deterministic)
template <int C>
__global__ __launch_bounds__(1024) void k_sum_records(const double *__restrict__ rec, size_t n,
                                                      double *__restrict__ out) {
    __shared__ double buf[C][16];
    const int tid = threadIdx.x;
    double s[C];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) s[cc] = 0.0;
    for (size_t k = tid; k < n; k += 1024)
#pragma unroll
        for (int cc = 0; cc < C; ++cc) s[cc] += rec[k * C + cc];
#pragma unroll
    for (int cc = 0; cc < C; ++cc) {
        s[cc] = block_sum_all<16>(s[cc], buf[cc], tid);
        if (tid == 0) out[cc] = s[cc];
    }
}

// ---- longitudinal k-means (latent_space.py:98-137 -> sklearn.cluster.KMeans, Lloyd) ------------
// scikit-learn's lloyd_iter_chunked_dense on the centred N x F matrix of time-stacked
// trajectories (F = T d): the E-step assigns a sample to argmin_k |c_k|^2 - 2 x.c_k (first
// minimum), the M-step averages the members.  The seeding (k-means++) draws from the caller's
// RandomState and stays on the host, as the BFGS driver of the conditional MLEs does.
// labels[i], changed (a count) by the first kernel; centres, member counts, squared shift by the
// second (one workgroup per cluster, members summed in index order: reproducible).
__global__ __launch_bounds__(256) void k_kmeans_assign(const double *__restrict__ X, int N, int F, int K,
                                                       const double *__restrict__ centers,
                                                       int32_t *__restrict__ labels,
                                                       int32_t *__restrict__ changed) {
    extern __shared__ double km_sC[];              // K x F centres, K squared norms
    double *sN = km_sC + (size_t)K * F;
    for (int q = threadIdx.x; q < K * F; q += 256) km_sC[q] = centers[q];
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 256) {
        double n2 = 0.0;
        for (int f = 0; f < F; ++f) n2 = fma(km_sC[k * F + f], km_sC[k * F + f], n2);
        sN[k] = n2;
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const double *x = X + (size_t)i * F;
    int best = 0;
    double bestv = 0.0;
    for (int k = 0; k < K; ++k) {
        double dot = 0.0;
        for (int f = 0; f < F; ++f) dot = fma(x[f], km_sC[k * F + f], dot);
        const double v = fma(-2.0, dot, sN[k]);
        if (k == 0 || v < bestv) { bestv = v; best = k; }
    }
    if (labels[i] != best) atomicAdd(changed, 1);
    labels[i] = best;
}

__global__ __launch_bounds__(256) void k_kmeans_update(const double *__restrict__ X, int N, int F,
                                                       const int32_t *__restrict__ labels,
                                                       const double *__restrict__ centers_old,
                                                       double *__restrict__ centers_new,
                                                       int32_t *__restrict__ counts,
                                                       double *__restrict__ shift_sq) {
    __shared__ double sRed[4];
    __shared__ int sCnt[4];
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int cnt = 0;
    for (int i = tid; i < N; i += 256) cnt += labels[i] == k;
    {
        double c = wave_sum_all((double)cnt);
        if (lane == 0) sCnt[wave] = (int)c;
    }
    __syncthreads();
    const int total = (sCnt[0] + sCnt[1]) + (sCnt[2] + sCnt[3]);
    __syncthreads();
    double sh = 0.0;
    for (int f = 0; f < F; ++f) {
        double v = 0.0;
        for (int i = tid; i < N; i += 256) v += labels[i] == k ? X[(size_t)i * F + f] : 0.0;
        v = wave_sum_all(v);
        if (lane == 0) sRed[wave] = v;
        __syncthreads();
        const double sum = (sRed[0] + sRed[1]) + (sRed[2] + sRed[3]);
        __syncthreads();
        const double cn = total > 0 ? sum / (double)total : centers_old[k * F + f];
        if (tid == 0) centers_new[k * F + f] = cn;
        const double df = cn - centers_old[k * F + f];
        sh = fma(df, df, sh);
    }
    if (tid == 0) { counts[k] = total; shift_sq[k] = sh; }
}

}  // namespace dlsm
