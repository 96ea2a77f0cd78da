// Latent-position block update: random-walk
// Metropolis over every (t, node), Gauss-Seidel inside a time slice, slices of
// equal parity concurrently (they are conditionally independent given the
// other parity: sample_latent_positions.py:132-140 / :187-199 couple (t, j)
// only to (t-1, j) and (t+1, j)).
#pragma once
#include "chain.hpp"
#include "device_common.hpp"
#include "kernels_loglik.hpp"

namespace dlsm {

constexpr int SW_THREADS = 1024;
constexpr int SW_WAVES = SW_THREADS / 64;
constexpr int SW_CHUNK = 256;          // proposals generated per LDS refill

// proposal record in LDS: x0[D], x1[D], logu, prior_delta
template <int D> struct PropRec { static constexpr int W = 2 * D + 2; };

// Generate the proposal of step (t, j): metropolis.py:44 + :49 with the
// engine's Philox draws; x0 = current position.
template <int D>
__device__ __forceinline__ void make_proposal(const ChainView &c, uint32_t iter,
                                              int t, int j, const double *x0,
                                              double step, double *x1,
                                              double &logu) {
#pragma unroll
    for (int d = 0; d < D; d += 2) {
        double u0, u1, z0, z1;
        philox_uniform2(c.seed, (uint32_t)j, (uint32_t)t | ((uint32_t)(d / 2) << 16),
                        iter, stream_word(c.chain, STREAM_SWEEP_NORMAL), u0, u1);
        box_muller(u0, u1, z0, z1);
        x1[d] = x0[d] + step * z0;
        if (d + 1 < D) x1[d + 1] = x0[d + 1] + step * z1;
    }
    double u0, u1;
    philox_uniform2(c.seed, (uint32_t)j, (uint32_t)t, iter,
                    stream_word(c.chain, STREAM_SWEEP_UNIFORM), u0, u1);
    logu = log(u0);
}

// Change of node j's network log-likelihood contribution from neighbour i when
// j moves x0 -> x1.
//   undirected (static_network_fast.pyx:17-44):
//     y (d0 - d1) + log((1 + E e^{-d0}) / (1 + E e^{-d1})),  E = e^{b}
__device__ __forceinline__ double delta_undirected(double d0, double d1, int y,
                                                   double E) {
    const double p0 = 1.0 + E * exp(-d0);
    const double p1 = 1.0 + E * exp(-d1);
    double v = log(p0 / p1);
    if (y) v += d0 - d1;
    return v;
}
//   directed (directed_likelihoods_fast.pyx:46-80), both directions:
//     eta_out = s - d a,  eta_in = s - d c,  s = b_in + b_out,
//     a = b_in / r_i + b_out / r_j,  c = b_in / r_j + b_out / r_i
__device__ __forceinline__ double delta_directed(double d0, double d1, int y_ji,
                                                 int y_ij, double a, double cc,
                                                 double Es) {
    const double num = (1.0 + Es * exp(-d0 * a)) * (1.0 + Es * exp(-d0 * cc));
    const double den = (1.0 + Es * exp(-d1 * a)) * (1.0 + Es * exp(-d1 * cc));
    double v = log(num / den);
    if (y_ji) v += (d0 - d1) * a;     // Y[j, i] : out direction of j
    if (y_ij) v += (d0 - d1) * cc;    // Y[i, j] : in direction of j
    return v;
}

// ---------------------------------------------------------------------------
// v1: one 1024-thread workgroup per time slice.  X[t] (and 1/radii) live in
// LDS; row j+1 of the bit-packed network is prefetched into an LDS double
// buffer while step j computes; proposals, log-uniforms and prior deltas of a
// chunk of steps are generated in parallel up front (counter RNG: no state to
// carry).  One barrier per MH step.
// ---------------------------------------------------------------------------
template <int D, int MODEL>
__global__ __launch_bounds__(SW_THREADS) void k_sweep_slice(ChainView c,
                                                            IterRef ir,
                                                            int parity) {
    const uint32_t iter = ir.get();
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int N = c.N, W = c.W;
    const int t = 2 * blockIdx.x + parity;
    if (t >= c.T) return;
    constexpr int PW = PropRec<D>::W;
    double *sX = smem;                                   // N*D
    double *sIR = sX + (size_t)N * D;                    // N (directed)
    double *sProp = sIR + (MODEL == DLSM_DIRECTED ? N : 0);   // SW_CHUNK*PW
    double *sRed = sProp + SW_CHUNK * PW;                // 2*SW_WAVES
    uint32_t *sRow = (uint32_t *)(sRed + 2 * SW_WAVES);  // 2*W (+2*W directed)
    uint32_t *sCol = sRow + 2 * W;
    const int tid = threadIdx.x;
    double *Xt = c.X + (size_t)t * N * D;
    for (int k = tid; k < N * D; k += SW_THREADS) sX[k] = Xt[k];
    if (MODEL == DLSM_DIRECTED)
        for (int k = tid; k < N; k += SW_THREADS) sIR[k] = 1.0 / c.radii[k];
    const uint32_t *rows = c.ybits + (size_t)t * N * W;
    const uint32_t *cols = MODEL == DLSM_DIRECTED ? c.ytbits + (size_t)t * N * W
                                                  : nullptr;
    if (tid < W) {
        sRow[tid] = rows[tid];
        if (MODEL == DLSM_DIRECTED) sCol[tid] = cols[tid];
    }
    double E, bin = 0.0, bout = 0.0;
    if (MODEL == DLSM_UNDIRECTED) {
        E = exp(c.intercept[0]);
    } else {
        bin = c.intercept[0]; bout = c.intercept[1];
        E = exp(bin + bout);
    }
    for (int j0 = 0; j0 < N; j0 += SW_CHUNK) {
        __syncthreads();     // sX loaded / previous chunk's records consumed
        const int nj = min(SW_CHUNK, N - j0);
        if (tid < nj) {
            const int j = j0 + tid;
            double x0[D], x1[D], logu;
#pragma unroll
            for (int d = 0; d < D; ++d) x0[d] = sX[j * D + d];
            make_proposal<D>(c, iter, t, j, x0, c.step[(size_t)t * N + j], x1, logu);
            const double pd = node_log_prior<D>(c, t, j, x1) -
                              node_log_prior<D>(c, t, j, x0);
            double *rec = sProp + tid * PW;
#pragma unroll
            for (int d = 0; d < D; ++d) { rec[d] = x0[d]; rec[D + d] = x1[d]; }
            rec[2 * D] = logu;
            rec[2 * D + 1] = pd;
        }
        __syncthreads();
        for (int jj = 0; jj < nj; ++jj) {
            const int j = j0 + jj;
            const int cur = j & 1;
            // prefetch row j+1 of the network into the other LDS buffer
            uint32_t wrow = 0, wcol = 0;
            const bool pf = (tid < W) && (j + 1 < N);
            if (pf) {
                wrow = rows[(size_t)(j + 1) * W + tid];
                if (MODEL == DLSM_DIRECTED) wcol = cols[(size_t)(j + 1) * W + tid];
            }
            const double *rec = sProp + jj * PW;
            double x0[D], x1[D];
#pragma unroll
            for (int d = 0; d < D; ++d) { x0[d] = rec[d]; x1[d] = rec[D + d]; }
            const double logu = rec[2 * D], pd = rec[2 * D + 1];
            const uint32_t *yr = sRow + cur * W;
            const uint32_t *yc = sCol + cur * W;
            const double irj = MODEL == DLSM_DIRECTED ? sIR[j] : 0.0;
            double acc = 0.0;
            for (int i = tid; i < N; i += SW_THREADS) {
                if (i == j) continue;
                const double d0 = dist_of<D>(&sX[i * D], x0, c.squared);
                const double d1 = dist_of<D>(&sX[i * D], x1, c.squared);
                if (MODEL == DLSM_UNDIRECTED) {
                    acc += delta_undirected(d0, d1, bit_of(yr, i), E);
                } else {
                    const double iri = sIR[i];
                    acc += delta_directed(d0, d1, bit_of(yr, i), bit_of(yc, i),
                                          bin * iri + bout * irj,
                                          bin * irj + bout * iri, E);
                }
            }
            if (pf) {
                sRow[(cur ^ 1) * W + tid] = wrow;
                if (MODEL == DLSM_DIRECTED) sCol[(cur ^ 1) * W + tid] = wcol;
            }
            const double total = block_sum_all<SW_WAVES>(acc, sRed + cur * SW_WAVES, tid);
            const double ratio = total + pd;
            const int accepted = !(logu >= ratio);      // metropolis.py:50
            if (tid == (j & (SW_THREADS - 1))) {
                if (accepted) {
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        sX[j * D + d] = x1[d];
                        Xt[(size_t)j * D + d] = x1[d];
                    }
                }
                const size_t tj = (size_t)t * N + j;
                double st = c.step[tj];
                int32_t na = c.nacc[tj], ns = c.nsteps[tj], un = c.until[tj];
                metropolis_bookkeeping(st, na, ns, un, c.tune, c.tune_interval, accepted);
                c.step[tj] = st; c.nacc[tj] = na; c.nsteps[tj] = ns; c.until[tj] = un;
            }
        }
    }
}

inline size_t sweep_slice_lds_bytes(int N, int D, int W, int model) {
    size_t dbl = (size_t)N * D + (model == DLSM_DIRECTED ? N : 0) +
                 (size_t)SW_CHUNK * (2 * D + 2) + 2 * SW_WAVES;
    size_t words = (size_t)(model == DLSM_DIRECTED ? 4 : 2) * W;
    return dbl * sizeof(double) + words * sizeof(uint32_t);
}

// ---------------------------------------------------------------------------
// Case-control sweep (a3 inside a9/a10): O(deg + 2C) gathered terms per step.
// 256 threads per slice; X stays in global memory (L2 resident; sc1 loads so a
// position written by this workgroup earlier in the sweep is always re-read
// from L2).  nctrl[T][N][2] = number of valid (non -1) in / out controls.
// ---------------------------------------------------------------------------
constexpr int CC_THREADS = 256;

__device__ __forceinline__ double load_sc1(const double *p) {
    unsigned long long v = __hip_atomic_load((const unsigned long long *)p,
                                             __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
    return __longlong_as_double((long long)v);
}

template <int D>
__global__ __launch_bounds__(CC_THREADS) void k_sweep_casecontrol(
    ChainView c, const int32_t *__restrict__ nctrl, IterRef ir, int parity) {
    const uint32_t iter = ir.get();
    constexpr int PW = PropRec<D>::W;
    __shared__ double sProp[SW_CHUNK * PW];
    __shared__ double sRed[2 * (CC_THREADS / 64)];
    const int N = c.N;
    const int t = 2 * blockIdx.x + parity;
    if (t >= c.T) return;
    const int tid = threadIdx.x;
    double *Xt = c.X + (size_t)t * N * D;
    const double bin = c.intercept[0], bout = c.intercept[1];
    for (int j0 = 0; j0 < N; j0 += SW_CHUNK) {
        __syncthreads();
        const int nj = min(SW_CHUNK, N - j0);
        for (int q = tid; q < nj; q += CC_THREADS) {
            const int j = j0 + q;
            double x0[D], x1[D], logu;
#pragma unroll
            for (int d = 0; d < D; ++d) x0[d] = load_sc1(&Xt[(size_t)j * D + d]);
            make_proposal<D>(c, iter, t, j, x0, c.step[(size_t)t * N + j], x1, logu);
            const double pd = node_log_prior<D>(c, t, j, x1) -
                              node_log_prior<D>(c, t, j, x0);
            double *rec = sProp + q * PW;
#pragma unroll
            for (int d = 0; d < D; ++d) { rec[d] = x0[d]; rec[D + d] = x1[d]; }
            rec[2 * D] = logu;
            rec[2 * D + 1] = pd;
        }
        __syncthreads();
        for (int jj = 0; jj < nj; ++jj) {
            const int j = j0 + jj;
            const size_t node = (size_t)t * N + j;
            const double *rec = sProp + jj * PW;
            double x0[D], x1[D];
#pragma unroll
            for (int d = 0; d < D; ++d) { x0[d] = rec[d]; x1[d] = rec[D + d]; }
            const double logu = rec[2 * D], pd = rec[2 * D + 1];
            const int in_deg = c.degree[node * 2], out_deg = c.degree[node * 2 + 1];
            const int nci = nctrl[node * 2], nco = nctrl[node * 2 + 1];
            const double rj = c.radii[j];
            const double adj_in = (double)(N - in_deg - 1) / (double)nci;
            const double adj_out = (double)(N - out_deg - 1) / (double)nco;
            const int total_terms = in_deg + out_deg + nci + nco;
            double acc = 0.0;
            for (int k = tid; k < total_terms; k += CC_THREADS) {
                int e, kind = 0, q = k;          // kind: 0 in-edge 1 out-edge 2 ctrl-in 3 ctrl-out
                if (q < in_deg) { e = c.in_edges[node * c.Din + q]; kind = 0; }
                else if ((q -= in_deg) < out_deg) { e = c.out_edges[node * c.Dout + q]; kind = 1; }
                else if ((q -= out_deg) < nci) { e = c.ctrl_in[node * c.C + q]; kind = 2; }
                else { q -= nci; e = c.ctrl_out[node * c.C + q]; kind = 3; }
                double xe[D];
#pragma unroll
                for (int d = 0; d < D; ++d) xe[d] = load_sc1(&Xt[(size_t)e * D + d]);
                const double re = c.radii[e];
                // a self reference sees the moved position (distance 0 both ways)
                const double d0 = e == j ? 0.0 : dist_of<D>(xe, x0, c.squared);
                const double d1 = e == j ? 0.0 : dist_of<D>(xe, x1, c.squared);
                const bool in_dir = (kind == 0 || kind == 2);
                const double a = in_dir ? (bin / rj + bout / re) : (bin / re + bout / rj);
                // literal form: eta = b_in (1 - d / r_.) + b_out (1 - d / r_.)
                const double eta0 = in_dir ? bin * (1 - d0 / rj) + bout * (1 - d0 / re)
                                           : bin * (1 - d0 / re) + bout * (1 - d0 / rj);
                const double eta1 = in_dir ? bin * (1 - d1 / rj) + bout * (1 - d1 / re)
                                           : bin * (1 - d1 / re) + bout * (1 - d1 / rj);
                (void)a;
                const double sp = log((1.0 + exp(eta1)) / (1.0 + exp(eta0)));
                if (kind < 2) acc += (eta1 - eta0) - sp;
                else acc -= (kind == 2 ? adj_in : adj_out) * sp;
            }
            const double total = block_sum_all<CC_THREADS / 64>(
                acc, sRed + (j & 1) * (CC_THREADS / 64), tid);
            const double ratio = total + pd;
            const int accepted = !(logu >= ratio);
            if (tid == 0) {
                if (accepted) {
#pragma unroll
                    for (int d = 0; d < D; ++d)
                        __hip_atomic_store((unsigned long long *)&Xt[(size_t)j * D + d],
                                           (unsigned long long)__double_as_longlong(x1[d]),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                double st = c.step[node];
                int32_t na = c.nacc[node], ns = c.nsteps[node], un = c.until[node];
                metropolis_bookkeeping(st, na, ns, un, c.tune, c.tune_interval, accepted);
                c.step[node] = st; c.nacc[node] = na; c.nsteps[node] = ns; c.until[node] = un;
            }
            // the store above must be visible to the gathers of the next step
            __syncthreads();
        }
    }
}

__global__ void k_count_controls(const int32_t *__restrict__ ctrl_in,
                                 const int32_t *__restrict__ ctrl_out, long nodes,
                                 int C, int32_t *__restrict__ nctrl) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nodes) return;
    int a = 0, b = 0;
    while (a < C && ctrl_in[i * C + a] >= 0) ++a;
    while (b < C && ctrl_out[i * C + b] >= 0) ++b;
    nctrl[i * 2] = a;
    nctrl[i * 2 + 1] = b;
}

// ---------------------------------------------------------------------------
// Post-sweep glue (two multi-workgroup passes, k_post_reduce / k_post_apply):
//   1. optional Procrustes rotation to X_ref (procrustes.py:20-35):
//        M = X^T X_ref, SVD M = U S V^T (one-sided Jacobi), R = U V^T, X <- X R
//   2. centring X -= mean over (t, i) (lsm.py:501)
//   3. optional LSM bookkeeping: latent prior terms of logp (lsm.py:604-613)
//      and the intercept proposal + log-uniform of sample_intercepts
//      (sample_coefficients.py:76-86) with Philox stream INTERCEPT.
// ---------------------------------------------------------------------------

template <int D>
__device__ void jacobi_polar(const double (&M)[D][D], double (&R)[D][D]) {
    double U[D][D], V[D][D];
    for (int a = 0; a < D; ++a)
        for (int b = 0; b < D; ++b) { U[a][b] = M[a][b]; V[a][b] = a == b ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < D - 1; ++p)
            for (int q = p + 1; q < D; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int k = 0; k < D; ++k) {
                    al += U[k][p] * U[k][p];
                    be += U[k][q] * U[k][q];
                    ga += U[k][p] * U[k][q];
                }
                if (fabs(ga) <= 1e-300 || fabs(ga) <= 1e-16 * sqrt(al * be)) continue;
                off += fabs(ga);
                const double zeta = (be - al) / (2.0 * ga);
                const double tt = (zeta >= 0 ? 1.0 : -1.0) /
                                  (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
                for (int k = 0; k < D; ++k) {
                    double up = U[k][p], uq = U[k][q];
                    U[k][p] = cs * up - sn * uq;
                    U[k][q] = sn * up + cs * uq;
                    double vp = V[k][p], vq = V[k][q];
                    V[k][p] = cs * vp - sn * vq;
                    V[k][q] = sn * vp + cs * vq;
                }
            }
        if (off == 0.0) break;
    }
    for (int q = 0; q < D; ++q) {
        double nrm = 0;
        for (int k = 0; k < D; ++k) nrm += U[k][q] * U[k][q];
        nrm = sqrt(nrm);
        for (int k = 0; k < D; ++k) U[k][q] /= nrm;
    }
    for (int a = 0; a < D; ++a)
        for (int b = 0; b < D; ++b) {
            double s = 0;
            for (int k = 0; k < D; ++k) s += U[a][k] * V[b][k];
            R[a][b] = s;
        }
}

// Pass 1 (many workgroups): per-workgroup sums over the rows r of X (T*N x D):
//   [ sum x (D) | M = X^T X_ref (D*D) | sum_{t=0} x (D) | sum_{t=0} |x|^2 |
//     sum_{t>0} |x_t - x_{t-1}|^2 ]
// Everything pass 2 needs follows from these by linearity / orthogonal invariance:
// mean(X R) = mean(X) R, |(x - m) R| = |x - m|, |(x_t - x_{t-1}) R| = |x_t - x_{t-1}|.
constexpr int PS_BLOCKS = 64;
constexpr int PS2_THREADS = 256;
template <int D> struct PostRec { static constexpr int W = 2 * D + D * D + 2; };
constexpr int POST_W_MAX = PostRec<DLSM_D_MAX>::W;      // buffers sized before the dimension is dispatched

// one row's contributions to the W sums from its operands (x, the row of the slice before, the
// reference's row).  own: everything that needs the row itself (sum x, M, the t = 0 sums); diff: the
// row's |x_t - x_{t-1}|^2 (rows of t >= 1)
template <int D>
__device__ __forceinline__ void post_row_accumulate(const double (&x)[D], const double (&xp)[D],
                                                    const double (&xr)[D], bool t0, bool own, bool diff,
                                                    bool has_ref, double (&acc)[PostRec<D>::W]) {
    if (own) {
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] += x[d];
        if (has_ref) {
#pragma unroll
            for (int a = 0; a < D; ++a)
#pragma unroll
                for (int b = 0; b < D; ++b) acc[D + a * D + b] += x[a] * xr[b];
        }
    }
    // (the last two sums get their addend - or an exact 0.0 - by name: written as `acc[..] += q` in
    // the two branches the compiler merged the updates into ONE access with a selected index, and the
    // pair lived in scratch memory: 12 scratch instructions in k_pipe_last_ride, once per iteration)
    double q0 = 0.0, q1 = 0.0;
    if (t0) {
        if (own) {
#pragma unroll
            for (int d = 0; d < D; ++d) { acc[D + D * D + d] += x[d]; q0 += x[d] * x[d]; }
        }
    } else if (diff) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const double df = x[d] - xp[d];
            q1 += df * df;
        }
    }
    acc[2 * D + D * D] += q0;
    acc[2 * D + D * D + 1] += q1;
}
template <int D>
__device__ __forceinline__ void post_row_load(const ChainView &c, const double *__restrict__ xref, long r,
                                              bool own, bool diff, double (&x)[D], double (&xp)[D],
                                              double (&xr)[D]) {
    const double *X = c.X;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        x[d] = X[r * D + d];
        xp[d] = r >= c.N && diff ? X[(r - c.N) * D + d] : 0.0;
        xr[d] = own && xref ? xref[r * D + d] : 0.0;
    }
}
template <int D>
__device__ __forceinline__ void post_row_terms(const ChainView &c, const double *__restrict__ xref,
                                               long r, bool own, bool diff, double (&acc)[PostRec<D>::W]) {
    double x[D], xp[D], xr[D];
    post_row_load<D>(c, xref, r, own, diff, x, xp, xr);
    post_row_accumulate<D>(x, xp, xr, r < c.N, own, diff, xref != nullptr, acc);
}

// Rows a pass over "what is final" leaves out (k_pipe_last_ride: the sums ride in the sweep's last
// launch, which still moves the nodes i >= jl of the slices of parity `par`): such a row's own terms,
// and the difference terms of every row i >= jl of t >= 1 (one of t, t - 1 has that parity); the
// launch's resolver workgroups add them.  jl < 0: nothing is left out.
__device__ __forceinline__ bool post_row_own_left(int t, int i, int jl, int par) {
    return jl >= 0 && i >= jl && (t & 1) == par;
}
__device__ __forceinline__ bool post_row_diff_left(int t, int i, int jl) { return jl >= 0 && i >= jl && t >= 1; }

template <int D, int NT = PS2_THREADS>
__device__ __forceinline__ void post_reduce_wg(
    const ChainView &c, const double *__restrict__ xref_in, int n_iter_procrustes, IterRef ir,
    double *__restrict__ rec, int wg, int nwg, int jl = -1, int par = 0) {
    // lsm.py:495: rotate only once it > tune + burn (n_iter_procrustes < 0: always)
    const double *xref = (xref_in && (n_iter_procrustes < 0 ||
                                      (int)ir.get() > n_iter_procrustes)) ? xref_in : nullptr;
    constexpr int W = PostRec<D>::W;
    __shared__ double sRed[2 * (NT / 64)];
    const int tid = threadIdx.x;
    const long rows = (long)c.T * c.N;
    double acc[W];
#pragma unroll
    for (int q = 0; q < W; ++q) acc[q] = 0.0;
    for (long r = (long)wg * NT + tid; r < rows; r += (long)nwg * NT) {
        const int t = (int)(r / c.N), i = (int)(r - (long)t * c.N);
        post_row_terms<D>(c, xref, r, !post_row_own_left(t, i, jl, par), !post_row_diff_left(t, i, jl), acc);
    }
#pragma unroll
    for (int q = 0; q < W; ++q) {
        const double v = block_sum_all<NT / 64>(acc[q], sRed + (q & 1) * (NT / 64), tid);
        if (tid == 0) rec[(size_t)wg * W + q] = v;
    }
}

template <int D>
__global__ __launch_bounds__(PS2_THREADS) void k_post_reduce(
    ChainView c, const double *__restrict__ xref_in, int n_iter_procrustes, IterRef ir,
    double *__restrict__ rec) {
    post_reduce_wg<D>(c, xref_in, n_iter_procrustes, ir, rec, (int)blockIdx.x, (int)gridDim.x);
}


// directed models: cand = [proposal pair | current pair] for intercept `which` (one thread;
// sample_coefficients.py:12-75)
__device__ __forceinline__ void dir_propose_intercept(const ChainView &c, LsmDeviceState *lsm,
                                                      const double *__restrict__ intercept,
                                                      int which, uint32_t iter) {
    double u0, u1, z0, z1;
    philox_uniform2(c.seed, (uint32_t)which, 0, iter, stream_word(c.chain, STREAM_INTERCEPT), u0, u1);
    box_muller(u0, u1, z0, z1);
    const double b0 = intercept[0], b1 = intercept[1];
    lsm->cand[0] = which == 0 ? b0 + lsm->i_step[0] * z0 : b0;
    lsm->cand[1] = which == 1 ? b1 + lsm->i_step[1] * z0 : b1;
    lsm->cand[2] = b0;
    lsm->cand[3] = b1;
    philox_uniform2(c.seed, (uint32_t)which, 1, iter, stream_word(c.chain, STREAM_INTERCEPT), u0, u1);
    lsm->logu = log(u0);
}
// ... and BOTH intercept steps' proposals at once (the case-control loop's four-candidate pass): the second
// step's proposal b_out' = b_out + step_out z and its uniform are functions of (seed, iteration) and of a
// step size the first step does not touch, so they can be drawn before the first step is decided
// (sample_coefficients.py:12-75: the same draws, the same two accept / reject rules in the same order).
// Leaves cand / logu as dir_propose_intercept(which = 0) does.
__device__ __forceinline__ void dir_propose_both(const ChainView &c, LsmDeviceState *lsm,
                                                 const double *__restrict__ intercept, uint32_t iter) {
    dir_propose_intercept(c, lsm, intercept, 0, iter);
    double u0, u1, z0, z1;
    philox_uniform2(c.seed, 1u, 0, iter, stream_word(c.chain, STREAM_INTERCEPT), u0, u1);
    box_muller(u0, u1, z0, z1);
    const double b0 = intercept[0], b1 = intercept[1];
    const double p0 = lsm->cand[0], p1 = b1 + lsm->i_step[1] * z0;
    lsm->cand8[0] = p0; lsm->cand8[1] = b1;
    lsm->cand8[2] = b0; lsm->cand8[3] = b1;
    lsm->cand8[4] = p0; lsm->cand8[5] = p1;
    lsm->cand8[6] = b0; lsm->cand8[7] = p1;
    philox_uniform2(c.seed, 1u, 1, iter, stream_word(c.chain, STREAM_INTERCEPT), u0, u1);
    lsm->logu2 = log(u0);
}
// Pass 2 (many workgroups): every workgroup sums the records in the same fixed
// order, gets R (one-sided Jacobi polar factor of M) and the mean, and applies
// x <- x R - mean R to its rows.  Workgroup 0 also leaves the LSM bookkeeping.
// xr_keep_alt: leave the records' second radius slot alone (the radii proposal, riding in the same
// launch, files itself there)
struct PostNoHook { __device__ __forceinline__ void operator()(long, const double *) const {} };
// hook(r, y): called with every row this workgroup has just centred (the fused last launch of the
// undirected loop draws the next sweep's proposal from it); do_rows = false: the sums, R, the
// shift and workgroup 0's bookkeeping only; draw_intercept = 0: the intercept proposal is not drawn
template <int D, class RowHook = PostNoHook>
__device__ __forceinline__ void post_apply_wg(
    const ChainView &c, int has_ref, int n_iter_procrustes, int do_center,
    const double *__restrict__ rec, int nrec, LsmDeviceState *lsm, IterRef ir,
    double *__restrict__ R_out, double *__restrict__ trace_X, double *__restrict__ xr, int xr_keep_alt,
    int wg, int nwg, int jl = -1, int par = 0, const double *__restrict__ xref_rows = nullptr,
    bool do_rows = true, int draw_intercept = 1, RowHook hook = RowHook()) {
    const uint32_t iter = ir.get();
    const int rotate = has_ref && (n_iter_procrustes < 0 || (int)iter > n_iter_procrustes);
    constexpr int W = PostRec<D>::W;
    __shared__ double sSum[W];
    __shared__ double sR[D * D];
    __shared__ double sShift[D];
    __shared__ double sRec[PS_BLOCKS * W];
    const int tid = threadIdx.x;
    const long rows = (long)c.T * c.N;
    double *X = c.X;
    // records -> LDS with all loads in flight, then a fixed-order sum per column
    for (int q = tid; q < nrec * W; q += PS2_THREADS) sRec[q] = rec[q];
    __syncthreads();
    if (tid < W) {
        double s = 0.0;
        for (int b = 0; b < nrec; ++b) s += sRec[b * W + tid];
        sSum[tid] = s;
    }
    __syncthreads();
    if (tid == 0) {
        double R[D][D];
        if (rotate && do_rows) {        // (a workgroup that centres no rows needs the mean only)
            double M[D][D];
            for (int a = 0; a < D; ++a)
                for (int b = 0; b < D; ++b) M[a][b] = sSum[D + a * D + b];
            jacobi_polar<D>(M, R);
        } else {
            for (int a = 0; a < D; ++a)
                for (int b = 0; b < D; ++b) R[a][b] = a == b ? 1.0 : 0.0;
        }
        double mean[D];
        for (int d = 0; d < D; ++d) mean[d] = do_center ? sSum[d] / (double)rows : 0.0;
        for (int b = 0; b < D; ++b) {
            double sh = 0.0;
            for (int a = 0; a < D; ++a) { sh += mean[a] * R[a][b]; sR[a * D + b] = R[a][b]; }
            sShift[b] = sh;
        }
        if (wg == 0) {
            if (R_out)
                for (int a = 0; a < D * D; ++a) R_out[a] = sR[a];
            if (lsm) {
                // lsm.py:604-613 on the rotated, centred positions
                double q0 = sSum[2 * D + D * D];
                double mm = 0.0, ms = 0.0;
                for (int d = 0; d < D; ++d) {
                    mm += mean[d] * mean[d];
                    ms += mean[d] * sSum[D + D * D + d];
                }
                q0 = q0 - 2.0 * ms + (double)c.N * mm;
                lsm->prior_x = -(0.5 * q0 / c.tau_sq +
                                 0.5 * sSum[2 * D + D * D + 1] / c.sigma_sq);
                if (!draw_intercept) {
                    // (drawn with the sweep's proposals: pipe_propose_intercept)
                } else if (c.model != DLSM_UNDIRECTED) {
                    // the first of the two intercept steps of the directed loops (and, for the case-control
                    // loop's four-candidate pass, the second one's proposal with it)
                    dir_propose_both(c, lsm, c.intercept, iter);
                } else {
                    double u0, u1, z0, z1;
                    philox_uniform2(c.seed, 0, 0, iter, stream_word(c.chain, STREAM_INTERCEPT),
                                    u0, u1);
                    box_muller(u0, u1, z0, z1);
                    const double b0 = c.intercept[0];
                    lsm->cand[0] = b0;
                    lsm->cand[1] = b0 + lsm->i_step[0] * z0;
                    philox_uniform2(c.seed, 0, 1, iter, stream_word(c.chain, STREAM_INTERCEPT),
                                    u0, u1);
                    lsm->logu = log(u0);
                }
            }
        }
    }
    __syncthreads();
    if (!do_rows || (!rotate && !do_center && !trace_X && !xr)) return;
    // the device-resident loop also files the final positions as row `iter` of its trace
    double *trow = trace_X ? trace_X + (size_t)iter * rows * D : nullptr;
    for (long r = (long)wg * PS2_THREADS + tid; r < rows;
         r += (long)nwg * PS2_THREADS) {
        double x[D], y[D];
#pragma unroll
        for (int a = 0; a < D; ++a) x[a] = X[r * D + a];
#pragma unroll
        for (int b = 0; b < D; ++b) {
            double sacc = 0.0;
#pragma unroll
            for (int a = 0; a < D; ++a) sacc += x[a] * sR[a * D + b];
            y[b] = sacc - sShift[b];
        }
#pragma unroll
        for (int b = 0; b < D; ++b) X[r * D + b] = y[b];
        if (trow) {
#pragma unroll
            for (int b = 0; b < D; ++b) trow[r * D + b] = y[b];
        }
        hook(r, y);
        if (xr) {       // the case-control log-likelihood's gather records (k_pack_xr's): [x | 1 / r | 1 / r]
            constexpr int RW = llcc_record_width(D);
            double rc[RW];
#pragma unroll
            for (int d = 0; d < RW; ++d) rc[d] = 0.0;
#pragma unroll
            for (int d = 0; d < D; ++d) rc[d] = y[d];
            rc[D] = rc[D + 1] = 1.0 / c.radii[r % c.N];
            if (xr_keep_alt) {
#pragma unroll
                for (int d = 0; d < RW; ++d)
                    if (d != D + 1) xr[r * RW + d] = rc[d];
            } else {
#pragma unroll
                for (int d = 0; d < RW; d += 2)
                    *(double2 *)(xr + r * RW + d) = make_double2(rc[d], rc[d + 1]);
            }
        }
    }
}

template <int D>
__global__ __launch_bounds__(PS2_THREADS) void k_post_apply(
    ChainView c, int has_ref, int n_iter_procrustes, int do_center,
    const double *__restrict__ rec, int nrec, LsmDeviceState *lsm, IterRef ir,
    double *__restrict__ R_out, double *__restrict__ trace_X, double *__restrict__ xr = nullptr,
    int jl = -1, int par = 0, const double *__restrict__ xref_rows = nullptr) {
    post_apply_wg<D>(c, has_ref, n_iter_procrustes, do_center, rec, nrec, lsm, ir, R_out, trace_X, xr, 0,
                     (int)blockIdx.x, (int)gridDim.x, jl, par, xref_rows);
}


// ---------------------------------------------------------------------------
// End of an undirected LSM iteration: fixed-order reduction of the fused
// log-likelihood records, intercept accept/reject + step-size adaptation
// (sample_coefficients.py:76-86, metropolis.py:96-136), log-posterior trace
// (lsm.py:576-625).  One workgroup.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void lsm_finalize_wg(
    const double *__restrict__ partials, int nrec, LsmDeviceState *lsm,
    double *__restrict__ intercept, double *__restrict__ trace_ic,
    double *__restrict__ trace_logp, IterRef ir) {
    const int it = (int)ir.get();
    __shared__ double scratch[4 * 256];
    __shared__ double sums[4];
    reduce_records(partials, nrec, 4, sums, scratch, threadIdx.x);
    if (threadIdx.x == 0) {
        const double b0 = lsm->cand[0], b1 = lsm->cand[1];
        const double ll0 = b0 * sums[0] - sums[1] - sums[2];
        const double ll1 = b1 * sums[0] - sums[1] - sums[3];
        const double pm = lsm->intercept_prior[0], v = lsm->intercept_var;
        const double lp0 = ll0 - (b0 - pm) * (b0 - pm) / (2 * v);
        const double lp1 = ll1 - (b1 - pm) * (b1 - pm) / (2 * v);
        const int accepted = !(lsm->logu >= lp1 - lp0);
        const double b = accepted ? b1 : b0;
        const double ll = accepted ? ll1 : ll0;
        intercept[0] = b;
        double st = lsm->i_step[0];
        int32_t na = lsm->i_nacc[0], ns = lsm->i_nsteps[0], un = lsm->i_until[0];
        metropolis_bookkeeping(st, na, ns, un, lsm->i_tune, lsm->i_tune_interval,
                               accepted);
        lsm->i_step[0] = st; lsm->i_nacc[0] = na; lsm->i_nsteps[0] = ns;
        lsm->i_until[0] = un;
        trace_ic[(size_t)it * 2] = b;
        trace_ic[(size_t)it * 2 + 1] = 0.0;
        trace_logp[it] = ll + lsm->prior_x - 0.5 * (b - pm) * (b - pm) / v;
    }
}

__global__ __launch_bounds__(256) void k_lsm_finalize(
    const double *__restrict__ partials, int nrec, LsmDeviceState *lsm,
    double *__restrict__ intercept, double *__restrict__ trace_ic,
    double *__restrict__ trace_logp, IterRef ir) {
    lsm_finalize_wg(partials, nrec, lsm, intercept, trace_ic, trace_logp, ir);
}

// iteration counter of the captured-graph path
__global__ void k_advance_iter(uint32_t *p) { *p += 1u; }

}  // namespace dlsm
