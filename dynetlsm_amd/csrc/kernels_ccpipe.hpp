// Pipelined speculative sweep for the case-control likelihood with SPARSE corrections
// (algo 5; a3 inside a9 / a10: directed_likelihoods_fast.pyx:83-182 under
// sample_latent_positions.py:92-206).
//
// The dense form (kernels_spec_pipe.hpp, MODEL = case-control) keeps the corrections H[k][m]
// - how the acceptance of node m changes node k's ratio - in 128 x 128 blocks, which caps a
// batch at 128 nodes: 81 latency-bound launches per sweep at T = 5, N = 10 000.  But node k's
// ratio only depends on the O(deg + 2C) nodes in its edge / control lists, so of the nodes not
// yet resolved when k is evaluated (the window: the previous batch and the earlier nodes of
// k's own batch) only a few matter: ~40 of 2048 at C4 with batches of 1024.  Here the
// evaluator files exactly those as a LIST per node, (window index, correction) pairs in term
// order, and the resolver runs the same fixed-point solve of the in-order accept / reject
// rule over the lists:
//
//     a_k = [ log u_k < r_k + sum_{(m, h) in list_k, m accepted} h ]
//
// A fixed point of a -> F(a) satisfies the triangular system, whose solution is unique, so it
// is the sequential scan's result.  Batches of 1024 nodes: ceil(N / 1024) + 2 launches per
// sweep (12 at C4), each with T resolver workgroups beside ~250 evaluator workgroups (one
// wavefront per (node, quarter of its terms)).  Same snapshot rule, same one-batch lag of the odd
// slices, same decisions as the dense form and the scalar oracle.
#pragma once
#include "kernels_spec_pipe.hpp"

namespace dlsm {

constexpr int CP_B = 1024;              // nodes per batch = threads of the resolver
constexpr int CP_THREADS = 1024;
constexpr int CP_WAVES = CP_THREADS / 64;
constexpr int CP_SUBS = 4;              // wavefronts per node in the evaluator

struct CcPipeBuf {
    double *prop;            // [T][N][2D + 2] : x1[D], u, (unused), x0[D] (snapshot)
    double *tot;             // [2][T][CP_B][CP_SUBS] : partial log-ratio of node k (snapshot neighbours)
    double *hval;            // [2][T][CP_B][CP_SUBS][cap] : corrections, in term order
    int32_t *hidx;           // same shape: window index m in [0, 2 CP_B): previous batch, then own
    int32_t *hcnt;           // [2][T][CP_B][CP_SUBS]
    unsigned long long *accmask;   // [T][CP_WAVES] : accepted nodes of the last resolved batch
    const int32_t *nctrl;    // valid controls per (t, i, direction)
    int cap, nbat;
};

// log-ratio contribution of one gathered term of node k when k moves x0 -> x1, its partner
// at xn (directed_likelihoods_fast.pyx:107-180):
//   edges    (eta1 - eta0) - [softplus(eta1) - softplus(eta0)]
//   controls - adj [softplus(eta1) - softplus(eta0)],  adj = (N - deg - 1) / n_controls
template <int D>
__device__ __forceinline__ double cc_term_delta(const double *xn, const double *xk0,
                                                const double *xk1, bool self, bool in_dir,
                                                bool edge, double wsp, double bin, double bout,
                                                double rj, double re, int squared) {
    const double d0 = self ? 0.0 : dist_of<D>(xn, xk0, squared);
    const double d1 = self ? 0.0 : dist_of<D>(xn, xk1, squared);
    const double e0 = in_dir ? bin * (1 - d0 / rj) + bout * (1 - d0 / re)
                             : bin * (1 - d0 / re) + bout * (1 - d0 / rj);
    const double e1 = in_dir ? bin * (1 - d1 / rj) + bout * (1 - d1 / re)
                             : bin * (1 - d1 / re) + bout * (1 - d1 / rj);
    const double sp = log((1.0 + exp(e1)) / (1.0 + exp(e0)));
    return (edge ? (e1 - e0) : 0.0) - wsp * sp;
}

// One wavefront: quarter `sub` of the terms of node k of batch `be` in slice t.
template <int D>
__device__ __forceinline__ void ccpipe_eval_item(const ChainView &c, const CcPipeBuf &pb, int be,
                                                 int t, int k, int sub, int lane) {
    constexpr int PW = 2 * D + 2;
    const int N = c.N;
    const int j0 = be * CP_B, jk = j0 + k;
    const int jprev = max(0, j0 - CP_B);       // nodes >= jprev: snapshot positions
    const int bb = be & 1;
    const size_t node = (size_t)t * N + jk;
    const double *Xt = c.X + (size_t)t * N * D;
    const double *props = pb.prop + (size_t)t * N * PW;
    double xk0[D], xk1[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        xk0[d] = props[(size_t)jk * PW + D + 2 + d];
        xk1[d] = props[(size_t)jk * PW + d];
    }
    const double bin = c.intercept[0], bout = c.intercept[1];
    const double rj = c.radii[jk];
    const int in_deg = c.degree[node * 2], out_deg = c.degree[node * 2 + 1];
    const int nci = pb.nctrl[node * 2], nco = pb.nctrl[node * 2 + 1];
    const double adj_in = (double)(N - in_deg - 1) / (double)nci;
    const double adj_out = (double)(N - out_deg - 1) / (double)nco;
    const int total_terms = in_deg + out_deg + nci + nco;
    const size_t slot = (((size_t)bb * c.T + t) * CP_B + k) * CP_SUBS + sub;
    double *hval = pb.hval + slot * pb.cap;
    int32_t *hidx = pb.hidx + slot * pb.cap;
    const unsigned long long below = (1ull << lane) - 1ull;
    double acc = 0.0;
    int cnt = 0;
    for (int q0 = 64 * sub; q0 < total_terms; q0 += 64 * CP_SUBS) {
        const int q = q0 + lane;
        int e = -1, kind = 0;
        if (q < total_terms) {
            int r = q;
            if (r < in_deg) { e = c.in_edges[node * c.Din + r]; kind = 0; }
            else if ((r -= in_deg) < out_deg) { e = c.out_edges[node * c.Dout + r]; kind = 1; }
            else if ((r -= out_deg) < nci) { e = c.ctrl_in[node * c.C + r]; kind = 2; }
            else { r -= nci; e = c.ctrl_out[node * c.C + r]; kind = 3; }
        }
        bool inwin = false;
        double h = 0.0;
        if (e >= 0) {
            const double *src = e < jprev ? Xt + (size_t)e * D : props + (size_t)e * PW + D + 2;
            double xe[D];
#pragma unroll
            for (int d = 0; d < D; ++d) xe[d] = src[d];
            const double re = c.radii[e];
            const bool in_dir = (kind == 0 || kind == 2);
            const double wsp = kind < 2 ? 1.0 : (kind == 2 ? adj_in : adj_out);
            const double contrib = cc_term_delta<D>(xe, xk0, xk1, e == jk, in_dir, kind < 2, wsp,
                                                    bin, bout, rj, re, c.squared);
            acc += contrib;
            inwin = e >= jprev && e < jk;
            if (inwin) {            // a node of the window: its acceptance changes this term
                double xe1[D];
#pragma unroll
                for (int d = 0; d < D; ++d) xe1[d] = props[(size_t)e * PW + d];
                h = cc_term_delta<D>(xe1, xk0, xk1, false, in_dir, kind < 2, wsp, bin, bout, rj,
                                     re, c.squared) - contrib;
            }
        }
        const unsigned long long m = __ballot(inwin);
        if (inwin) {
            const int pos = cnt + __popcll(m & below);
            hidx[pos] = e - jprev;
            hval[pos] = h;
        }
        cnt += __popcll(m);
    }
    const double total = wave_sum_all(acc);
    if (lane == 0) {
        pb.tot[slot] = total;
        pb.hcnt[slot] = cnt;
    }
}

// Resolve batch b of slice t: thread k owns node k of the batch.
template <int D>
__device__ __forceinline__ void ccpipe_resolve(const ChainView &c, const CcPipeBuf &pb, int b, int t,
                                               unsigned long long (*sMask)[CP_WAVES],
                                               unsigned long long *sPrev, int *sChanged) {
    constexpr int PW = 2 * D + 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = c.N;
    const int j0 = b * CP_B;
    const int nb = min(CP_B, N - j0);
    const int ncross = j0 - max(0, j0 - CP_B);
    const int bb = b & 1;
    const int k = tid;
    const bool valid = k < nb;
    const int kc = min(k, nb - 1);
    unsigned long long *accg = pb.accmask + (size_t)t * CP_WAVES;
    if (tid < CP_WAVES) sPrev[tid] = b > 0 ? accg[tid] : 0ull;
    const size_t slot0 = (((size_t)bb * c.T + t) * CP_B + kc) * CP_SUBS;
    double r = 0.0;
    int cnts[CP_SUBS];
#pragma unroll
    for (int s = 0; s < CP_SUBS; ++s) { r += pb.tot[slot0 + s]; cnts[s] = pb.hcnt[slot0 + s]; }
    const double *pr = pb.prop + ((size_t)t * N + j0 + kc) * PW;
    double x0[D], x1[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { x1[d] = pr[d]; x0[d] = pr[D + 2 + d]; }
    // prior terms of the step's logp closure, with the neighbouring slices as they are now
    // (the odd slices run one batch behind: kernels_spec_pipe.hpp)
    r += node_log_prior<D>(c, t, j0 + kc, x1) - node_log_prior<D>(c, t, j0 + kc, x0);
    const double lu = log(pr[D]);
    const size_t tjc = (size_t)t * N + j0 + kc;
    double st = c.step[tjc];
    int32_t na = c.nacc[tjc], ns = c.nsteps[tjc], un = c.until[tjc];
    __syncthreads();                                   // sPrev visible
    // the previous batch's acceptances, final by now: cross entries in list order
    int nown = 0;
#pragma unroll
    for (int s = 0; s < CP_SUBS; ++s) {
        const double *hv = pb.hval + (slot0 + s) * pb.cap;
        const int32_t *hi = pb.hidx + (slot0 + s) * pb.cap;
        for (int e = 0; e < cnts[s]; ++e) {
            const int m = hi[e];
            if (m < ncross) {
                if ((sPrev[m >> 6] >> (m & 63)) & 1ull) r += hv[e];
            } else {
                ++nown;
            }
        }
    }
    {
        const unsigned long long g = __ballot(valid && !(lu >= r));
        if (lane == 0) sMask[0][wave] = g;
    }
    if (tid == 0) *sChanged = 0;
    __syncthreads();
    int cur = 0;
    for (int pass = 0; pass < CP_B + 2; ++pass) {
        double s_own = 0.0;
        if (nown > 0) {
#pragma unroll
            for (int s = 0; s < CP_SUBS; ++s) {
                const double *hv = pb.hval + (slot0 + s) * pb.cap;
                const int32_t *hi = pb.hidx + (slot0 + s) * pb.cap;
                for (int e = 0; e < cnts[s]; ++e) {
                    const int m = hi[e] - ncross;
                    if (m >= 0 && ((sMask[cur][m >> 6] >> (m & 63)) & 1ull)) s_own += hv[e];
                }
            }
        }
        const unsigned long long g = __ballot(valid && !(lu >= r + s_own));
        if (lane == 0) {
            sMask[cur ^ 1][wave] = g;
            if (g != sMask[cur][wave]) atomicOr(sChanged, 1);
        }
        __syncthreads();
        const int changed = *sChanged;
        cur ^= 1;
        __syncthreads();
        if (!changed) break;
        if (tid == 0) *sChanged = 0;
        __syncthreads();
    }
    const unsigned long long mine = sMask[cur][wave];
    const int accepted = (int)((mine >> lane) & 1ull);
    if (valid) {
        const size_t tj = (size_t)t * N + j0 + k;
        if (accepted) {
#pragma unroll
            for (int d = 0; d < D; ++d) c.X[tj * D + d] = x1[d];
        }
        metropolis_bookkeeping(st, na, ns, un, c.tune, c.tune_interval, accepted);
        c.step[tj] = st; c.nacc[tj] = na; c.nsteps[tj] = ns; c.until[tj] = un;
    }
    if (lane == 0) accg[wave] = mine;
}

// Launch l: even slices resolve batch l and evaluate batch l + 1; odd slices resolve batch
// l - 1 and evaluate batch l.  Workgroups [0, T) resolve, the rest evaluate.
template <int D>
__global__ __launch_bounds__(CP_THREADS) void k_ccpipe_step(ChainView c, CcPipeBuf pb, int l) {
    __shared__ unsigned long long sMask[2][CP_WAVES];
    __shared__ unsigned long long sPrev[CP_WAVES];
    __shared__ int sChanged;
    const int T = c.T;
    if ((int)blockIdx.x < T) {
        const int t = blockIdx.x;
        const int b = l - (t & 1);
        if (b >= 0 && b < pb.nbat) ccpipe_resolve<D>(c, pb, b, t, sMask, sPrev, &sChanged);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int nE = (T + 1) / 2, nO = T / 2;
    const int beE = l + 1, beO = l;
    const int nbE = (beE >= 0 && beE < pb.nbat) ? min(CP_B, c.N - beE * CP_B) : 0;
    const int nbO = (beO >= 0 && beO < pb.nbat) ? min(CP_B, c.N - beO * CP_B) : 0;
    const int nodesE = nE * nbE, nodes = nodesE + nO * nbO;
    const int nwaves = ((int)gridDim.x - T) * CP_WAVES;
    const int gw = __builtin_amdgcn_readfirstlane(
        ((int)blockIdx.x - T) * CP_WAVES + (int)(threadIdx.x >> 6));
    for (int item = gw; item < nodes * CP_SUBS; item += nwaves) {
        const int q = item / CP_SUBS, sub = item - q * CP_SUBS;
        const bool odd = q >= nodesE;
        const int qq = odd ? q - nodesE : q;
        const int nb = odd ? nbO : nbE;
        const int k = qq % nb;
        const int t = 2 * (qq / nb) + (odd ? 1 : 0);
        ccpipe_eval_item<D>(c, pb, odd ? beO : beE, t, k, sub, lane);
    }
}

}  // namespace dlsm
