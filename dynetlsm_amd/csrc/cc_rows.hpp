// Case-control model: a node's gathered terms as ONE row of int32 (round 5), shared by the sparse pipelined
// sweep (kernels_ccpipe.hpp) and the likelihood pass (kernels_loglik.hpp, k_loglik_casecontrol_rows).
#pragma once
#include "device_common.hpp"
#include "ccs_plan.hpp"

namespace dlsm {

// Row of node (t, i), `tw` int32 slots:
//   [0..3]  in_deg, out_deg, n_in_controls, n_out_controls
//   [4..7]  adj_in, adj_out as float64: (N - deg - 1) / n_controls (directed_likelihoods_fast.pyx:131,170 -
//           two float64 divisions per item otherwise, ~60 of the evaluator's ~800 vector instructions)
//   [8]     i: the node the row belongs to (round 6: the rows of a slice are stored SORTED, below); [9..11] spare
//   [12.. ] out-edges, out-controls, in-edges, in-controls back to back.  The OUT lists lead: the
//           likelihood pass (:208-270) walks exactly those, from a known offset - its indices leave with
//           the header, and out-edges + out-controls fill two 64-term trips where the three fixed slots of
//           the prefetch form (edges | controls 0-63 | controls 64-127) left 38 % of the lanes idle.
// The sweep's evaluator used to read the counts first and the four lists behind them (two round trips
// through a memory system that 2560 wavefronts of gathers keep busy); with the row it requests the counts
// and the first 256 indices at once, as coalesced 256-byte reads.  The rows change only when the edge
// tables or the controls do (upload / set / resample: every n_resample_control = 100 iterations);
// k_cc_rows rebuilds them then.
// Round 6: rows in the order the kernels want to MEET them.  On a network drawn from the model the nodes' term
// counts are skewed (out-degree 19 on average, 63 at the 99th percentile, 108 at most, at config 4) where the
// degree-regular network of rounds 1-5 had 20 for every node: a node beyond 256 terms is a second trip of the
// sweep's evaluator, beyond 128 out-terms a third trip of the pass, and both kernels dealt nodes to wavefronts by
// index - a SIMD with three evaluator items, two of them long, ended the launch (config 4: 2347 it/s against 2522
// on the degree-regular network).  The rows of every batch of CC_SORT_B nodes of a slice are now stored by
// DESCENDING term count (ties by index: k_cc_pos) and carry their node's index; the sweep deals a launch's items
// rank-major, so the longest items meet the wavefront slots that start first and the SIMDs with one item less,
// and the pass's two nodes per wavefront are neighbours in rank.  The order changes only with the rows.
constexpr int CC_SORT_B = 512;      // = CP_B (kernels_ccpipe.hpp)
constexpr int CP_HDR = 12;          // int32 slots of a row's header
__host__ __device__ constexpr int cp_terms_width(int cap) { return (CP_HDR + cap + 3) / 4 * 4; }
// pos[t N + i]: where node i's row lies among its slice's rows - inside its batch of CC_SORT_B, by descending
// number of terms (grid (batches, T), CC_SORT_B threads)
__global__ __launch_bounds__(CC_SORT_B) void k_cc_pos(ChainView c, const int32_t *nctrl, int32_t *pos) {
    __shared__ int key[CC_SORT_B];
    const int b = blockIdx.x, t = blockIdx.y, k = threadIdx.x;
    const int j0 = b * CC_SORT_B, nb = min(CC_SORT_B, c.N - j0);
    const long node = (long)t * c.N + j0 + min(k, nb - 1);
    const int mine = c.degree[node * 2] + c.degree[node * 2 + 1] + nctrl[node * 2] + nctrl[node * 2 + 1];
    key[k] = k < nb ? mine : -1;
    __syncthreads();
    if (k >= nb) return;
    int r = 0;
    for (int m = 0; m < nb; ++m) r += (key[m] > mine || (key[m] == mine && m < k)) ? 1 : 0;
    pos[node] = j0 + r;
}

__global__ __launch_bounds__(256) void k_cc_rows(ChainView c, const int32_t *nctrl, const int32_t *pos,
                                                 int32_t *terms, int tw) {
    const long node = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (node >= (long)c.T * c.N) return;
    const int in_deg = c.degree[node * 2], out_deg = c.degree[node * 2 + 1];
    const int nci = nctrl[node * 2], nco = nctrl[node * 2 + 1];
    const int t = (int)(node / c.N);
    int32_t *row = terms + ((long)t * c.N + pos[node]) * tw;
    if (lane < 4) row[lane] = lane == 0 ? in_deg : (lane == 1 ? out_deg : (lane == 2 ? nci : nco));
    if (lane >= 8 && lane < 12) row[lane] = lane == 8 ? (int)(node - (long)t * c.N) : 0;
    if (lane == 4) ((double *)row)[2] = (double)(c.N - in_deg - 1) / (double)nci;
    if (lane == 5) ((double *)row)[3] = (double)(c.N - out_deg - 1) / (double)nco;
    const int total = in_deg + out_deg + nci + nco;
    for (int q = lane; q < tw - CP_HDR; q += 64) {
        int r = q, e = 0;
        if (q < total) {
            if (r < out_deg) e = c.out_edges[node * c.Dout + r];
            else if ((r -= out_deg) < nco) e = c.ctrl_out[node * c.C + r];
            else if ((r -= nco) < in_deg) e = c.in_edges[node * c.Din + r];
            else e = c.ctrl_in[node * c.C + (r - in_deg)];
        }
        row[CP_HDR + q] = e;
    }
}

// The likelihood pass's walking order (round 6): a slice's rows by DESCENDING (out_deg, n_out_controls), ties by
// node, every row cut into ENTRIES of at most CC_ENT_TERMS = 128 out-terms (two 64-term trips): entry = place of
// the row in storage (pos) | segment << 24.
//  * Rows with equal out-degree and control count have the same control weight adj_out = (N - out_deg - 1) /
//    n_out_controls, and the pass keeps ONE running product of a wavefront's control factors while the weight
//    stays the same - a logarithm per run of rows instead of per row (a third of the four-candidate pass's vector
//    instructions).
//  * An entry is what one step of the pass's pipeline requests ahead, so a row of 200 out-terms is two steps like
//    any other two and a wavefront's share of the list - a contiguous run of entries - is the same work for every
//    wavefront (rows beyond two trips used to load their further terms in place, 2 - 3 us each, all of them in
//    the wavefronts that hold the top of the order).
// count[t] = entries of slice t.  Ranks and entry offsets by counting, N^2 comparisons per slice, only when the
// rows are rebuilt (grid (ceil(N / 256), T)); `order` holds T x N x emax int32.
__global__ __launch_bounds__(256) void k_cc_order(ChainView c, const int32_t *nctrl, const int32_t *pos, int emax,
                                                  int32_t *order, int32_t *count) {
    __shared__ int key[256];
    const int t = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x, N = c.N;
    const long base = (long)t * N;
    // key: (out_deg, n_out_controls); the entries of a row follow from it
    auto key_of = [&](int j) { return cc_order_key(c.degree[(base + j) * 2 + 1], nctrl[(base + j) * 2 + 1]); };
    auto ents_of = [&](int k) { return cc_order_entries(k); };
    const int mine = i < N ? key_of(i) : -1;
    int r = 0, start = 0;
    for (int j0 = 0; j0 < N; j0 += 256) {
        __syncthreads();
        key[threadIdx.x] = j0 + threadIdx.x < N ? key_of(j0 + threadIdx.x) : -2;
        __syncthreads();
        const int nj = min(256, N - j0);
        for (int m = 0; m < nj; ++m) {
            const bool before = key[m] > mine || (key[m] == mine && j0 + m < i);
            r += before ? 1 : 0;
            start += before ? ents_of(key[m]) : 0;
        }
    }
    if (i >= N) return;
    const int ne = ents_of(mine);
    int32_t *o = order + base * emax + start;
    for (int sgm = 0; sgm < ne; ++sgm) o[sgm] = pos[base + i] | (sgm << 24);
    if (r == N - 1) count[t] = start + ne;
}

}  // namespace dlsm
