// Case-control model: a node's gathered terms as ONE row of int32 (round 5), shared by the sparse pipelined
// sweep (kernels_ccpipe.hpp) and the likelihood pass (kernels_loglik.hpp, k_loglik_casecontrol_rows).
#pragma once
#include "device_common.hpp"

namespace dlsm {

// Row of node (t, i), `tw` int32 slots:
//   [0..3]  in_deg, out_deg, n_in_controls, n_out_controls
//   [4..7]  adj_in, adj_out as float64: (N - deg - 1) / n_controls (directed_likelihoods_fast.pyx:131,170 -
//           two float64 divisions per item otherwise, ~60 of the evaluator's ~800 vector instructions)
//   [8.. ]  out-edges, out-controls, in-edges, in-controls back to back.  The OUT lists lead: the
//           likelihood pass (:208-270) walks exactly those, from a known offset - its indices leave with
//           the header, and out-edges + out-controls fill two 64-term trips where the three fixed slots of
//           the prefetch form (edges | controls 0-63 | controls 64-127) left 38 % of the lanes idle.
// The sweep's evaluator used to read the counts first and the four lists behind them (two round trips
// through a memory system that 2560 wavefronts of gathers keep busy); with the row it requests the counts
// and the first 256 indices at once, as coalesced 256-byte reads.  The rows change only when the edge
// tables or the controls do (upload / set / resample: every n_resample_control = 100 iterations);
// k_cc_rows rebuilds them then.
constexpr int CP_HDR = 8;           // int32 slots of a row's header
__host__ __device__ constexpr int cp_terms_width(int cap) { return (CP_HDR + cap + 3) / 4 * 4; }
__global__ __launch_bounds__(256) void k_cc_rows(ChainView c, const int32_t *nctrl, int32_t *terms, int tw) {
    const long node = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (node >= (long)c.T * c.N) return;
    const int in_deg = c.degree[node * 2], out_deg = c.degree[node * 2 + 1];
    const int nci = nctrl[node * 2], nco = nctrl[node * 2 + 1];
    int32_t *row = terms + node * tw;
    if (lane < 4) row[lane] = lane == 0 ? in_deg : (lane == 1 ? out_deg : (lane == 2 ? nci : nco));
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

}  // namespace dlsm
