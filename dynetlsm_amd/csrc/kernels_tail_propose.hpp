// The last launch of an iteration of the device-resident loops is one workgroup (the intercept's
// accept / reject and trace row of the LSM; the HDP's hyper-parameters, 14-16 us of dependent
// draws) with the rest of the chip idle, and the first launch of the next iteration (the sweep's
// proposal pass, 5 us) needs nothing that workgroup produces except the intercept for its two
// constants.  Here the proposal pass of iteration it + 1 rides in iteration it's last launch as
// workgroups 1 ..; workgroup 0 writes the two constants.  Same draws (the Philox counters are the
// node's, the slice's and the iteration's), same values in the proposal buffer; the sweep then
// starts with its first batch.
#pragma once
#include "kernels_hdploop.hpp"
#include "kernels_spec_pipe.hpp"
#include "kernels_sweep.hpp"

namespace dlsm {

__host__ __device__ inline int propose_blocks(int T, int N) { return ((N + 255) / 256) * T; }

template <int D>
__global__ __launch_bounds__(256) void k_lsm_finalize_propose(
    const double *__restrict__ partials, int nrec, LsmDeviceState *lsm,
    double *__restrict__ intercept, double *__restrict__ trace_ic,
    double *__restrict__ trace_logp, IterRef ir, ChainView c, ProposeBuf nb) {
    if (blockIdx.x == 0) {
        lsm_finalize_wg(partials, nrec, lsm, intercept, trace_ic, trace_logp, ir);
        if (threadIdx.x == 0) pipe_propose_consts(c, nb.consts, intercept);     // its own store above
        return;
    }
    pipe_propose_rows<D>(c, nb, ir.get() + 1u, (int)blockIdx.x - 1, (int)threadIdx.x);
}

static_assert(HH_THREADS == 256, "the proposal pass is laid out for 256 threads");
template <int D>
__global__ __launch_bounds__(HH_THREADS) void k_hdp_hypers_propose(ChainView c, HdpLoopBuf hb,
                                                                   HdpDeviceState *hs, HdpTrace tr,
                                                                   IterRef ir, ProposeBuf nb) {
    if (blockIdx.x == 0) {
        // the intercept of the next sweep was settled in stage 1
        if (threadIdx.x == 0) pipe_propose_consts(c, nb.consts, c.intercept);
        hdp_hypers_wg(c, hb, hs, tr, ir);
        return;
    }
    pipe_propose_rows<D>(c, nb, ir.get() + 1u, (int)blockIdx.x - 1, (int)threadIdx.x);
}

}  // namespace dlsm
