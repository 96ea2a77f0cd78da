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
        if (threadIdx.x == 0) {                 // behind its own stores above
            pipe_propose_consts(c, nb.consts, intercept);
            if (nb.lsm_draw) pipe_propose_intercept(c, nb.lsm_draw, intercept, ir.get() + 1u);
        }
        return;
    }
    pipe_propose_rows<D>(c, nb, ir.get() + 1u, (int)blockIdx.x - 1, (int)threadIdx.x);
}

// The undirected LSM loop's last TWO launches in one.  Distances do not change under the rotation
// and the shift of the centring pass, so the likelihood pass can read the positions as the sweep
// left them (its intercept proposal was drawn with the sweep's proposals), and what is left - centre
// and rotate the positions, the intercept's accept / reject, the trace row, the next sweep's
// proposal pass - needs one launch: every workgroup redoes the centring pass's prologue (the riding
// sums' records + the rows they left out -> R, shift); workgroup 0 takes the latent prior terms from
// it and finishes the iteration; workgroup 1 + w centres its rows (a node per thread), files them in
// X and in the trace, and draws the node's proposal for the next sweep from the centred row.
struct PostFusedArgs {
    int has_ref, n_iter_procrustes;
    const double *rec; int nrec, jl, par;
    const double *xref_rows;
    double *trace_X;
};
static_assert(PS2_THREADS == 256, "the proposal pass is laid out for 256 threads");
// workgroup `bx` of 1 + nblk (nblk = the proposal pass's workgroups)
template <int D>
__device__ __forceinline__ void lsm_finalize_apply_propose_wg(
    const double *__restrict__ partials, int nrec, LsmDeviceState *lsm,
    double *__restrict__ intercept, double *__restrict__ trace_ic,
    double *__restrict__ trace_logp, IterRef ir, const ChainView &c, const ProposeBuf &nb,
    const PostFusedArgs &pa, int bx, int nblk) {
    if (bx == 0) {
        // sums, R, shift and the latent prior terms (lsm->prior_x); no rows, no intercept draw
        post_apply_wg<D>(c, pa.has_ref, pa.n_iter_procrustes, 1, pa.rec, pa.nrec, lsm, ir, nullptr, nullptr,
                         nullptr, 0, 0, 1, pa.jl, pa.par, pa.xref_rows, false, 0);
        lsm_finalize_wg(partials, nrec, lsm, intercept, trace_ic, trace_logp, ir);
        if (threadIdx.x == 0) {
            pipe_propose_consts(c, nb.consts, intercept);
            if (nb.lsm_draw) pipe_propose_intercept(c, nb.lsm_draw, intercept, ir.get() + 1u);
        }
        return;
    }
    const uint32_t next = ir.get() + 1u;
    auto hook = [&](long r, const double *y) {
        const int t = (int)(r / c.N), j = (int)(r - (long)t * c.N);
        double x0[D];
#pragma unroll
        for (int d = 0; d < D; ++d) x0[d] = y[d];
        pipe_propose_row_from<D>(c, nb, next, t, j, x0);
    };
    post_apply_wg<D>(c, pa.has_ref, pa.n_iter_procrustes, 1, pa.rec, pa.nrec, nullptr, ir, nullptr, pa.trace_X,
                     nullptr, 0, bx - 1, nblk, pa.jl, pa.par, pa.xref_rows, true, 0, hook);
}

template <int D>
__global__ __launch_bounds__(256) void k_lsm_finalize_apply_propose(
    const double *__restrict__ partials, int nrec, LsmDeviceState *lsm,
    double *__restrict__ intercept, double *__restrict__ trace_ic,
    double *__restrict__ trace_logp, IterRef ir, ChainView c, ProposeBuf nb, PostFusedArgs pa) {
    lsm_finalize_apply_propose_wg<D>(partials, nrec, lsm, intercept, trace_ic, trace_logp, ir, c, nb, pa,
                                     (int)blockIdx.x, (int)gridDim.x - 1);
}

// HDP-LPCM loop on two queues with the next sweep's head on the second one: the intercept step behind its
// pass (k_hdp_intercept_fork's workgroup, which also leaves the sweep's two constants) and the sweep's
// proposal pass in ONE launch of that queue - the rows' proposals need positions and step sizes only
template <int D>
__global__ __launch_bounds__(256) void k_hdp_intercept_fork_propose(
    const double *__restrict__ partials, int nrec, LsmDeviceState *lsm, HdpDeviceState *hs,
    double *__restrict__ intercept, double *__restrict__ trace_ic, int it, ChainView c, ProposeBuf nb) {
    if (blockIdx.x == 0) {
        hdp_intercept_wg(partials, nrec, lsm, hs, intercept, trace_ic, it);
        if (threadIdx.x == 0) pipe_propose_consts(c, nb.consts, intercept);     // behind its own stores
        return;
    }
    pipe_propose_rows<D>(c, nb, (uint32_t)it + 1u, (int)blockIdx.x - 1, (int)threadIdx.x);
}

static_assert(HH_THREADS == 256, "the proposal pass is laid out for 256 threads");
template <int D>
__global__ __launch_bounds__(HH_THREADS) void k_hdp_hypers_propose(ChainView c, HdpLoopBuf hb,
                                                                   HdpDeviceState *hs, HdpTrace tr,
                                                                   IterRef ir, ProposeBuf nb, HdpFork fk) {
    __builtin_amdgcn_s_setprio(3);
    if (blockIdx.x == 0) {
        // the intercept of the next sweep was settled in stage 1 - or, with the likelihood pass on a
        // queue of its own (HdpFork), by k_hdp_intercept_fork: then a lane of the wavefronts that draw
        // nothing waits for its flag
        if (fk.flags) {
            if (threadIdx.x == HH_THREADS - 1) {
                hdp_fork_acquire_settled(fk);
                // (consts == NULL: the sweep's head - proposal pass, constants, first launch - was enqueued on
                // the second queue and the flag says it has ended)
                if (nb.consts) pipe_propose_consts(c, nb.consts, c.intercept);
            }
        } else if (threadIdx.x == 0) pipe_propose_consts(c, nb.consts, c.intercept);   // (nb.lsm_draw is NULL here)
        hdp_hypers_wg(c, hb, hs, tr, ir);
        return;
    }
    pipe_propose_rows<D>(c, nb, ir.get() + 1u, (int)blockIdx.x - 1, (int)threadIdx.x);
}

}  // namespace dlsm
