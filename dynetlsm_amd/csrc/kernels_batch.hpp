// Several chains of the same network in ONE launch (undirected model, pipelined sweep).
//
// A chain alone leaves the chip to its launch floors: 18 sweep launches of 11 us, of which 4 us are
// the floor and the wait for the first operands, and T resolver workgroups hold a compute unit each
// for 7.5 of those 11 us.  Chains on streams of their own overlap some of that (bench.py
// --chains-per-gpu), but every chain still pays every floor, and a launch that wants every compute
// unit for its one-workgroup-per-CU grid only fills in behind the other chains' stragglers.
//
// Here the C chains of a batch share the launches of the pipelined sweep (k_pipe_step's roles,
// kernels_spec_pipe.hpp):
//   * workgroups [0, C T) resolve batch l of "their" (chain, slice) - and then JOIN the evaluators;
//   * every workgroup evaluates items of batch l + 1 drawn from ONE list over all chains
//     (chain-major, each chain's items in k_pipe_step's own order), dealt statically: round 0 goes
//     to the workgroups that have nothing to resolve, every later round to all of them;
//   * the chains' state is addressed through per-chain ChainView / PipeBuf copies in the kernel
//     arguments (constant address space: a wave-uniform index costs scalar loads only).
// An item is computed by exactly the code of the single-chain launch on exactly its operands, with
// the same split into parts: the batch is bit for bit the C single-chain runs
// (tests/test_gpu_batch.py).  The sweep's last launch (resolve + riding centring sums) and the
// iteration's last launch (centring, accept / reject, trace row, next proposal pass) take the chain
// from blockIdx.y.
// Reference counterpart: examples/homogeneous_simulation.py:177-184 refits seeds one after the other.
#pragma once
#include "kernels_spec_pipe.hpp"
#include "kernels_tail_propose.hpp"

namespace dlsm {

constexpr int BATCH_MAXC = 8;

struct PipeBatchArgs {
    int nc, pad_;
    ChainView c[BATCH_MAXC];
    PipeBuf pb[BATCH_MAXC];
};
static_assert(sizeof(PipeBatchArgs) <= 4000, "kernel arguments are limited to 4 KB");

template <int D, int MODEL_>
__global__ __launch_bounds__(PP_THREADS) void k_pipe_step_batch(PipeBatchArgs a, int l) {
    constexpr int MODEL = MODEL_ == PIPE_UNDIRECTED_LONG ? DLSM_UNDIRECTED : MODEL_;
    constexpr bool TP = MODEL_ != DLSM_UNDIRECTED;
    static_assert(MODEL == DLSM_UNDIRECTED, "the batch form covers the undirected model");
    extern __shared__ __attribute__((aligned(16))) double pp_sH[];      // 128 x 128
    __shared__ double sPart[PP_WAVES * 64];
    __shared__ unsigned long long sMask[2][2];
    __shared__ int sPrev[3 * PP_B];
    __shared__ int sOwn[PP_B + 1];
    __shared__ unsigned char sSat[PP_B];
    const int nc = a.nc;
    const int T = a.c[0].T, N = a.c[0].N;
    const int nres = nc * T;
    const int bx = (int)blockIdx.x;
    if (bx < nres) {
        const int ci = bx / T, t = bx - ci * T;
        const int b = l - (t & 1);
        if (b >= 0 && b < a.pb[ci].nbat) {
            __shared__ double sCross[PP_B];
            __shared__ unsigned long long sSatMask[2];
            __shared__ double sTab[EXPTAB_N];
            exp_table_fill(sTab, threadIdx.x);
            __syncthreads();
            row_resolve<D, false>(a.c[ci], a.pb[ci], b, t, pp_sH, sPart, sMask, nullptr, sCross, sSatMask, sTab
#ifdef DLSM_PIPE_TIMING
                                  , l + 1
#endif
                                  );
        }
        __syncthreads();                    // the block's LDS is free for the evaluators' table
    }
    const PipeBuf &pb0 = a.pb[0];
    const int lane = threadIdx.x & 63;
    const int nE = (T + 1) / 2, nO = T / 2;
    const int beE = l + 1, beO = l;                  // batch evaluated (even / odd slices)
    const int nbE = (beE >= 0 && beE < pb0.nbat) ? min(PP_B, N - beE * PP_B) : 0;
    const int nbO = (beO >= 0 && beO < pb0.nbat) ? min(PP_B, N - beO * PP_B) : 0;
    const int nslE = nbE > 0 ? nE : 0, nslO = nbO > 0 ? nO : 0, nsl = nslE + nslO;
    const int nitems1 = (pb0.parts * nsl) << 7;      // one chain's items, k_pipe_step's order
    if (nitems1 == 0) return;
    const int nitems = nc * nitems1;
    const float inv_nsl = 1.0f / (float)max(nsl, 1);
    const float inv_n1 = 1.0f / (float)nitems1;
    exp_table11_fill<PP_THREADS>(pp_sH, threadIdx.x);
    __syncthreads();
    // round 0: the workgroups with nothing to resolve; rounds >= 1: everybody
    const int wave = (int)(threadIdx.x >> 6);
    const int We = ((int)gridDim.x - nres) * PP_WAVES, W = (int)gridDim.x * PP_WAVES;
    const bool evalonly = bx >= nres;
    const int slot = __builtin_amdgcn_readfirstlane(evalonly ? (bx - nres) * PP_WAVES + wave
                                                             : We + bx * PP_WAVES + wave);
    for (int q = evalonly ? slot : We + slot; q < nitems; q = (q < We ? We + slot : q + W)) {
        int ci = (int)(((float)q + 0.5f) * inv_n1);              // q / nitems1 (q < 2^22)
        int q1 = q - ci * nitems1;
        if (q1 < 0) { --ci; q1 += nitems1; } else if (q1 >= nitems1) { ++ci; q1 -= nitems1; }
        const int k = q1 & (PP_B - 1);
        const int r = q1 >> 7;
        const int p = (int)(((float)r + 0.5f) * inv_nsl);        // r / nsl (r < 2^20)
        const int si = r - p * nsl;
        const bool odd = si >= nslE;
        const int be = odd ? beO : beE;
        const int nb = odd ? nbO : nbE;
        if (k >= nb) continue;
        const int t = odd ? 2 * (si - nslE) + 1 : 2 * si;
        PipeItemPre<D> pre;
        PipeHPre<D> hp;
        pipe_item_prologue<D, MODEL>(a.c[ci], a.pb[ci], be, t, k, p, lane, pre);
#if DLSM_H_FIRST
        pipe_h_prefetch<D, MODEL, 1>(a.c[ci], a.pb[ci], be, nb, t, k, p, lane, hp);
#endif
        pipe_eval_item<D, MODEL, TP, 1, false, false, DLSM_H_FIRST != 0, DLSM_H_FIRST != 0>(
            a.c[ci], a.pb[ci], be, nb, t, k, p, lane, pp_sH, nullptr, pre, hp
#ifdef DLSM_PIPE_TIMING
                                        , -1, 0
#endif
                                        );
    }
}

// the sweep's last launch (k_pipe_last_ride) for the chain blockIdx.y
struct RideBatchArgs {
    int nc, pad_;
    ChainView c[BATCH_MAXC];
    PipeBuf pb[BATCH_MAXC];
    PipePostRide pr[BATCH_MAXC];
};
static_assert(sizeof(RideBatchArgs) <= 4000, "kernel arguments are limited to 4 KB");
template <int D>
__global__ __launch_bounds__(PP_THREADS) void k_pipe_last_ride_batch(RideBatchArgs a, int l) {
    extern __shared__ __attribute__((aligned(16))) double pp_sH[];      // 128 x 128
    __shared__ double sPart[PP_WAVES * 64];
    __shared__ unsigned long long sMask[2][2];
    __shared__ int sPrev[3 * PP_B];
    __shared__ int sOwn[PP_B + 1];
    __shared__ unsigned char sSat[PP_B];
    const int ci = (int)blockIdx.y;
    pipe_last_ride_wg<D>(a.c[ci], a.pb[ci], l, a.pr[ci], (int)blockIdx.x, pp_sH, sPart, sMask, sPrev, sOwn, sSat);
}

// the iteration's last launch (k_lsm_finalize_apply_propose) for the chain blockIdx.y
struct FinBatchChain {
    const double *partials; int nrec, pad_;
    LsmDeviceState *lsm;
    double *intercept, *trace_ic, *trace_logp;
    ProposeBuf nb;
    PostFusedArgs pa;
};
struct FinBatchArgs {
    int nc, pad_;
    ChainView c[BATCH_MAXC];
    FinBatchChain f[BATCH_MAXC];
};
static_assert(sizeof(FinBatchArgs) <= 4000, "kernel arguments are limited to 4 KB");
template <int D>
__global__ __launch_bounds__(256) void k_lsm_finalize_apply_propose_batch(FinBatchArgs a, IterRef ir) {
    const int ci = (int)blockIdx.y;
    const FinBatchChain &f = a.f[ci];
    lsm_finalize_apply_propose_wg<D>(f.partials, f.nrec, f.lsm, f.intercept, f.trace_ic, f.trace_logp, ir,
                                     a.c[ci], f.nb, f.pa, (int)blockIdx.x, (int)gridDim.x - 1);
}

}  // namespace dlsm
