// Control-node resampling for the case-control likelihood (a7) and the
// chip-wide speculative-batch sweep (algo 2).
#pragma once
#include "chain.hpp"
#include "device_common.hpp"
#include "kernels_sweep.hpp"

namespace dlsm {

// ---------------------------------------------------------------------------
// DirectedCaseControlSampler.sample (case_control_likelihood.py:75-112):
// for every (t, i) and each direction draw n = min(C, #zeros) distinct nodes
// that are neither i nor a neighbour in that direction; pad with -1.
// One wave per (t, i); a per-wave LDS bitmap of N bits marks excluded nodes;
// lanes draw candidates with Philox (stream CONTROLS) and are admitted in lane
// order, which makes the result a deterministic function of (seed, iter, t, i).
// A uniformly random k-subset is what rng.choice(..., replace=False) returns.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resample_controls(
    ChainView c, int32_t *__restrict__ ctrl_in, int32_t *__restrict__ ctrl_out,
    uint32_t iter) {
    extern __shared__ __attribute__((aligned(16))) uint32_t sbits[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NW = (c.N + 31) / 32;
    uint32_t *bm = sbits + (size_t)wave * NW;
    const long node = (long)blockIdx.x * (blockDim.x / 64) + wave;
    if (node >= (long)c.T * c.N) return;
    const int t = (int)(node / c.N), i = (int)(node % c.N);
    for (int dir = 0; dir < 2; ++dir) {           // 0 = out, 1 = in (reference order)
        const int deg = c.degree[node * 2 + (dir == 0 ? 1 : 0)];
        const int32_t *edges = dir == 0 ? c.out_edges + node * c.Dout
                                        : c.in_edges + node * c.Din;
        int32_t *dst = (dir == 0 ? ctrl_out : ctrl_in) + node * c.C;
        for (int w = lane; w < NW; w += 64) bm[w] = 0u;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) atomicOr(&bm[i >> 5], 1u << (i & 31));
        for (int k = lane; k < deg; k += 64) {
            const int e = edges[k];
            atomicOr(&bm[e >> 5], 1u << (e & 31));
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        const int n_zeros = c.N - deg - 1;          // as the reference counts them
        const int n_sample = n_zeros < c.C ? n_zeros : c.C;
        int filled = 0;
        if (n_sample == n_zeros) {
            // every admissible node is a control: enumerate them
            for (int base = 0; base < c.N && filled < n_sample; base += 64) {
                const int v = base + lane;
                const bool ok = v < c.N && !((bm[v >> 5] >> (v & 31)) & 1u);
                const unsigned long long m = __ballot(ok);
                const int pos = filled + __popcll(m & ((1ull << lane) - 1ull));
                if (ok && pos < n_sample) dst[pos] = v;
                filled += __popcll(m);
            }
            if (filled > n_sample) filled = n_sample;
        } else {
            for (uint32_t round = 0; filled < n_sample && round < 32768u; ++round) {
                U4 r = philox4x32_10(c.seed, (uint32_t)i * 64u + (uint32_t)lane,
                                     (uint32_t)t | ((uint32_t)dir << 16) | (round << 17),
                                     iter, stream_word(c.chain, STREAM_CONTROLS));
                const int cand = (int)__umulhi(r.x, (uint32_t)c.N);
                for (int l = 0; l < 64 && filled < n_sample; ++l) {
                    const int v = __shfl(cand, l, 64);
                    const uint32_t word = bm[v >> 5];
                    if (!((word >> (v & 31)) & 1u)) {
                        if (lane == 0) {
                            bm[v >> 5] = word | (1u << (v & 31));
                            dst[filled] = v;
                        }
                        ++filled;
                    }
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        for (int k = filled + lane; k < c.C; k += 64) dst[k] = -1;
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace dlsm
