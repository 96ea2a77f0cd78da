// The parts of the pipelined sweep's LDS evaluators (kernels_pipe_lds.hpp) as trip lists: plain C++ shared by the
// host (capi.hip fills PipeLds::plan with pipe_plan_entry), the device (PipePlan::trip) and a GPU-free test of the
// partition (tests/test_pipe_plan_cpu.py compiles this header with g++).
#pragma once
#include <stdint.h>
#include <stddef.h>
#ifdef __HIPCC__
#define DLSM_PLAN_HD __host__ __device__
#else
#define DLSM_PLAN_HD
#endif

namespace dlsm {

constexpr int PL_EXPTAB_DOUBLES = 2048;     // = EXPTAB11_N (device_common.hpp): the evaluators' exp table in front of the rows

constexpr int PL_WIN = 4;           // window trips a part can hold (P = 1: all four)

// trips (and LDS rows) the longest part can hold
DLSM_PLAN_HD inline int pipe_lds_trip_cap(int ntrip, int parts) { return (ntrip + parts - 1) / parts + PL_WIN; }
// bytes of dynamic LDS the evaluators need: exp table, the longest part's rows, its window trips' proposals
DLSM_PLAN_HD inline size_t pipe_lds_eval_bytes(int N, int D, int parts) {
    const int ntrip = (N + 63) / 64;
    return ((size_t)PL_EXPTAB_DOUBLES + (size_t)(pipe_lds_trip_cap(ntrip, parts) + PL_WIN) * 64 * D) * sizeof(double);
}

// The trips of a node of batch `be` (its workgroup's first node k0) as P parts.  Window trips: [glo, glo + nwin)
// = the previous batch's two (be > 0), trip 2 be of the own batch, and trip 2 be + 1 when the workgroup's nodes
// reach into it (k0 >= 64); window trip i goes to part i mod P and is the part's trip i / P.  The other ntrip -
// nwin trips, in ascending order, are cut into runs r_0 .. r_{P-1} with r_j + 2 w_j as equal as integers allow.
// (w, r, s) depend on (nwin, p) only: the host computes the 4 P triples once (pipe_plan_entry) and the launch
// carries them as kernel arguments (PipeBuf::plan) - the evaluators' row requests wait for nothing but a decode.
struct PipePlan {
    int glo, nwin;      // the window's trips
    int w, r, s;        // this part: window trips, other trips, rank of its first other trip
    DLSM_PLAN_HD inline int trips() const { return w + r; }
    DLSM_PLAN_HD inline int trip(int u, int p, int P) const {      // the part's u-th trip
        if (u < w) return glo + p + P * u;
        const int rho = s + (u - w);
        return rho < glo ? rho : rho + nwin;
    }
};
// w | r << 3 | s << 16 of part p when the window holds nwin trips (1 .. 4) of ntrip
inline uint32_t pipe_plan_entry(int ntrip, int P, int nwin, int p) {
    auto wof = [&](int j) { return j < nwin ? (nwin - j + P - 1) / P : 0; };
    const int R = ntrip - nwin, S = R + 2 * nwin;
    const int q = S / P, rem = S % P;
    const bool balanced = R >= 0 && q >= 2 * wof(0);             // (tiny N: plain runs of the other trips)
    const int Rp = R > 0 ? R : 0;
    int s = 0, w = 0, r = 0;
    for (int j = 0; j <= p; ++j) {
        const int wj = wof(j);
        const int rj = balanced ? q + (j < rem ? 1 : 0) - 2 * wj : Rp / P + (j < Rp % P ? 1 : 0);
        if (j < p) s += rj; else { w = wj; r = rj; }
    }
    return (uint32_t)w | ((uint32_t)r << 3) | ((uint32_t)s << 16);
}
// the window of a workgroup whose first node is k0 of batch be, and its part p's (w, r, s) from a plan entry
DLSM_PLAN_HD inline PipePlan pipe_plan_from_entry(uint32_t e, int ntrip, int be, int k0) {
    PipePlan pl;
    pl.glo = be > 0 ? 2 * be - 2 : 0;
    const int ghi0 = 2 * be + (k0 >= 64 ? 1 : 0);
    const int ghi = ghi0 < ntrip - 1 ? ghi0 : ntrip - 1;
    pl.nwin = ghi - pl.glo + 1;                                  // 1 .. 4 (trip 2 be exists: the batch has nodes)
    pl.w = (int)(e & 7u); pl.r = (int)((e >> 3) & 0x1fffu); pl.s = (int)(e >> 16);
    return pl;
}

}  // namespace dlsm
