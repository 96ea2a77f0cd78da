// GPU-free check of the pipelined sweep's part plan (dynetlsm_amd/csrc/pipe_plan.hpp): for every network size,
// number of parts, batch and workgroup class the parts' trip lists partition the slice's trips - every trip of 64
// neighbours in exactly one part, the window's trips as the FIRST trips of their parts (where the evaluator finishes
// the H factors and masks the node itself), nothing beyond the LDS capacity the host sizes.  Test infrastructure
// (tests/test_pipe_plan_cpu.py builds and runs it, under ASan / UBSan).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../dynetlsm_amd/csrc/pipe_plan.hpp"

using namespace dlsm;

int main() {
    long cases = 0;
    for (int N = 2; N <= 13000; N += (N < 700 ? 1 : 37)) {
        const int ntrip = (N + 63) / 64, nbat = (N + 127) / 128;
        for (int P = 1; P <= 8; ++P) {
            uint32_t plan[4][8];
            for (int nw = 1; nw <= 4; ++nw)
                for (int p = 0; p < 8; ++p) plan[nw - 1][p] = p < P ? pipe_plan_entry(ntrip, P, nw, p) : 0u;
            const int cap = pipe_lds_trip_cap(ntrip, P);
            for (int be = 0; be < nbat; ++be) {
                const int nb = N - be * 128 < 128 ? N - be * 128 : 128;
                for (int k0 = 0; k0 < nb; k0 += 16) {
                    std::vector<int> seen(ntrip, 0), win(ntrip, 0);
                    int glo = -1, nwin = -1;
                    for (int p = 0; p < P; ++p) {
                        const int ghi0 = 2 * be + (k0 >= 64 ? 1 : 0);
                        const int nw = (ghi0 < ntrip - 1 ? ghi0 : ntrip - 1) - (be > 0 ? 2 * be - 2 : 0) + 1;
                        if (nw < 1 || nw > 4) { printf("nwin %d out of range: N %d be %d k0 %d\n", nw, N, be, k0); return 1; }
                        const PipePlan pl = pipe_plan_from_entry(plan[nw - 1][p], ntrip, be, k0);
                        if (pl.nwin != nw) { printf("nwin mismatch\n"); return 1; }
                        glo = pl.glo; nwin = pl.nwin;
                        if (pl.w < 0 || pl.w > PL_WIN || pl.r < 0 || pl.trips() > cap) {
                            printf("capacity: N %d P %d p %d be %d k0 %d: w %d r %d cap %d\n", N, P, p, be, k0, pl.w, pl.r, cap);
                            return 1;
                        }
                        for (int u = 0; u < pl.trips(); ++u) {
                            const int g = pl.trip(u, p, P);
                            if (g < 0 || g >= ntrip) { printf("trip %d outside [0, %d): N %d P %d p %d be %d k0 %d u %d\n", g, ntrip, N, P, p, be, k0, u); return 1; }
                            ++seen[g];
                            if (u < pl.w) ++win[g];
                        }
                    }
                    for (int g = 0; g < ntrip; ++g) {
                        const bool inwin = g >= glo && g < glo + nwin;
                        if (seen[g] != 1 || win[g] != (inwin ? 1 : 0)) {
                            printf("partition: N %d P %d be %d k0 %d trip %d seen %d as window %d (window [%d, %d))\n",
                                   N, P, be, k0, g, seen[g], win[g], glo, glo + nwin);
                            return 1;
                        }
                    }
                    // the node itself (trip 2 be or 2 be + 1) and every node of the window lie in window trips
                    const int gself_lo = (be * 128 + k0) / 64, gself_hi = (be * 128 + (k0 + 15 < nb ? k0 + 15 : nb - 1)) / 64;
                    if (gself_lo < glo || gself_hi >= glo + nwin) { printf("self outside the window: N %d be %d k0 %d\n", N, be, k0); return 1; }
                    ++cases;
                }
            }
        }
    }
    printf("check_pipe_plan ok: %ld (network, parts, batch, workgroup) cases\n", cases);
    return 0;
}
