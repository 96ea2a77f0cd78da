// GPU-free check of the case-control likelihood pass's plan arithmetic (dynetlsm_amd/csrc/ccs_plan.hpp): the walking
// order's key and entries, the wavefronts' shares of a slice's entries, the workgroups per slice.  Test infrastructure
// (tests/test_ccs_plan_cpu.py builds and runs it with g++ under ASan / UBSan).
#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../dynetlsm_amd/csrc/ccs_plan.hpp"

using namespace dlsm;

static int fail(const char *what, long a, long b, long c) { printf("%s: %ld %ld %ld\n", what, a, b, c); return 1; }

int main() {
    long cases = 0;
    // 1. the key: one int32 without overflow inside its bounds, ordered as (out_deg, n_out_controls)
    if (!cc_order_key_holds(32767, 65535) || cc_order_key_holds(32768, 0) || cc_order_key_holds(0, 65536))
        return fail("key bounds", 0, 0, 0);
    if (cc_order_key(32767, 65535) != INT_MAX) return fail("largest key", cc_order_key(32767, 65535), 0, 0);
    for (int d = 0; d < 400; d += 7)
        for (int c = 0; c < 400; c += 11) {
            if (!(cc_order_key(d + 1, 0) > cc_order_key(d, c))) return fail("key order (out_deg)", d, c, 0);
            if (!(cc_order_key(d, c + 1) > cc_order_key(d, c))) return fail("key order (controls)", d, c, 0);
            const int nt = d + c, ne = cc_order_entries(cc_order_key(d, c));
            if (ne < 1 || ne * CC_ENT_TERMS < nt || (nt > 0 && (ne - 1) * CC_ENT_TERMS >= nt)) return fail("entries", d, c, ne);
            if (ne > cc_order_entries_max(d, c)) return fail("entries beyond the buffer's width", d, c, ne);
            ++cases;
        }
    // 2. the shares: contiguous, in order, covering [0, E), equal to one entry - also where g E needs 64 bits
    const int Es[] = {0, 1, 2, 63, 64, 65, 4999, 11523, 50000, 1 << 24, 1500000000};
    const int Gs[] = {1, 2, 3, 4, 16, 196, 784, 816, 1000, 4096, 8192};
    for (int E : Es)
        for (int G : Gs) {
            int next = 0, lo = INT_MAX, hi = 0;
            for (int g = 0; g < G; ++g) {
                int e0, n;
                ccs_share(g, G, E, e0, n);
                if (e0 != next || n < 0) return fail("share not contiguous", E, G, g);
                next = e0 + n; lo = std::min(lo, n); hi = std::max(hi, n);
            }
            if (next != E || hi - lo > 1) return fail("shares do not cover / unequal", E, G, hi - lo);
            ++cases;
        }
    // 3. the grid: resident (at most n_cu bpc / T workgroups per slice), inside the records' room, whole rounds
    for (int N = 1; N <= 40000; N += (N < 300 ? 1 : 97))
        for (int T = 1; T <= 12; ++T)
            for (int bpc = 1; bpc <= 6; ++bpc)
                for (int nwv : {4, 16}) {
                    const int cap = (N + 15) / 16, n_cu = 256;
                    const int wps = ccs_workgroups_per_slice(N, T, n_cu, bpc, nwv, cap);
                    if (wps < 1 || wps > cap) return fail("grid outside the records' room", N, T, wps);
                    if (wps > std::max(1, n_cu * bpc / T)) return fail("grid not resident", N, T, wps);
                    // (N entries: every wavefront of the slice takes the same number of rounds, to one)
                    const int G = wps * nwv;
                    if (wps < cap && wps < std::max(1, n_cu * bpc / T)) {
                        const int r_hi = (N + G - 1) / G, r_full = (N + std::max(1, n_cu * bpc / T) * nwv - 1) / (std::max(1, n_cu * bpc / T) * nwv);
                        if (r_hi != r_full) return fail("trimmed grid takes more rounds than the full one", N, T, wps);
                    }
                    ++cases;
                }
    // 4. the order as k_cc_order builds it (ranks and entry offsets by counting), on random rows: every row once per
    //    segment, segments in order, rows by descending key with ties by index, the count = the entries written
    srand(12345);
    for (int rep = 0; rep < 40; ++rep) {
        const int N = 1 + rand() % 700, C = rand() % 150;
        std::vector<int> key(N);
        for (int i = 0; i < N; ++i) key[i] = cc_order_key(rand() % (rep % 2 ? 300 : 40), std::min(C, N - 1));
        const int emax = cc_order_entries_max(299, C);
        std::vector<int> order((size_t)N * emax, -1);
        int count = -1;
        for (int i = 0; i < N; ++i) {
            int r = 0, start = 0;
            for (int j = 0; j < N; ++j) {
                const bool before = key[j] > key[i] || (key[j] == key[i] && j < i);
                r += before ? 1 : 0;
                start += before ? cc_order_entries(key[j]) : 0;
            }
            const int ne = cc_order_entries(key[i]);
            if ((size_t)start + ne > order.size()) return fail("entries beyond the buffer", N, i, start);
            for (int s = 0; s < ne; ++s) order[start + s] = i | (s << 24);
            if (r == N - 1) count = start + ne;
        }
        int prev = INT_MAX, prev_i = -1, k = 0;
        while (k < count) {
            const int i = order[k] & 0xFFFFFF;
            if (order[k] < 0 || (order[k] >> 24) != 0) return fail("entry list: a row's first segment expected", N, k, order[k]);
            if (key[i] > prev || (key[i] == prev && i < prev_i)) return fail("order not descending", N, k, i);
            const int ne = cc_order_entries(key[i]);
            for (int s = 0; s < ne; ++s)
                if (order[k + s] != (i | (s << 24))) return fail("segments", N, k, s);
            prev = key[i]; prev_i = i; k += ne;
        }
        if (k != count) return fail("count", N, k, count);
        ++cases;
    }
    printf("check_ccs_plan ok (%ld cases)\n", cases);
    return 0;
}
