// The arithmetic of the case-control likelihood pass's walking order and of its launch (cc_rows.hpp: k_cc_order,
// kernels_loglik_ccstream.hpp, capi.hip: launch_ccs) as plain C++ shared by the host, the device and a GPU-free check
// (tests/test_ccs_plan_cpu.py compiles this header with g++ under ASan / UBSan): the order's key, the entries a row is
// cut into, the wavefronts' shares of a slice's entries, the workgroups per slice.
#pragma once
#include <stdint.h>
#ifdef __HIPCC__
#define DLSM_CCS_HD __host__ __device__
#else
#define DLSM_CCS_HD
#endif

namespace dlsm {

constexpr int CC_ENT_TERMS = 128;       // out-terms of an entry: two 64-term trips, what one pipeline step requests ahead

// rows are ranked by DESCENDING (out_deg, n_out_controls) - equal keys = equal control weight adj_out; one int32
DLSM_CCS_HD inline int cc_order_key(int out_deg, int n_out_controls) { return out_deg * 65536 + n_out_controls; }
// ... which holds edge lists below 32768 out-edges and 65536 controls (beyond: the rows form of the pass)
DLSM_CCS_HD inline bool cc_order_key_holds(int max_out_deg, int n_control) { return max_out_deg < 32768 && n_control < 65536; }
// entries of a row: ceil(out-terms / CC_ENT_TERMS), at least one (a row without out-terms is an empty step)
DLSM_CCS_HD inline int cc_order_entries(int key) {
    const int nt = (key >> 16) + (key & 65535);
    const int n = (nt + CC_ENT_TERMS - 1) / CC_ENT_TERMS;
    return n > 1 ? n : 1;
}
// entries a slice can have at most per row (the order's buffer: T x N x this)
DLSM_CCS_HD inline int cc_order_entries_max(int max_out_deg, int n_control) {
    const int n = (max_out_deg + n_control + CC_ENT_TERMS - 1) / CC_ENT_TERMS;
    return n > 1 ? n : 1;
}

// wavefront g of G takes the entries [e0, e0 + n) of a slice's E: contiguous, equal to one entry, covering [0, E)
DLSM_CCS_HD inline void ccs_share(int g, int G, int E, int &e0, int &n) {
    e0 = (int)((long long)g * E / G);
    n = (int)((long long)(g + 1) * E / G) - e0;
}

// workgroups per slice of a launch that stays resident (bpc workgroups of nwv wavefronts per CU, T slices), trimmed so
// that the ~N entries of a slice are whole rounds of the slice's wavefronts; at most `cap` (the records' room)
DLSM_CCS_HD inline int ccs_workgroups_per_slice(int N, int T, int n_cu, int bpc, int nwv, int cap) {
    int wps0 = n_cu * bpc / T;
    if (wps0 < 1) wps0 = 1;
    const int rounds = (N + wps0 * nwv - 1) / (wps0 * nwv);
    const int gb = (N + rounds - 1) / rounds;
    int wps = (gb + nwv - 1) / nwv;
    if (wps < 1) wps = 1;
    return wps < cap ? wps : (cap > 1 ? cap : 1);
}

}  // namespace dlsm
