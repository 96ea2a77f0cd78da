// CPU sanitizer driver (test infrastructure) for the product's host-side C++ that needs no
// GPU: the native table draws on numpy's bit-generator interface (csrc/host_draws.hpp) and
// the tridiagonal eigen-solver of the Lanczos step (csrc/host_tridiag.hpp).  Built with
// -fsanitize=address,undefined by tests/test_sanitizers_cpu.py.  Exit code 0 = no report.
#include <stdint.h>
#include <stdio.h>

#include <cmath>
#include <vector>

#include "../../dynetlsm_amd/csrc/host_draws.hpp"
#include "../../dynetlsm_amd/csrc/host_tridiag.hpp"

static uint64_t g_state = 12345;
static uint64_t next64(void *) {
    g_state = g_state * 6364136223846793005ull + 1442695040888963407ull;
    return g_state;
}
static uint32_t next32(void *) { return (uint32_t)(next64(nullptr) >> 32); }
static double nextd(void *) { return (double)(next64(nullptr) >> 11) / 9007199254740992.0; }

int main() {
    dlsm::NumpyBitGen g{nullptr, next64, next32, nextd, next64};
    // table draws: shapes incl. empty cells, an underflowed dish weight, kappa = 0
    for (int trial = 0; trial < 20; ++trial) {
        const int T = 1 + trial % 4, K = 1 + trial % 7;
        std::vector<double> n((size_t)T * K * K), beta(K);
        double tot = 0.0;
        for (auto &b : beta) { b = nextd(nullptr) + 1e-3; tot += b; }
        for (auto &b : beta) b /= tot;
        if (trial % 5 == 0) beta[0] = 1e-310;
        for (auto &v : n) v = nextd(nullptr) < 0.3 ? 0.0 : std::floor(40.0 * nextd(nullptr));
        std::vector<int64_t> m((size_t)T * K * K);
        const int rc = dlsm::host_sample_tables(&g, T, K, n.data(), beta.data(), 0.7, 1.3,
                                                trial % 3 ? 4.0 : 0.0, m.data());
        if (rc != 0) { fprintf(stderr, "host_sample_tables rc %d\n", rc); return 2; }
        for (size_t q = 0; q < m.size(); ++q)
            if (m[q] < 0 || (double)m[q] > n[q]) { fprintf(stderr, "bad table count\n"); return 3; }
    }
    {   // a probability outside [0, 1] is reported, not drawn
        double n[1] = {3.0}, beta[1] = {-1.0};
        int64_t m[1];
        if (dlsm::host_sample_tables(&g, 1, 1, n, beta, 1.0, 1.0, 0.0, m) != -1) return 4;
    }
    // tridiagonal eigenpairs: residuals and orthogonality for sizes 1 .. 96, incl. clusters
    for (int k = 1; k <= 96; k += (k < 8 ? 1 : 11)) {
        std::vector<double> a(k), b(k > 1 ? k - 1 : 1, 0.0);
        for (int i = 0; i < k; ++i) a[i] = (k % 3 == 0) ? 2.0 : 4.0 * nextd(nullptr) - 2.0;
        for (int i = 0; i + 1 < k; ++i) b[i] = (k % 3 == 0) ? -1.0 : nextd(nullptr) - 0.5;
        const int nd = k < 4 ? k : 4;
        std::vector<double> theta(nd);
        std::vector<std::vector<double>> S;
        dlsm_host::tridiag_top(a.data(), b.data(), k, nd, theta.data(), S);
        for (int m = 0; m < nd; ++m) {
            double res = 0.0, nrm = 0.0;
            for (int i = 0; i < k; ++i) {
                double v = (a[i] - theta[m]) * S[m][i];
                if (i > 0) v += b[i - 1] * S[m][i - 1];
                if (i + 1 < k) v += b[i] * S[m][i + 1];
                res += v * v; nrm += S[m][i] * S[m][i];
            }
            if (!(std::sqrt(res) < 1e-8) || !(std::fabs(nrm - 1.0) < 1e-10)) {
                fprintf(stderr, "tridiag_top: k=%d m=%d residual %.3e norm %.12f\n", k, m,
                        std::sqrt(res), nrm);
                return 5;
            }
            if (m > 0 && theta[m] > theta[m - 1] + 1e-12) return 6;
        }
    }
    printf("sanitize_host ok\n");
    return 0;
}
