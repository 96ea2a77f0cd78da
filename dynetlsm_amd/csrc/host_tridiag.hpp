// Top eigenpairs of a symmetric tridiagonal matrix on the host (k <= a few hundred): the
// small dense problem inside the Lanczos eigen step of the initialisation pipeline
// (latent_space.py:71-89; the reference calls scipy.linalg.eigh on the N x N matrix).  Plain
// C++ with no HIP dependency, so that the CPU sanitizer job (tests/test_sanitizers_cpu.py) can
// compile it on its own.
#pragma once
#include <math.h>

#include <algorithm>
#include <vector>

namespace dlsm_host {

// ---- top eigenpairs of a symmetric tridiagonal matrix (host, k <= a few 100) ----
// Sturm-sequence bisection for the eigenvalues, inverse iteration with partial
// pivoting for the vectors (the scheme of LAPACK's dstebz / dstein).
int sturm_count(const double *a, const double *b, int k, double x, double tiny) {
    int cnt = 0;
    double q = a[0] - x;
    if (q < 0.0) ++cnt;
    for (int i = 1; i < k; ++i) {
        if (fabs(q) < tiny) q = q < 0.0 ? -tiny : tiny;
        q = a[i] - x - b[i - 1] * b[i - 1] / q;
        if (q < 0.0) ++cnt;
    }
    return cnt;
}

// solve (T - x I) s = rhs in place (rhs -> s); Gaussian elimination with partial
// pivoting on the tridiagonal (fill-in: a second superdiagonal)
void tridiag_shift_solve(const double *a, const double *b, int k, double x, double tiny,
                         std::vector<double> &s) {
    std::vector<double> d(k), u1(k, 0.0), u2(k, 0.0), l(k, 0.0);
    for (int i = 0; i < k; ++i) d[i] = a[i] - x;
    for (int i = 0; i + 1 < k; ++i) u1[i] = b[i];
    for (int i = 0; i + 1 < k; ++i) {
        const double sub = b[i];
        if (fabs(d[i]) >= fabs(sub)) {
            if (fabs(d[i]) < tiny) d[i] = tiny;
            const double m = sub / d[i];
            d[i + 1] -= m * u1[i];
            // u1[i + 1] unchanged (u2[i] == 0 here)
            s[i + 1] -= m * s[i];
        } else {
            // swap rows i and i+1
            const double m = d[i] / sub;
            const double di = sub, u1i = d[i + 1], u2i = (i + 2 < k) ? b[i + 1] : 0.0;
            const double nd = u1[i] - m * u1i;
            const double nu = -m * u2i;
            d[i] = di; u1[i] = u1i; u2[i] = u2i;
            d[i + 1] = nd;
            if (i + 2 < k) u1[i + 1] = nu;
            const double si = s[i + 1];
            s[i + 1] = s[i] - m * si;
            s[i] = si;
        }
    }
    if (fabs(d[k - 1]) < tiny) d[k - 1] = tiny;
    for (int i = k - 1; i >= 0; --i) {
        double v = s[i];
        if (i + 1 < k) v -= u1[i] * s[i + 1];
        if (i + 2 < k) v -= u2[i] * s[i + 2];
        s[i] = v / d[i];
    }
}

// theta[m], S[m][0..k) for the nd algebraically largest eigenvalues (descending)
void tridiag_top(const double *a, const double *b, int k, int nd, double *theta,
                 std::vector<std::vector<double>> &S) {
    double lo = a[0], hi = a[0], nrm = 0.0;
    for (int i = 0; i < k; ++i) {
        const double r = (i > 0 ? fabs(b[i - 1]) : 0.0) + (i + 1 < k ? fabs(b[i]) : 0.0);
        lo = std::min(lo, a[i] - r);
        hi = std::max(hi, a[i] + r);
        nrm = std::max(nrm, fabs(a[i]) + r);
    }
    const double tiny = std::max(nrm, 1e-300) * 2.3e-16;
    S.assign(nd, std::vector<double>(k, 0.0));
    for (int m = 0; m < nd; ++m) {
        const int want = k - m;            // smallest x with count(x) >= want
        double l = lo - tiny, r = hi + tiny;
        for (int it = 0; it < 200; ++it) {
            const double mid = 0.5 * (l + r);
            if (mid <= l || mid >= r) break;
            if (sturm_count(a, b, k, mid, tiny) >= want) r = mid; else l = mid;
        }
        theta[m] = 0.5 * (l + r);
        std::vector<double> &s = S[m];
        for (int i = 0; i < k; ++i) s[i] = 1.0 + 0.01 * ((i * 7919 + m * 104729) % 97);
        for (int it = 0; it < 4; ++it) {
            tridiag_shift_solve(a, b, k, theta[m], tiny, s);
            for (int p = 0; p < m; ++p) {              // clusters: keep the basis orthogonal
                double dot = 0.0;
                for (int i = 0; i < k; ++i) dot += s[i] * S[p][i];
                for (int i = 0; i < k; ++i) s[i] -= dot * S[p][i];
            }
            double ss = 0.0;
            for (int i = 0; i < k; ++i) ss += s[i] * s[i];
            const double inv = 1.0 / sqrt(ss);
            for (int i = 0; i < k; ++i) s[i] *= inv;
        }
    }
}

}  // namespace dlsm_host
