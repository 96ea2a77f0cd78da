// Host-side auxiliary draws of the sticky HDP that must stay on the caller's MT19937
// stream: the reference draws them from its numpy RandomState in a fixed
// order (sample_auxillary.py:6-28), and the golden traces pin that order, so the stream is
// consumed here through numpy's own bit-generator interface instead of a device RNG.
#pragma once
#include <math.h>
#include <stdint.h>

namespace dlsm {

// layout of numpy's bitgen_t (numpy/random/bitgen.h), reached from Python through
// RandomState._bit_generator.ctypes.bit_generator
struct NumpyBitGen {
    void *state;
    uint64_t (*next_uint64)(void *);
    uint32_t (*next_uint32)(void *);
    double (*next_double)(void *);
    uint64_t (*next_raw)(void *);
};

// RandomState.binomial(1, p): numpy's legacy inversion sampler with n = 1.  It runs with
// pe = min(p, 1 - p), compares one uniform with qn = exp(log(1 - pe)) and mirrors the result
// when p > 0.5.  qn is within a few ulp of 1 - pe, so the libm calls are only made when the
// uniform falls in that band (or next to 1, where the sampler may ask for a second uniform).
static inline int legacy_bernoulli(NumpyBitGen *g, double p) {
    const bool mirrored = !(p <= 0.5);
    const double pe = mirrored ? 1.0 - p : p;
    const double q = 1.0 - pe;
    double U = g->next_double(g->state);
    int X;
    if (U < q * (1.0 - 1e-14) && U < 1.0 - 1e-9) X = 0;
    else if (U > q * (1.0 + 1e-14) && U < 1.0 - 1e-9) X = 1;
    else {
        const double qn = exp(1.0 * log(q));
        const double bound = fmin(1.0, pe + 10.0 * sqrt(pe * q + 1.0));
        double px = qn;
        X = 0;
        while (U > px) {
            ++X;
            if ((double)X > (double)(int64_t)bound) { X = 0; px = qn; U = g->next_double(g->state); }
            else { U -= px; px = ((1 - X + 1) * pe * px) / (X * q); }
        }
    }
    return mirrored ? 1 - X : X;
}

// m[t][j][k] = number of tables of restaurant (t, j) serving dish k, given the transition
// counts n (float64 [T][K][K], n[0][0][:] = initial counts): customer c of a cell opens a new
// table with probability pr / (pr + c), pr = alpha_init beta_k at t = 0 and
// alpha beta_k + kappa [j = k] afterwards.  Cells in the reference's order: (0, 0, k), then
// (t, j, k) for t = 1 .. T-1.  Returns -1 if a probability is not in [0, 1] (numpy raises).
static inline int host_sample_tables(NumpyBitGen *g, int T, int K, const double *n,
                                     const double *beta, double alpha_init, double alpha,
                                     double kappa, int64_t *m) {
    for (size_t q = 0; q < (size_t)T * K * K; ++q) m[q] = 0;
    for (int t = 0; t < T; ++t)
        for (int j = 0; j < (t == 0 ? 1 : K); ++j)
            for (int k = 0; k < K; ++k) {
                const size_t cell = ((size_t)t * K + j) * K + k;
                const int64_t cnt = (int64_t)n[cell];
                const double pr = t == 0 ? alpha_init * beta[k]
                                         : alpha * beta[k] + kappa * (j == k ? 1.0 : 0.0);
                int64_t tables = 0;
                for (int64_t c = 0; c < cnt; ++c) {
                    const double p = pr / (pr + (double)c);
                    if (!(p >= 0.0 && p <= 1.0)) return -1;
                    tables += legacy_bernoulli(g, p);
                }
                m[cell] = tables;
            }
    return 0;
}

}  // namespace dlsm
