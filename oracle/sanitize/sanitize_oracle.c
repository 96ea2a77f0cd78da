/* CPU sanitizer driver (test infrastructure): exercises the C oracle's entry points on small
 * random inputs under -fsanitize=address,undefined (tests/test_sanitizers_cpu.py builds and
 * runs it).  Exit code 0 = no report. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../dynetlsm_oracle.h"

static double urand(uint64_t *s) {
    *s = *s * 6364136223846793005ull + 1442695040888963407ull;
    return (double)((*s >> 11) + 1) / 9007199254740993.0;
}

int main(void) {
    enum { T = 3, N = 41, D = 2, K = 4, C = 5 };
    uint64_t s = 7;
    double *Y = calloc((size_t)T * N * N, sizeof(double));
    double *Yd = calloc((size_t)T * N * N, sizeof(double));
    double *X = malloc(sizeof(double) * T * N * D);
    double radii[N], mu[K * D], sigma[K], w[T * K * K];
    int64_t z[T * N];
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                if (i != j && urand(&s) < 0.2) Yd[((size_t)t * N + i) * N + j] = 1.0;
                if (i < j && urand(&s) < 0.2) {
                    Y[((size_t)t * N + i) * N + j] = 1.0;
                    Y[((size_t)t * N + j) * N + i] = 1.0;
                }
            }
    for (int q = 0; q < T * N * D; ++q) X[q] = 2.0 * urand(&s) - 1.0;
    double rs = 0.0;
    for (int i = 0; i < N; ++i) { radii[i] = urand(&s); rs += radii[i]; }
    for (int i = 0; i < N; ++i) radii[i] /= rs;
    for (int q = 0; q < K * D; ++q) mu[q] = 2.0 * urand(&s) - 1.0;
    for (int k = 0; k < K; ++k) sigma[k] = 0.3 + urand(&s);
    for (int q = 0; q < T * N; ++q) z[q] = (int64_t)(urand(&s) * K) % K;
    for (int r = 0; r < T * K; ++r) {
        double tot = 0.0;
        for (int k = 0; k < K; ++k) { w[r * K + k] = urand(&s); tot += w[r * K + k]; }
        for (int k = 0; k < K; ++k) w[r * K + k] /= tot;
    }
    double acc = 0.0;
    acc += orc_loglik_undirected(Y, X, T, N, D, 0.4, 0);
    acc += orc_loglik_undirected(Y, X, T, N, D, 0.4, 1);
    acc += orc_loglik_directed(Yd, X, radii, T, N, D, 0.3, 0.6, 0);
    for (int j = 0; j < N; j += 7) {
        acc += orc_partial_loglikelihood(Y + (size_t)N * N, X + (size_t)N * D, N, D, 0.4, j, 0);
        acc += orc_directed_partial_loglikelihood(Yd, X, radii, N, D, 0.3, 0.6, j, 0);
    }
    /* sweeps: random-walk prior (all three models need their own state; the undirected and
     * the directed one here) and the AR-mixture prior */
    double step[T * N];
    int32_t nacc[T * N], nsteps[T * N], until[T * N];
    for (int q = 0; q < T * N; ++q) { step[q] = 0.1; nacc[q] = 0; nsteps[q] = 0; until[q] = 2; }
    orc_chain c;
    memset(&c, 0, sizeof(c));
    c.T = T; c.N = N; c.D = D; c.model = 0; c.Y = Y; c.X = X;
    c.intercept[0] = 0.4; c.prior_kind = 0; c.tau_sq = 2.0; c.sigma_sq = 0.1;
    c.step_size = step; c.n_accepted = nacc; c.n_steps = nsteps; c.steps_until_tune = until;
    c.tune = 4; c.tune_interval = 2; c.seed = 99; c.chain = 1;
    for (uint32_t it = 1; it <= 3; ++it) { c.iter = it; orc_sweep_positions(&c); }
    orc_scalar_sampler is = {0.1, 0, 0, 100, -1, 100};
    for (uint32_t it = 4; it <= 5; ++it) { c.iter = it; acc += orc_lsm_iteration_undirected(&c, &is, 0.4, 2.0); }
    c.prior_kind = 1; c.mu = mu; c.sigma = sigma; c.lmbda = 0.8; c.z = z; c.K = K;
    for (uint32_t it = 6; it <= 7; ++it) { c.iter = it; orc_sweep_positions(&c); }
    c.model = 1; c.Y = Yd; c.radii = radii; c.intercept[1] = 0.6; c.prior_kind = 0;
    for (uint32_t it = 8; it <= 9; ++it) { c.iter = it; orc_sweep_positions(&c); }
    orc_center(X, T, N, D);
    /* label block update and the Gaussian table */
    int64_t zo[T * N], nk[T * K];
    double n[T * K * K], table[T * K];
    orc_sample_labels(X, mu, sigma, 0.8, w, T, N, D, K, 5, 0, 3, zo, n, nk);
    orc_gaussian_likelihood(X, N * D, mu, sigma, 0.8, T, D, K, 1, table);
    for (int q = 0; q < T * N; ++q) if (zo[q] < 0 || zo[q] >= K) return 3;
    /* case-control lists */
    int64_t deg[T * N * 2], ie[T * N * N], oe[T * N * N], ci[T * N * C], co[T * N * C];
    memset(ie, 0, sizeof(ie)); memset(oe, 0, sizeof(oe));
    int Din = 0, Dout = 0;
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < N; ++i) {
            int a = 0, b = 0;
            for (int j = 0; j < N; ++j) {
                if (Yd[((size_t)t * N + j) * N + i] == 1.0) ie[((size_t)t * N + i) * N + a++] = j;
                if (Yd[((size_t)t * N + i) * N + j] == 1.0) oe[((size_t)t * N + i) * N + b++] = j;
            }
            deg[(t * N + i) * 2] = a; deg[(t * N + i) * 2 + 1] = b;
            if (a > Din) Din = a;
            if (b > Dout) Dout = b;
            for (int q = 0; q < C; ++q) {
                ci[(t * N + i) * C + q] = q < 3 ? (i + 1 + q) % N : -1;
                co[(t * N + i) * C + q] = q < 4 ? (i + 2 + q) % N : -1;
            }
        }
    acc += orc_approx_loglik_directed(X, radii, oe, N, deg, co, C, T, N, D, 0.3, 0.6, 0);
    for (int j = 0; j < N; j += 5)
        acc += orc_approx_directed_partial_loglikelihood(X, radii, ie, N, oe, N, deg, ci, co, C, N,
                                                         D, 0.3, 0.6, j, 0, j & 1);
    /* the case-control sweep (model 2) */
    c.model = 2; c.Y = NULL; c.in_edges = ie; c.Din = N; c.out_edges = oe; c.Dout = N;
    c.degree = deg; c.ctrl_in = ci; c.ctrl_out = co; c.C = C;
    for (uint32_t it = 10; it <= 11; ++it) { c.iter = it; orc_sweep_positions(&c); }
    (void)Din; (void)Dout;
    free(Y); free(Yd); free(X);
    if (!isfinite(acc)) { fprintf(stderr, "non-finite accumulator\n"); return 2; }
    printf("sanitize_oracle ok %.6f\n", acc);
    return 0;
}
