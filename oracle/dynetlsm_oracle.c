/*
 * dynetlsm_oracle.c -- CPU ORACLE (test infrastructure, see dynetlsm_oracle.h).
 *
 * Scalar double-precision restatement of the reference's Gibbs hot path.
 * Reference citations are relative to joshloyal/dynetlsm @ v0.1.0.
 * Not shipped, not a fallback: the product path fails without its HIP library.
 */
#include "dynetlsm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al. 2011), the engine's counter RNG        */
/* ------------------------------------------------------------------ */
void orc_philox4x32(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2,
                    uint32_t c3, uint32_t out[4]) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static double u53(uint32_t hi, uint32_t lo) {
    /* 53 random bits -> (0, 1] ; every value exactly representable */
    double k = (double)(hi >> 5) * 67108864.0 + (double)(lo >> 6);
    return (k + 1.0) * (1.0 / 9007199254740992.0);
}

void orc_philox_uniform2(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2,
                         uint32_t c3, double u[2]) {
    uint32_t r[4];
    orc_philox4x32(seed, c0, c1, c2, c3, r);
    u[0] = u53(r[0], r[1]);
    u[1] = u53(r[2], r[3]);
}

static void box_muller(const double u[2], double z[2]) {
    double r = sqrt(-2.0 * log(u[0]));
    double a = 6.283185307179586476925286766559 * u[1];
    z[0] = r * cos(a);
    z[1] = r * sin(a);
}

static uint32_t stream_word(uint32_t chain, uint32_t stream) {
    return (chain << 8) | stream;
}

/* ------------------------------------------------------------------ */
/* per-dyad pieces                                                     */
/* ------------------------------------------------------------------ */
static double dist_of(const double *a, const double *b, int D, int squared) {
    double s = 0.0;
    for (int d = 0; d < D; ++d) {
        double df = a[d] - b[d];
        s += df * df;
    }
    return squared ? s : sqrt(s);
}

/* the reference writes log(1 + exp(eta)) literally (static_network_fast.pyx:42) */
static double log1pexp_ref(double eta) { return log(1.0 + exp(eta)); }

/* a1 with the node's position given explicitly (the closure sets X[t, j] = x
 * before calling, sample_latent_positions.py:101) */
static double partial_x(const double *Y, const double *X, int N, int D,
                        double intercept, int node, const double *x, int squared) {
    double ll = 0.0;
    for (int i = 0; i < N; ++i) {
        if (i == node) continue;
        double eta = intercept - dist_of(X + (size_t)i * D, x, D, squared);
        ll += Y[(size_t)node * N + i] * eta;
        ll -= log1pexp_ref(eta);
    }
    return ll;
}

double orc_partial_loglikelihood(const double *Y, const double *X, int N, int D,
                                 double intercept, int node, int squared) {
    return partial_x(Y, X, N, D, intercept, node, X + (size_t)node * D, squared);
}

static double directed_partial_x(const double *Y, const double *X,
                                 const double *radii, int N, int D, double b_in,
                                 double b_out, int node, const double *x,
                                 int squared) {
    double ll = 0.0;
    for (int j = 0; j < N; ++j) {
        if (j == node) continue;
        double dist = dist_of(X + (size_t)j * D, x, D, squared);
        /* Y_ijt : directed_likelihoods_fast.pyx:70-73 */
        double eta = b_in * (1 - dist / radii[j]);
        eta += b_out * (1 - dist / radii[node]);
        ll += Y[(size_t)node * N + j] * eta - log1pexp_ref(eta);
        /* Y_jit : :75-78 */
        eta = b_in * (1 - dist / radii[node]);
        eta += b_out * (1 - dist / radii[j]);
        ll += Y[(size_t)j * N + node] * eta - log1pexp_ref(eta);
    }
    return ll;
}

double orc_directed_partial_loglikelihood(const double *Y, const double *X,
                                          const double *radii, int N, int D,
                                          double b_in, double b_out, int node,
                                          int squared) {
    return directed_partial_x(Y, X, radii, N, D, b_in, b_out, node,
                              X + (size_t)node * D, squared);
}

static double approx_partial_x(const double *X, const double *radii,
                               const int64_t *in_edges, int Din,
                               const int64_t *out_edges, int Dout,
                               const int64_t *degree, const int64_t *ctrl_in,
                               const int64_t *ctrl_out, int C, int N, int D,
                               double b_in, double b_out, int node,
                               const double *x, int squared, int ref_compat) {
    int in_deg = (int)degree[(size_t)node * 2 + 0];
    int out_deg = (int)degree[(size_t)node * 2 + 1];
    double ll = 0.0;
#define POS(idx) ((idx) == node ? x : X + (size_t)(idx) * D)
    /* in edges :107-118 */
    for (int j = 0; j < in_deg; ++j) {
        int64_t e = in_edges[(size_t)node * Din + j];
        double dist = dist_of(POS(e), x, D, squared);
        double eta = b_in * (1 - dist / radii[node]);
        eta += b_out * (1 - dist / radii[e]);
        ll += eta - log1pexp_ref(eta);
    }
    /* out edges :121-132 */
    for (int j = 0; j < out_deg; ++j) {
        int64_t e = out_edges[(size_t)node * Dout + j];
        double dist = dist_of(POS(e), x, D, squared);
        double eta = b_in * (1 - dist / radii[e]);
        eta += b_out * (1 - dist / radii[node]);
        ll += eta - log1pexp_ref(eta);
    }
    /* control estimate, in direction :135-155 */
    double control = 0.0, n_ctrl = 0.0;
    for (int j = 0; j < C; ++j) {
        int64_t e = ctrl_in[(size_t)node * C + j];
        if (e == -1) break;
        double dist = dist_of(POS(e), x, D, squared);
        double eta = b_in * (1 - dist / radii[node]);
        eta += b_out * (1 - dist / radii[e]);
        control += log1pexp_ref(eta);
        n_ctrl += 1;
    }
    ll -= ((double)(N - in_deg - 1) / n_ctrl) * control;
    /* control estimate, out direction :157-180 */
    control = 0.0; n_ctrl = 0.0;
    for (int j = 0; j < C; ++j) {
        int64_t sentinel = ref_compat ? ctrl_in[(size_t)node * C + j]
                                      : ctrl_out[(size_t)node * C + j];
        if (sentinel == -1) break;
        int64_t e = ctrl_out[(size_t)node * C + j];
        double dist = dist_of(POS(e), x, D, squared);
        double eta = b_in * (1 - dist / radii[e]);
        eta += b_out * (1 - dist / radii[node]);
        control += log1pexp_ref(eta);
        n_ctrl += 1;
    }
    ll -= ((double)(N - out_deg - 1) / n_ctrl) * control;
#undef POS
    return ll;
}

double orc_approx_directed_partial_loglikelihood(
    const double *X, const double *radii, const int64_t *in_edges, int Din,
    const int64_t *out_edges, int Dout, const int64_t *degree,
    const int64_t *ctrl_in, const int64_t *ctrl_out, int C, int N, int D,
    double b_in, double b_out, int node, int squared, int ref_compat) {
    return approx_partial_x(X, radii, in_edges, Din, out_edges, Dout, degree,
                            ctrl_in, ctrl_out, C, N, D, b_in, b_out, node,
                            X + (size_t)node * D, squared, ref_compat);
}

double orc_loglik_undirected(const double *Y, const double *X, int T, int N,
                             int D, double intercept, int squared) {
    /* upper triangle k=1, every dyad once: network_likelihoods.py:30-33 */
    double ll = 0.0;
    for (int t = 0; t < T; ++t) {
        const double *Yt = Y + (size_t)t * N * N;
        const double *Xt = X + (size_t)t * N * D;
        for (int i = 0; i < N; ++i)
            for (int j = i + 1; j < N; ++j) {
                double eta = intercept - dist_of(Xt + (size_t)i * D,
                                                 Xt + (size_t)j * D, D, squared);
                ll += Yt[(size_t)i * N + j] * eta - log1pexp_ref(eta);
            }
    }
    return ll;
}

double orc_loglik_directed(const double *Y, const double *X, const double *radii,
                           int T, int N, int D, double b_in, double b_out,
                           int squared) {
    double ll = 0.0;
    for (int t = 0; t < T; ++t) {
        const double *Yt = Y + (size_t)t * N * N;
        const double *Xt = X + (size_t)t * N * D;
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                if (i == j) continue;
                double dist = dist_of(Xt + (size_t)i * D, Xt + (size_t)j * D, D,
                                      squared);
                double d_in = 1 - dist / radii[j];
                double d_out = 1 - dist / radii[i];
                double eta = b_in * d_in + b_out * d_out;
                ll += Yt[(size_t)i * N + j] * eta - log1pexp_ref(eta);
            }
    }
    return ll;
}

double orc_approx_loglik_directed(const double *X, const double *radii,
                                  const int64_t *out_edges, int Dout,
                                  const int64_t *degree, const int64_t *ctrl_out,
                                  int C, int T, int N, int D, double b_in,
                                  double b_out, int squared) {
    double ll = 0.0;
    for (int t = 0; t < T; ++t) {
        const double *Xt = X + (size_t)t * N * D;
        for (int i = 0; i < N; ++i) {
            size_t ti = (size_t)t * N + i;
            int out_deg = (int)degree[ti * 2 + 1];
            for (int j = 0; j < out_deg; ++j) {
                int64_t e = out_edges[ti * Dout + j];
                double dist = dist_of(Xt + (size_t)e * D, Xt + (size_t)i * D, D,
                                      squared);
                double eta = b_in * (1 - dist / radii[e]);
                eta += b_out * (1 - dist / radii[i]);
                ll += eta - log1pexp_ref(eta);
            }
            double control = 0.0, n_ctrl = 0.0;
            for (int j = 0; j < C; ++j) {
                int64_t e = ctrl_out[ti * C + j];
                if (e == -1) break;
                double dist = dist_of(Xt + (size_t)e * D, Xt + (size_t)i * D, D,
                                      squared);
                double eta = b_in * (1 - dist / radii[e]);
                eta += b_out * (1 - dist / radii[i]);
                control += log1pexp_ref(eta);
                n_ctrl += 1;
            }
            ll -= ((double)(N - out_deg - 1) / n_ctrl) * control;
        }
    }
    return ll;
}

double orc_spherical_normal_log_pdf(const double *x, const double *mean, int D,
                                    double var) {
    double ss = 0.0;
    for (int k = 0; k < D; ++k) ss += (x[k] - mean[k]) * (x[k] - mean[k]);
    ss *= 0.5 * (1. / var);
    return -0.5 * D * log(2 * M_PI * var) - ss;
}

void orc_gaussian_likelihood(const double *Xi, int ldx, const double *mu,
                             const double *sigma, double lmbda, int T, int D,
                             int K, int normalize, double *out) {
    double muk[64];
    for (int t = 0; t < T; ++t) {
        for (int k = 0; k < K; ++k) {
            if (t == 0) {
                out[t * K + k] = orc_spherical_normal_log_pdf(
                    Xi, mu + (size_t)k * D, D, sigma[k]);
            } else {
                for (int j = 0; j < D; ++j)
                    muk[j] = lmbda * mu[(size_t)k * D + j] +
                             (1 - lmbda) * Xi[(size_t)(t - 1) * ldx + j];
                out[t * K + k] = orc_spherical_normal_log_pdf(
                    Xi + (size_t)t * ldx, muk, D, sigma[k]);
            }
        }
        if (normalize) {
            double m = out[t * K];
            for (int k = 1; k < K; ++k) if (out[t * K + k] > m) m = out[t * K + k];
            for (int k = 0; k < K; ++k) out[t * K + k] -= m;
        }
        for (int k = 0; k < K; ++k) out[t * K + k] = exp(out[t * K + k]);
    }
}

/* ------------------------------------------------------------------ */
/* sweep                                                               */
/* ------------------------------------------------------------------ */
double orc_node_logp(const orc_chain *c, int t, int j, const double *x) {
    const int N = c->N, D = c->D, T = c->T;
    const double *Xt = c->X + (size_t)t * N * D;
    double ll;
    if (c->model == 0) {
        ll = partial_x(c->Y + (size_t)t * N * N, Xt, N, D, c->intercept[0], j, x,
                       c->squared);
    } else if (c->model == 1) {
        ll = directed_partial_x(c->Y + (size_t)t * N * N, Xt, c->radii, N, D,
                                c->intercept[0], c->intercept[1], j, x,
                                c->squared);
    } else {
        size_t tn = (size_t)t * N;
        ll = approx_partial_x(Xt, c->radii, c->in_edges + tn * c->Din, c->Din,
                              c->out_edges + tn * c->Dout, c->Dout,
                              c->degree + tn * 2, c->ctrl_in + tn * c->C,
                              c->ctrl_out + tn * c->C, c->C, N, D,
                              c->intercept[0], c->intercept[1], j, x, c->squared,
                              0);
    }
    if (c->prior_kind == 0) {
        /* sample_latent_positions.py:132-140 */
        double s = 0.0;
        if (t == 0) {
            for (int d = 0; d < D; ++d) s += x[d] * x[d];
            ll -= 0.5 * s / c->tau_sq;
        } else {
            const double *xp = c->X + ((size_t)(t - 1) * N + j) * D;
            for (int d = 0; d < D; ++d) s += (x[d] - xp[d]) * (x[d] - xp[d]);
            ll -= 0.5 * s / c->sigma_sq;
        }
        if (t < T - 1) {
            const double *xn = c->X + ((size_t)(t + 1) * N + j) * D;
            s = 0.0;
            for (int d = 0; d < D; ++d) s += (xn[d] - x[d]) * (xn[d] - x[d]);
            ll -= 0.5 * s / c->sigma_sq;
        }
    } else {
        /* sample_latent_positions.py:187-199 */
        const double lm = c->lmbda;
        int64_t zt = c->z[(size_t)t * N + j];
        const double *m = c->mu + (size_t)zt * D;
        double s = 0.0;
        if (t == 0) {
            for (int d = 0; d < D; ++d) s += (x[d] - m[d]) * (x[d] - m[d]);
        } else {
            const double *xp = c->X + ((size_t)(t - 1) * N + j) * D;
            for (int d = 0; d < D; ++d) {
                double df = x[d] - (1 - lm) * xp[d] - lm * m[d];
                s += df * df;
            }
        }
        ll -= 0.5 * s / c->sigma[zt];
        if (t < T - 1) {
            int64_t zn = c->z[(size_t)(t + 1) * N + j];
            const double *mn = c->mu + (size_t)zn * D;
            const double *xn = c->X + ((size_t)(t + 1) * N + j) * D;
            s = 0.0;
            for (int d = 0; d < D; ++d) {
                double df = xn[d] - (1 - lm) * x[d] - lm * mn[d];
                s += df * df;
            }
            ll -= 0.5 * s / c->sigma[zn];
        }
    }
    return ll;
}

/* metropolis.py:5-20 */
static double tune_rw(double step, double rate) {
    if (rate < 0.001) step *= 0.1;
    else if (rate < 0.05) step *= 0.5;
    else if (rate < 0.25) step *= 0.9;
    else if (rate > 0.95) step *= 10.0;
    else if (rate > 0.75) step *= 2.0;
    else if (rate > 0.4) step *= 1.1;
    return step;
}

/* metropolis.py:110-136 (counter bookkeeping incl. the tune_interval+1 window) */
static void metropolis_bookkeeping(double *step, int32_t *n_acc, int32_t *n_steps,
                                   int32_t *until, int tune, int tune_interval,
                                   int accepted) {
    *n_acc += accepted;
    *n_steps += 1;
    if (tune >= 0) {
        if (*n_steps < tune && *until == 0) {
            double rate = (double)*n_acc / (double)tune_interval;
            *step = tune_rw(*step, rate);
            *n_acc = 0;
            *until = tune_interval;
        } else {
            *until -= 1;
        }
    }
}

static void sweep_slice(orc_chain *c, int t) {
    const int N = c->N, D = c->D;
    double x0[64], x[64], u2[2], z2[2];
    for (int j = 0; j < N; ++j) {
        size_t tj = (size_t)t * N + j;
        double *Xtj = c->X + tj * D;
        /* metropolis.py:44 : x = x0 + step * randn(d) */
        for (int d = 0; d < D; d += 2) {
            orc_philox_uniform2(c->seed, (uint32_t)j,
                                (uint32_t)t | ((uint32_t)(d / 2) << 16), c->iter,
                                stream_word(c->chain, ORC_STREAM_SWEEP_NORMAL), u2);
            box_muller(u2, z2);
            x0[d] = Xtj[d];
            x[d] = x0[d] + c->step_size[tj] * z2[0];
            if (d + 1 < D) {
                x0[d + 1] = Xtj[d + 1];
                x[d + 1] = x0[d + 1] + c->step_size[tj] * z2[1];
            }
        }
        /* :47 : logp(x) - logp(x0), proposal first */
        double ratio = orc_node_logp(c, t, j, x) - orc_node_logp(c, t, j, x0);
        orc_philox_uniform2(c->seed, (uint32_t)j, (uint32_t)t, c->iter,
                            stream_word(c->chain, ORC_STREAM_SWEEP_UNIFORM), u2);
        int accepted = !(log(u2[0]) >= ratio);   /* :50 */
        if (accepted) memcpy(Xtj, x, sizeof(double) * D);
        metropolis_bookkeeping(&c->step_size[tj], &c->n_accepted[tj],
                               &c->n_steps[tj], &c->steps_until_tune[tj], c->tune,
                               c->tune_interval, accepted);
    }
}

void orc_sweep_positions(orc_chain *c) {
    for (int t = 0; t < c->T; t += 2) sweep_slice(c, t);
    for (int t = 1; t < c->T; t += 2) sweep_slice(c, t);
}

void orc_center(double *X, int T, int N, int D) {
    for (int d = 0; d < D; ++d) {
        double s = 0.0;
        for (size_t i = 0; i < (size_t)T * N; ++i) s += X[i * D + d];
        s /= (double)((size_t)T * N);
        for (size_t i = 0; i < (size_t)T * N; ++i) X[i * D + d] -= s;
    }
}

/* ------------------------------------------------------------------ */
/* labels                                                              */
/* ------------------------------------------------------------------ */
void orc_sample_labels(const double *X, const double *mu, const double *sigma,
                       double lmbda, const double *w, int T, int N, int D, int K,
                       uint64_t seed, uint32_t chain, uint32_t iter, int64_t *z,
                       double *n, int64_t *nk) {
    double *L = (double *)malloc(sizeof(double) * T * K);
    double *bm = (double *)malloc(sizeof(double) * T * K);
    double *pm = (double *)malloc(sizeof(double) * T * K);
    memset(n, 0, sizeof(double) * T * K * K);
    memset(nk, 0, sizeof(int64_t) * T * K);
    for (int i = 0; i < N; ++i) {
        /* sample_labels.py:159 : X[:, i] is a T x D view with row stride N*D */
        orc_gaussian_likelihood(X + (size_t)i * D, N * D, mu, sigma, lmbda, T, D,
                                K, 0, L);
        for (int k = 0; k < K; ++k) bm[(T - 1) * K + k] = 1.0;
        /* backward messages :164-170 */
        for (int t = T - 1; t > 0; --t) {
            for (int k = 0; k < K; ++k) pm[t * K + k] = L[t * K + k] * bm[t * K + k];
            double tot = 0.0;
            for (int r = 0; r < K; ++r) {
                double s = 0.0;
                for (int k = 0; k < K; ++k)
                    s += w[((size_t)t * K + r) * K + k] * pm[t * K + k];
                bm[(t - 1) * K + r] = s;
                tot += s;
            }
            for (int r = 0; r < K; ++r) bm[(t - 1) * K + r] /= tot;
        }
        for (int k = 0; k < K; ++k) pm[k] = L[k] * bm[k];
        /* forward sampling :173-188 */
        int64_t zprev = 0;
        for (int t = 0; t < T; ++t) {
            const double *wrow = (t == 0) ? w : w + ((size_t)t * K + zprev) * K;
            double u2[2];
            orc_philox_uniform2(seed, (uint32_t)i, (uint32_t)t, iter,
                                stream_word(chain, ORC_STREAM_LABELS), u2);
            /* sample_categorical :16-19 */
            double total = 0.0;
            for (int k = 0; k < K; ++k) total += wrow[k] * pm[t * K + k];
            double u = u2[0] * total;
            double cdf = 0.0;
            int64_t zt = 0;
            for (int k = 0; k < K; ++k) {
                cdf += wrow[k] * pm[t * K + k];
                zt += (u > cdf);
            }
            z[(size_t)t * N + i] = zt;
            if (t == 0) n[zt] += 1;
            else n[((size_t)t * K + zprev) * K + zt] += 1;
            nk[t * K + zt] += 1;
            zprev = zt;
        }
    }
    free(L); free(bm); free(pm);
}

/* ------------------------------------------------------------------ */
/* LSM iteration (undirected)                                          */
/* ------------------------------------------------------------------ */
double orc_lsm_log_prior(const double *X, int T, int N, int D, double tau_sq,
                         double sigma_sq, const double *intercept, int n_intercept,
                         const double *intercept_prior, double intercept_var) {
    double lp = 0.0;
    for (int t = 0; t < T; ++t) {
        double s = 0.0;
        for (int i = 0; i < N; ++i) {
            const double *x = X + ((size_t)t * N + i) * D;
            double r = 0.0;
            if (t == 0) {
                for (int d = 0; d < D; ++d) r += x[d] * x[d];
                s += 0.5 * r / tau_sq;
            } else {
                const double *xp = X + ((size_t)(t - 1) * N + i) * D;
                for (int d = 0; d < D; ++d) r += (x[d] - xp[d]) * (x[d] - xp[d]);
                s += 0.5 * r / sigma_sq;
            }
        }
        lp -= s;
    }
    for (int k = 0; k < n_intercept; ++k) {
        double df = intercept[k] - intercept_prior[k];
        lp -= 0.5 * (df * df) / intercept_var;
    }
    return lp;
}

double orc_lsm_iteration_undirected(orc_chain *c, orc_scalar_sampler *is,
                                    double intercept_prior, double intercept_var) {
    orc_sweep_positions(c);
    orc_center(c->X, c->T, c->N, c->D);
    /* sample_coefficients.py:76-86 : scalar RW-MH on the intercept */
    double u2[2], z2[2];
    orc_philox_uniform2(c->seed, 0, 0, c->iter,
                        stream_word(c->chain, ORC_STREAM_INTERCEPT), u2);
    box_muller(u2, z2);
    double b0 = c->intercept[0];
    double b1 = b0 + is->step_size * z2[0];
    double ll1 = orc_loglik_undirected(c->Y, c->X, c->T, c->N, c->D, b1, c->squared);
    double ll0 = orc_loglik_undirected(c->Y, c->X, c->T, c->N, c->D, b0, c->squared);
    double lp1 = ll1 - (b1 - intercept_prior) * (b1 - intercept_prior) / (2 * intercept_var);
    double lp0 = ll0 - (b0 - intercept_prior) * (b0 - intercept_prior) / (2 * intercept_var);
    orc_philox_uniform2(c->seed, 0, 1, c->iter,
                        stream_word(c->chain, ORC_STREAM_INTERCEPT), u2);
    int accepted = !(log(u2[0]) >= lp1 - lp0);
    double ll = accepted ? ll1 : ll0;
    if (accepted) c->intercept[0] = b1;
    metropolis_bookkeeping(&is->step_size, &is->n_accepted, &is->n_steps,
                           &is->steps_until_tune, is->tune, is->tune_interval,
                           accepted);
    /* lsm.py:576-625 */
    double b0p = intercept_prior;
    return ll + orc_lsm_log_prior(c->X, c->T, c->N, c->D, c->tau_sq, c->sigma_sq,
                                  c->intercept, 1, &b0p, intercept_var);
}
