/*
 * dynetlsm_hip.h -- C-ABI of the MI355X (gfx950) engine for DynetLSM's
 * Metropolis-within-Gibbs hot path.
 *
 * The reference (joshloyal/dynetlsm v0.1.0) has no FFI: its seams are Python
 * function calls.  Each entry point below names the reference
 * call it replaces (file:line relative to the reference tree).  A chain handle
 * owns ALL device state (bit-packed network, latent positions, Metropolis
 * state, Philox key); one handle per chain per device.  A handle is not
 * thread-safe; different handles may be used from different host threads.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (DLSM_E_*);
 *     dlsm_last_error() gives the message.  No exceptions, no callbacks.
 *   - host arrays are C-contiguous little-endian float64 / int64 / int32 as
 *     stated; the callee never keeps a host pointer past return.
 *   - calls are synchronous at return (results are in the host buffers) except
 *     the *_async / run calls, which only enqueue on the handle's stream and
 *     are completed by dlsm_synchronize().
 *   - counter RNG: Philox4x32-10 keyed by `seed`, counter (index, t|draw<<16,
 *     iteration, chain<<8|stream); the CPU oracle uses the same draws.
 */
#ifndef DYNETLSM_HIP_H
#define DYNETLSM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DLSM_ABI_VERSION 1

enum {
    DLSM_OK = 0,
    DLSM_E_ARG = -1,      /* bad argument / wrong state */
    DLSM_E_HIP = -2,      /* HIP runtime error */
    DLSM_E_NODEV = -3,    /* no usable gfx950 device */
    DLSM_E_DATA = -4,     /* network entries not in {0,1} etc. */
    DLSM_E_LIMIT = -5     /* size beyond what the kernels support */
};

enum { DLSM_UNDIRECTED = 0, DLSM_DIRECTED = 1, DLSM_DIRECTED_CASE_CONTROL = 2 };
enum { DLSM_PRIOR_RANDOM_WALK = 0, DLSM_PRIOR_MIXTURE = 1 };

typedef struct dlsm_chain dlsm_chain;

int dlsm_abi_version(void);
int dlsm_device_count(int *count);
/* message of the last failing call on this handle (or of dlsm_create if NULL) */
const char *dlsm_last_error(const dlsm_chain *h);

/* ---- lifecycle ------------------------------------------------------- */
/* replaces the per-fit allocations of lsm.py:371-383,451-470 /
 * hdp_lpcm.py:691-747: one chain of a T x N x N network, D latent features */
int dlsm_create(int device, int T, int N, int D, int model, uint64_t seed,
                uint32_t chain_id, dlsm_chain **out);
void dlsm_destroy(dlsm_chain *h);
int dlsm_synchronize(dlsm_chain *h);

/* ---- network --------------------------------------------------------- */
/* Y: T*N*N float64 row-major exactly as fit(Y) receives it (lsm.py:341).
 * Packed on device to 1 bit/dyad (+ the transpose for directed models).
 * Entries other than 0.0 / 1.0 -> DLSM_E_DATA (the estimators impute -1 coded
 * dyads before the upload, as imputer.py / lsm.py:345-359 do). */
int dlsm_upload_network(dlsm_chain *h, const double *Y);
/* the network as the chain holds it (1 bit per dyad; [T][N][W] uint32 words,
 * W = dlsm_network_packed_words / (T N), followed by the transposed copy for the directed
 * model), so that one chain's upload can be broadcast to the chains on the other GPUs
 * (RCCL) without a host round trip of the float64 tensor: at T=10, N=2000 5 MB instead of
 * 320 MB.  `buf` may be a device or a host pointer (hipMemcpyDefault).  The reference has
 * no counterpart (examples/homogeneous_simulation.py:178-184 refits seeds serially). */
int dlsm_network_packed_words(dlsm_chain *h, int64_t *n_words);
int dlsm_get_network_packed(dlsm_chain *h, uint32_t *buf, int64_t n_words);
int dlsm_set_network_packed(dlsm_chain *h, const uint32_t *buf, int64_t n_words);
/* case-control: zero padded edge lists and degrees exactly as
 * DirectedCaseControlSampler.init builds them (case_control_likelihood.py:45-68):
 * in_edges T*N*Din, out_edges T*N*Dout, degree T*N*2 (col 0 in, col 1 out). */
int dlsm_upload_edges(dlsm_chain *h, const int64_t *in_edges, int Din,
                      const int64_t *out_edges, int Dout, const int64_t *degree);
/* control_nodes_in_/out_: T*N*C int64, -1 padded (case_control_likelihood.py:79-82) */
int dlsm_set_controls(dlsm_chain *h, const int64_t *ctrl_in,
                      const int64_t *ctrl_out, int C);
int dlsm_get_controls(dlsm_chain *h, int64_t *ctrl_in, int64_t *ctrl_out);
/* DirectedCaseControlSampler.sample (case_control_likelihood.py:75-112) on
 * device: for every (t, i) draw min(n_control, #zeros) distinct non-neighbours
 * != i, separately for the out and the in direction, Philox stream CONTROLS. */
int dlsm_resample_controls(dlsm_chain *h, uint32_t iter, int n_control);

/* ---- chain state (so host and device paths can be swapped mid-chain) --- */
int dlsm_set_positions(dlsm_chain *h, const double *X);        /* T*N*D */
int dlsm_get_positions(dlsm_chain *h, double *X);
int dlsm_set_intercepts(dlsm_chain *h, const double *b, int n); /* 1 or 2 */
int dlsm_get_intercepts(dlsm_chain *h, double *b, int n);
int dlsm_set_radii(dlsm_chain *h, const double *radii);         /* N */
int dlsm_get_radii(dlsm_chain *h, double *radii);
int dlsm_set_squared(dlsm_chain *h, int squared);
/* T x N grid of random-walk Metropolis samplers (metropolis.py:85-94);
 * tune < 0 means tune=None */
int dlsm_set_samplers(dlsm_chain *h, const double *step_size,
                      const int32_t *n_accepted, const int32_t *n_steps,
                      const int32_t *steps_until_tune, int tune, int tune_interval);
int dlsm_get_samplers(dlsm_chain *h, double *step_size, int32_t *n_accepted,
                      int32_t *n_steps, int32_t *steps_until_tune);
/* prior of sample_latent_positions (sample_latent_positions.py:132-140) */
int dlsm_set_prior_random_walk(dlsm_chain *h, double tau_sq, double sigma_sq);
/* prior of sample_latent_positions_mixture (:187-199): mu K*D, sigma K
 * (variances), z T*N int64; z = NULL keeps the labels already on the device (those of
 * the last dlsm_sample_labels / dlsm_set_prior_mixture with the same K) */
int dlsm_set_prior_mixture(dlsm_chain *h, const double *mu, const double *sigma,
                           double lmbda, const int64_t *z, int K);

/* ---- kernels --------------------------------------------------------- */
/* a4/a5/a6: dynamic_network_loglikelihood_undirected (network_likelihoods.py:26-33),
 * directed_network_loglikelihood_fast (directed_likelihoods_fast.pyx:185-205),
 * approx_directed_network_loglikelihood (:208-270) at the handle's current X
 * for m candidate intercepts (m*1 undirected, m*2 directed: [b_in, b_out]).
 * intercepts == NULL evaluates the handle's current intercept (m = 1). */
int dlsm_loglik_full(dlsm_chain *h, int m, const double *intercepts, double *out);
/* directed models: out[0] at the handle's radii, out[1] at radii_alt, both at
 * the current intercepts (sample_radii, sample_coefficients.py:91-121) */
int dlsm_loglik_full_radii(dlsm_chain *h, const double *radii_alt, double *out);
/* a1/a2/a3: partial_loglikelihood (static_network_fast.pyx:17-44),
 * directed_partial_loglikelihood (directed_likelihoods_fast.pyx:46-80),
 * approx_directed_partial_loglikelihood (:83-182) of node j at time t with
 * X[t, j] replaced by x (x == NULL: current position).  with_prior != 0 adds
 * the prior terms of the sweep's logp closure. */
int dlsm_loglik_partial(dlsm_chain *h, int t, int j, const double *x,
                        int with_prior, double *out);
/* the same for every (t, j) at the current state: out T*N */
int dlsm_loglik_partial_all(dlsm_chain *h, int with_prior, double *out);
/* a8-a10: one sweep of sample_latent_positions / _mixture: T*N random-walk MH
 * steps, even-t slices concurrently then odd-t slices, Gauss-Seidel inside a
 * slice; updates X and the sampler grid on device.  algo 0 = auto,
 * 1 = one workgroup per slice, 2 = speculative batches over the whole chip,
 * 3 = speculative batches with one launch pair per two 128-node sub-batches,
 * 4 = pipelined speculative batches: one fused launch resolves batch b and
 *     evaluates batch b + 1, both parities in the same launches,
 * 5 = (case-control model) the pipelined form with sparse correction lists and batches
 *     of 512 nodes.
 * (6 = two batches per launch and 7 = one persistent launch per sweep existed in rounds 2-4,
 *  measured slower than 4 and were removed in round 5: DLSM_E_ARG.) */
int dlsm_sweep_positions(dlsm_chain *h, uint32_t iter, int algo);
/* the algorithm `algo` resolves to for this handle (what 0 = auto picks): 1, 2, 4 or 5; a
 * non-zero `algo` is returned as it is after the same checks as dlsm_sweep_positions */
int dlsm_resolve_sweep_algo(dlsm_chain *h, int algo);
/* lsm.py:501 / hdp_lpcm.py:852 */
int dlsm_center(dlsm_chain *h);
/* longitudinal_procrustes_rotation (procrustes.py:28-35): X <- X R with
 * R = argmin ||X R - X_ref||_F ; X_ref T*N*D ; R_out (D*D, may be NULL) */
int dlsm_procrustes(dlsm_chain *h, const double *X_ref, double *R_out);
/* a11: compute_gaussian_likelihood(X[:, node], mu, sigma, lmbda, normalize)
 * (gaussian_likelihood_fast.pyx:30-54) with the mixture prior's parameters */
int dlsm_gaussian_likelihood(dlsm_chain *h, int node, int normalize, double *out);
/* a12: sample_labels_block (sample_labels.py:134-190); w T*K*K (w[0,0,:] the
 * initial distribution).  Writes z (T*N), n (T*K*K), nk (T*K) and keeps the new
 * z on device as the mixture prior's labels. */
int dlsm_sample_labels(dlsm_chain *h, uint32_t iter, const double *w, int64_t *z,
                       double *n, int64_t *nk);

/* the O(T N) sums of the HDP-LPCM conjugate updates over the nodes that
 * carry a label, at the handle's positions and labels (the z of the last
 * dlsm_sample_labels / dlsm_set_prior_mixture), K = the mixture prior's components:
 *   stage 0 (hdp_lpcm.py:901-921)  out T*K*D : sum_i V_ti, V_0 = X_0, V_t = X_t - (1-lmbda) X_{t-1}
 *   stage 1 (:924-938)             out T*K   : sum_i squared residuals about the NEW mu
 *   stage 2 (:941-954)             out T*K*2 : the two sums of the lambda update per (t, k)
 *   stage 3 (:1213-1262)           out T*K   : the node terms of DynamicNetworkHDPLPCM.logp
 *                                              (label transitions w, Gaussian and inverse-gamma
 *                                              terms; a, b the variance prior's parameters)
 * mu K*D, sigma K, w T*K*K are the values the update holds at that point (they are not
 * stored in the handle).  Fixed summation order: bitwise reproducible. */
int dlsm_hdp_label_sums(dlsm_chain *h, int stage, const double *mu, const double *sigma,
                        double lmbda, const double *w, double a, double b, double *out);

/* ---- device-resident LSM chain (lsm.py:474-572, fully observed network) -- */
typedef struct {
    double intercept_prior[2];
    double intercept_variance_prior;
    /* intercept samplers (metropolis.py:85-94); [0] = b or b_in, [1] = b_out */
    double i_step_size[2];
    int32_t i_n_accepted[2], i_n_steps[2], i_steps_until_tune[2];
    int32_t i_tune, i_tune_interval;           /* i_tune < 0 == None */
    int32_t n_iter_procrustes;                 /* lsm.py:362-368 */
    int32_t sweep_algo;                        /* as dlsm_sweep_positions */
    /* directed models: the radii sampler (scaled-Dirichlet proposal, metropolis.py:57-82;
     * r_tune < 0 == None, as DynamicNetworkLSM sets it) */
    double r_step_size;
    int32_t r_n_accepted, r_n_steps, r_steps_until_tune, r_tune, r_tune_interval, r_pad;
} dlsm_lsm_config;
int dlsm_lsm_configure(dlsm_chain *h, const dlsm_lsm_config *cfg);
int dlsm_lsm_get_config(dlsm_chain *h, dlsm_lsm_config *cfg);
/* device trace buffers Xs_[n_total,T,N,D], intercepts_[n_total,2],
 * logps_[n_total]; row 0 is filled from the current state, logp0 given */
int dlsm_trace_alloc(dlsm_chain *h, int n_total, double logp0);
/* enqueue iterations it = first .. first+count-1: sweep, Procrustes to row
 * `procrustes_ref` of the trace if it > n_iter_procrustes, centring, then
 *   undirected: intercept RW-MH fused with the log-posterior trace;
 *   directed / case-control: intercept_in, intercept_out (sample_coefficients.py:12-75)
 *   and radii (:91-121) MH steps, each around one fused two-candidate pass, with Philox
 *   draws (the caller resamples controls between calls at its own cadence).
 * Asynchronous. */
int dlsm_lsm_run(dlsm_chain *h, int first, int count, int procrustes_ref);
int dlsm_trace_read(dlsm_chain *h, int first, int count, double *Xs,
                    double *intercepts, double *logps);
/* rows first .. first+count-1 of the radii trace (count*N), directed models */
int dlsm_trace_read_radii(dlsm_chain *h, int first, int count, double *radii);

/* ---- device-resident HDP-LPCM chain (hdp_lpcm.py:823-1069) --------------------------
 * the whole Gibbs iteration of DynamicNetworkHDPLPCM._fit on the device -
 * sweep with the AR-mixture prior, centring, intercept MH, label block update
 * (sample_labels.py:134-190), table counts and override variables
 * (sample_auxillary.py:6-50), beta / w0 / w Dirichlet draws (hdp_lpcm.py:887-898), cluster
 * means, variances and the blending coefficient (:901-954), the variance hyper-parameters
 * (:957-972), the concentration parameters (sample_concentration.py:6-21,
 * hdp_lpcm.py:977-1023) and the log-posterior trace (:1188-1280) - with Philox draws (stream
 * HDP = 6, counter (index, kind | attempt << 8, iteration)): no host round trip inside an
 * iteration.  The mixture's state (mu, sigma, lmbda, z) is what dlsm_set_prior_mixture
 * left on the device.  Directed and case-control models: the intercept step is the two steps of
 * sample_coefficients.py:12-75 followed by the radii step (:91-121), as in dlsm_lsm_run's
 * directed loop (the caller resamples the controls between calls at its own cadence); the radii
 * trace is read with dlsm_trace_read_radii. */
typedef struct {
    /* resampled by the loop */
    double gamma, alpha_init, alpha, kappa, mean_variance_prior, b;
    /* fixed hyper-priors (hdp_lpcm.py:760-793); has_a0 / has_c0 = 0 switch the updates of
     * mean_variance_prior / b off (mean_variance_prior_std / sigma_prior_std = None) */
    double a, a0, b0, c0, d0;
    int32_t has_a0, has_c0;
    double lambda_prior, lambda_variance_prior;
    double gamma_prior_shape, gamma_prior_rate, alpha_init_shape, alpha_init_rate,
           alpha_kappa_shape, alpha_kappa_rate;
    double intercept_prior, intercept_variance_prior;
    /* the intercept's random-walk sampler (metropolis.py:85-94); i_tune < 0 == None */
    double i_step_size;
    int32_t i_n_accepted, i_n_steps, i_steps_until_tune, i_tune, i_tune_interval;
    int32_t sweep_algo;                        /* as dlsm_sweep_positions */
    /* directed models (hdp_lpcm.py:731-747, sample_coefficients.py:12-121): the fields above
     * describe intercept_in, these intercept_out and the radii sampler (scaled-Dirichlet
     * proposal, metropolis.py:57-82; r_tune < 0 == None) */
    double intercept_prior_out, i_step_size_out;
    int32_t i_n_accepted_out, i_n_steps_out, i_steps_until_tune_out, r_tune;
    double r_step_size;
    int32_t r_n_accepted, r_n_steps, r_steps_until_tune, r_tune_interval;
} dlsm_hdp_config;
/* beta K, weights T*K*K (weights[0,0,:] the initial distribution), K = the mixture prior's */
int dlsm_hdp_configure(dlsm_chain *h, const dlsm_hdp_config *cfg, const double *beta,
                       const double *weights);
/* the current hyper-parameters and sampler state */
int dlsm_hdp_get_config(dlsm_chain *h, dlsm_hdp_config *cfg);
/* device trace of n_total samples: Xs_, intercepts_, logps_ (shared with the LSM loop's
 * buffers), mus_, sigmas_, zs_, betas_, weights_, lambdas_ and the six resampled
 * hyper-parameters; row 0 = the current state, logp0 given */
int dlsm_hdp_trace_alloc(dlsm_chain *h, int n_total, double logp0);
/* enqueue iterations first .. first+count-1.  Asynchronous.
 * Undirected model: the intercept step's likelihood pass (sample_coefficients.py:76-86 inside
 * hdp_lpcm.py:855-874) - and behind it the head of the next iteration's sweep, which reads positions,
 * step sizes and the settled intercept only - runs on a second queue of the handle beside the label
 * update and the conjugate draws (hdp_lpcm.py:876-1023), which do not read its result; the queues hand over through words in
 * device memory and are joined before the call's log-posterior pass, so the call still orders like one
 * stream.  Chosen while the handle is the process's only live chain (environment: DLSM_HDP_QUEUES=1
 * never, =2 always); the trace is bit for bit the one-queue trace.  Processes that SHARE a device
 * should set DLSM_HDP_QUEUES=1 (multichain.launch_ranks does): queues of two processes taking turns on
 * one GPU lose more than the second queue hides. */
int dlsm_hdp_run(dlsm_chain *h, int first, int count);
/* queues the last dlsm_hdp_run call of this handle used: 1 or 2 */
int dlsm_hdp_queues(dlsm_chain *h, int *queues);
/* rows first .. first+count-1; any pointer may be NULL.  zs count*T*N int64, hypers count*6
 * [gamma, alpha_init, alpha, kappa, mean_variance_prior, b]; intercepts count*2 (undirected
 * model: the second column carries the network log-likelihood of the stored state) */
int dlsm_hdp_trace_read(dlsm_chain *h, int first, int count, double *Xs, double *intercepts,
                        double *logps, double *mus, double *sigmas, int64_t *zs, double *betas,
                        double *weights, double *lambdas, double *hypers);
/* the mirror of dlsm_hdp_trace_read (intercepts: count*1 for the undirected model, count*2 for
 * the directed ones): rows first ..
 * first+count-1 of the device-resident trace from host arrays; any pointer may be NULL.  Lets a
 * caller continue or post-process on the device a trace it holds on the host (the reference
 * keeps Xs_, zs_, ... as attributes of the fitted estimator, hdp_lpcm.py:691-747). */
int dlsm_hdp_trace_write(dlsm_chain *h, int first, int count, const double *Xs,
                         const double *intercepts, const double *logps, const double *mus,
                         const double *sigmas, const int64_t *zs, const double *betas,
                         const double *weights, const double *lambdas);
/* auxiliary variables of the last iteration (tests): m T*K*K tables, m_bar K, w_over (T-1)*K
 * override counts, n T*K*K and nk T*K label counts */
int dlsm_hdp_get_aux(dlsm_chain *h, int64_t *m, double *m_bar, int64_t *w_over, int64_t *n,
                     int64_t *nk);

/* ---- starting values (generalized_mds + conditional MLEs) -- */
/* shortest_path_dissimilarity (latent_space.py:36-44) of every time slice, from
 * the bit-packed network (symmetrised for directed models, as
 * csgraph.shortest_path(directed=False, unweighted=True) does): hop counts
 * [T][N][N] kept on the device as uint16, unconnected pairs = largest finite
 * hop count of the slice + 1. */
int dlsm_init_shortest_paths(dlsm_chain *h);
/* slice t of the above as float64 N*N */
int dlsm_init_get_dissimilarity(dlsm_chain *h, int t, double *out);
/* MDS(dissimilarity='precomputed').fit_transform(D[t]) (latent_space.py:66-68):
 * sklearn's metric SMACOF (_smacof_single, scikit-learn >= 1.7 convergence rule
 * (old_stress - stress) / (sum dis^2 / 2) < eps) run from n_init starting
 * configurations X0 (n_init*N*D, the caller draws them: RandomState.uniform)
 * concurrently.  Per run: final configuration, raw stress and n_iter; the caller
 * keeps the run of least stress as sklearn.manifold.smacof does. */
int dlsm_init_smacof(dlsm_chain *h, int t, int n_init, const double *X0, int max_iter,
                     double eps, double *X_out, double *stress_out, int32_t *n_iter_out);
/* one step t >= 1 of generalized_mds (latent_space.py:71-89): top-D eigenpairs of
 * alpha H(-D_t^2/2)H + beta X_prev X_prev^T (alpha = 1/(1+lmbda), beta =
 * lmbda/(1+lmbda)) by Lanczos with full reorthogonalisation (at most max_lanczos
 * vectors, stop at relative residual tol), X = V sqrt(evals), then the Procrustes
 * rotation onto X_prev (procrustes.py:20-25).  X_prev, X_out: N*D. */
int dlsm_init_gmds_step(dlsm_chain *h, int t, const double *X_prev, double lmbda,
                        int max_lanczos, double tol, double *X_out, double *evals_out,
                        int32_t *n_lanczos_out, double *resid_out);
/* objective and gradient of the conditional MLEs at the handle's positions:
 * undirected (scale_intercept_mle, lsm.py:32-70): p0 = log scale, p1 = intercept,
 *   out = [loglik, scale_grad, undirected_intercept_grad];
 * directed (directed_intercept_mle, lsm.py:73-97): p0 = b_in, p1 = b_out,
 *   out = [loglik, directed_intercept_grad in, out]
 *   (directed_likelihoods_fast.pyx:20-43). */
int dlsm_init_mle_sums(dlsm_chain *h, double p0, double p1, double *out);
/* the Lloyd iterations of longitudinal_kmeans (latent_space.py:98-137: sklearn.cluster.KMeans on
 * the N x F matrix of time-stacked trajectories, F = T d), scikit-learn's _kmeans_single_lloyd:
 * X N*F (centred by the caller, as KMeans.fit does), centers_init K*F (k-means++ seeding, drawn by
 * the caller from its RandomState), tol the absolute tolerance (KMeans' tol x mean feature
 * variance).  centers_out K*F, labels_out N, n_iter_out.  empty_out != 0: a cluster lost all its
 * members at iteration n_iter_out (scikit-learn then relocates it to the farthest sample): the
 * outputs are not set and the caller finishes with the library. */
int dlsm_init_kmeans_lloyd(dlsm_chain *h, const double *X, int N, int F, int K,
                           const double *centers_init, int max_iter, double tol, double *centers_out,
                           int32_t *labels_out, int32_t *n_iter_out, int32_t *empty_out);
/* free the hop matrices */
int dlsm_init_release(dlsm_chain *h);

/* ---- post-loop processing of the stored labels -------- */
/* _calculate_posterior_cooccurrences (hdp_lpcm.py:1180-1186, label_utils.py:40-62): zs is
 * the kept part of zs_ (zs_[n_burn:], S*T*N int64, labels < K); cooc_out (T*N*N, may be
 * NULL) = fraction of the S samples in which nodes i and j share a label at time t.  The
 * labels and the matrices stay on the device for dlsm_post_expected_vi_sums. */
int dlsm_post_cooccurrence(dlsm_chain *h, const int64_t *zs, int S, int K, double *cooc_out);
/* the sample-dependent sum of posterior_expected_vi (model_selection/posterior_vi.py:23-43):
 * out[t*S + s] = sum_i log2( sum_j cooc[t,i,j] [z_stj == z_sti] ) for every kept sample */
int dlsm_post_expected_vi_sums(dlsm_chain *h, double *out);
int dlsm_post_release(dlsm_chain *h);
/* The same processing straight from the device-resident trace of dlsm_hdp_run (rows
 * first .. first + count - 1 = the kept samples), so that fit() does not move the trace to the host
 * and back (hdp_lpcm.py:1085-1162):
 *   label_counts: nk count*T*K int32, nodes per label, time and sample - what approx_bic.py:26-51
 *     (models' sizes), posterior_vi.py:31-36 (the n_k log2 n_k term) and label_utils.py:73-81
 *     (posterior group counts) are computed from;
 *   cooccurrence: as dlsm_post_cooccurrence; cooc_out (T*N*N) and row_sums (T*N: sum_j C_t[i][j],
 *     posterior_vi.py:43) may be NULL; the matrices stay on the device for the VI sums and
 *     dlsm_post_get_cooccurrence;
 *   align: hdp_lpcm.py:1141-1146 - every row (positions and cluster means) rotated onto row
 *     `ref_row` by the orthogonal Procrustes rotation over all T N positions (procrustes.py:20-35,
 *     scipy.linalg.orthogonal_procrustes), in place on the device;
 *   mean: X_mean_ (T*N*D) = mean of the rows (hdp_lpcm.py:1149). */
int dlsm_post_trace_label_counts(dlsm_chain *h, int first, int count, int32_t *nk);
int dlsm_post_trace_cooccurrence(dlsm_chain *h, int first, int count, double *cooc_out,
                                 double *row_sums);
int dlsm_post_get_cooccurrence(dlsm_chain *h, double *cooc_out);
int dlsm_post_trace_align(dlsm_chain *h, int first, int count, int ref_row);
int dlsm_post_trace_mean(dlsm_chain *h, int first, int count, double *X_mean);
/* latent_marginal_loglikelihood (model_selection/approx_bic.py:54-76): forward algorithm over
 * the label chain of every node, at trace row `row` (row < 0: the handle's current positions);
 * init_w K, trans_w T*K*K (trans_w[0] unused), mu K*D, sigma K (variances), K <= 64 */
int dlsm_post_latent_marginal_loglik(dlsm_chain *h, int row, const double *init_w,
                                     const double *trans_w, const double *mu, const double *sigma,
                                     double lmbda, int K, double *out);

/* ---- one-step-ahead forecasts (undirected models) ------ */
/* the accumulation of forecast_probas / forecast_probas_pp_ (hdp_lpcm.py:555-626):
 * out[i,j] = (1/S) sum_s expit(intercepts[s] - |Xs[s,i] - Xs[s,j]|), Xs S*N*D the sampled
 * one-step-ahead positions (drawn by the caller: the draws keep the reference's MT19937
 * order), out N*N; zero_diag != 0 zeroes the diagonal as forecast_probas does. */
int dlsm_forecast_mean_probas(dlsm_chain *h, const double *Xs, const double *intercepts, int S,
                              int zero_diag, double *out);
/* marginal_forecast (forecast.pyx:79-128): out[i,j] = sum_s W[s,i] W[s,j]
 * expit(intercepts[s] - |x_i - x_j|) / sum_s W[s,i] W[s,j] for i != j, 0 on the diagonal;
 * x N*D the plug-in positions, W S*N the per-sample mixture densities of x_i (O(S N K), computed
 * by the caller: mixture_normal_pdf, forecast.pyx:39-54). */
int dlsm_forecast_marginal(dlsm_chain *h, const double *x, const double *W, const double *intercepts,
                           int S, double *out);

/* ---- host-stream auxiliary draws ----------------------- */
/* sample_tables (sample_auxillary.py:6-28): m T*K*K int64 = tables per (restaurant, dish)
 * given the transition counts n T*K*K (n[0,0,:] = initial counts) and beta K.  The
 * Bernoulli draws come from the CALLER's numpy RandomState, in the reference's order:
 * numpy_bitgen is the address of its bitgen_t (RandomState._bit_generator.ctypes.bit_generator).
 * Runs on the host; needs no chain.  -1: a probability outside [0, 1] (numpy raises there). */
int dlsm_host_sample_tables(void *numpy_bitgen, int T, int K, const double *n, const double *beta,
                            double alpha_init, double alpha, double kappa, int64_t *m);

/* ---- measurement ------------------------------------------------------ */
enum {
    DLSM_K_LOGLIK = 0, DLSM_K_SWEEP = 1, DLSM_K_CENTER = 2, DLSM_K_LABELS = 3,
    DLSM_K_FINALIZE = 4,
    DLSM_K_SWEEP_EVAL = 5,     /* k_spec_eval launches of the speculative sweep */
    DLSM_K_SWEEP_RESOLVE = 6,  /* k_spec_resolve launches */
    DLSM_K_INIT = 7,           /* every kernel of the dlsm_init_* calls */
    DLSM_K_HDP_TAIL = 8,       /* dlsm_hdp_run: everything after the label update */
    DLSM_K_COUNT = 9
};
/* when enabled every launch of the kernel classes above is bracketed by HIP
 * events on the handle's stream; read returns accumulated ms and launches */
int dlsm_profile_enable(dlsm_chain *h, int on);
int dlsm_profile_read(dlsm_chain *h, int kernel, double *total_ms, int *launches);
/* while profiling, the sweep's dominant kernel (k_spec_eval) also stamps the 100 MHz
 * wall clock in-kernel (min start / max end over its workgroups): mean duration in us
 * without the dispatch latency that brackets of HIP events include */
int dlsm_profile_read_eval_stamps(dlsm_chain *h, double *mean_us, int *launches);
/* wall-clock of a region on the handle's stream, by HIP events */
int dlsm_timer_start(dlsm_chain *h);
int dlsm_timer_stop(dlsm_chain *h, double *ms);

#ifdef __cplusplus
}
#endif
#endif
