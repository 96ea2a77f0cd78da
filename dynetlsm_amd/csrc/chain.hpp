// Host-side chain handle and the POD view of it that kernels receive by value.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/dynetlsm_hip.h"

namespace dlsm {

// Everything a kernel needs to know about one chain (device pointers + scalars).
struct ChainView {
    int T, N, D, model, squared;
    int W;                      // uint32 words per bit row (multiple of 4)
    const uint32_t *ybits;      // [T][N][W]  bit i of row j = Y[t, j, i]
    const uint32_t *ytbits;     // [T][N][W]  bit i of row j = Y[t, i, j] (directed)
    // undirected: the same words column-block-major, [T][W / 2][Ncm] of 8 bytes - word (i, cb)
    // = bits 64 cb .. 64 cb + 63 of row i - so that the rows of one 64-column block are contiguous
    // (the log-likelihood pass reads a tile's row words as whole lines); Ncm = N rounded up to its 128-row tiles
    const unsigned long long *ycm; int Ncm;
    // case-control (int32 on device)
    const int32_t *in_edges;  int Din;
    const int32_t *out_edges; int Dout;
    const int32_t *degree;      // [T][N][2]
    const int32_t *ctrl_in;     // [T][N][C], -1 padded
    const int32_t *ctrl_out;  int C;
    double *X;                  // [T][N][D]
    const double *intercept;    // device [2]
    const double *radii;        // device [N]
    int prior_kind;
    double tau_sq, sigma_sq;
    const double *mu; const double *sigma;
    const double *lmbda_p;      // device scalar: the blending coefficient (set_prior_mixture / the HDP loop)
    const int32_t *z; int K;
    double *step; int32_t *nacc; int32_t *nsteps; int32_t *until;
    int tune, tune_interval;
    uint64_t seed; uint32_t chain;
};

// What the proposal pass of the pipelined sweeps writes (kernels_spec_pipe.hpp): the iteration's
// last launch can carry the next sweep's pass (kernels_tail_propose.hpp)
struct LsmDeviceState;
struct ProposeBuf {
    double *prop, *consts;
    int32_t *sync; int nsync, queue0;
    LsmDeviceState *lsm_draw;   // not NULL: the pass also draws the undirected loop's intercept proposal
};

// intercept sampler + LSM bookkeeping that lives on the device
struct LsmDeviceState {
    double intercept_prior[2];
    double intercept_var;
    double i_step[2];
    int32_t i_nacc[2], i_nsteps[2], i_until[2];
    int32_t i_tune, i_tune_interval;
    double cand[4];             // candidate intercepts of the current MH step
    double prior_x;             // latent-position prior terms of logp (lsm.py:604-613)
    double logu;                // log-uniform of the intercept accept test
    uint32_t iter;              // iteration counter of the captured-graph path
    uint32_t pad_;
    // directed loop: radii sampler (Dirichlet proposal), current log-likelihood, proposal
    // density ratio of the radii step
    double r_step;
    int32_t r_nacc, r_nsteps, r_until, r_tune, r_tune_interval, r_pad_;
    double ll_cur, dir_q;
    double r_logu;              // log-uniform of the radii step's accept test (its own word: the proposal
                                // is closed while the intercept steps still use `logu`)
    // case-control loop: BOTH intercept steps around one four-candidate pass (kernels_dirloop.hpp):
    // (b_in', b_out), (b_in, b_out), (b_in', b_out'), (b_in, b_out') and the second step's log-uniform
    double cand8[8];
    double logu2;
};

// Device-resident HDP-LPCM loop (hdp_lpcm.py:823-1069): hyper-parameters the loop resamples,
// the fixed hyper-priors, the blending coefficient and the carried log-likelihood.  The
// intercept sampler lives in LsmDeviceState.
struct HdpDeviceState {
    double gamma, alpha_init, alpha, kappa, mvp, b;     // resampled (hdp_lpcm.py:957-1023)
    double a, a0, b0, c0, d0;
    int32_t has_a0, has_c0;
    double lambda_prior, lambda_var;
    double gamma_shape, gamma_rate, alpha0_shape, alpha0_rate, ak_shape, ak_rate;
    double lmbda;               // current blending coefficient (ChainView::lmbda_p points here)
    double ll;                  // network log-likelihood after the intercept step
    // totals of the auxiliary variables of the current iteration (k_hdp_globals)
    double mbar_total, mbar_positive, m00_total, m_rest_total, override_total;
};

struct ProfileSlot {
    double ms = 0.0;
    int launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

}  // namespace dlsm

struct dlsm_chain {
    int device = 0;
    int T = 0, N = 0, D = 0, model = 0, squared = 0;
    uint64_t seed = 0; uint32_t chain = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;          // second queue of the speculative sweep
    void *stage = nullptr;                  // pinned host staging for the small copies of the C-ABI
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    // HDP-LPCM loop, undirected: the intercept's likelihood pass on a queue of its own beside the label
    // update and the conjugate draws, handed over through device flags (kernels_hdploop.hpp, HdpFork)
    hipStream_t fork_stream = nullptr; hipEvent_t fork_ev = nullptr;
    int32_t *fork_flags = nullptr; int32_t fork_ticket = 0; bool fork_armed = false;
    int32_t *fork_err_host = nullptr, *fork_err_dev = nullptr;      // sticky error word (mapped host memory)
    bool fork_wait_value = false;           // the queues' waits are hipStreamWaitValue32 (else the gate kernel)
    bool ll_beside_chain = false;           // the next undirected likelihood pass runs on the second queue
    // the pipelined sweep in two pieces (HDP-LPCM loop on two queues): sweep_part 1 = its head only (the
    // proposal pass and the first, evaluate-only launch - neither reads what the conjugate draws produce);
    // head_done_for = the iteration whose head has been enqueued already (the sweep proper skips it)
    int sweep_part = 0; long head_done_for = -1;
    // network
    int W = 0;
    uint32_t *ybits = nullptr, *ytbits = nullptr;
    unsigned long long *ycm = nullptr;      // undirected: column-block-major copy of ybits
    bool have_network = false;
    int Din = 0, Dout = 0, C = 0;
    int32_t *in_edges = nullptr, *out_edges = nullptr, *degree = nullptr;
    int32_t *ctrl_in = nullptr, *ctrl_out = nullptr;
    bool have_edges = false, have_controls = false;
    // state
    double *X = nullptr, *intercept = nullptr, *radii = nullptr, *radii_alt = nullptr;
    bool have_X = false, have_radii = false;
    double *step = nullptr; int32_t *nacc = nullptr, *nsteps = nullptr, *until = nullptr;
    int tune = -1, tune_interval = 100;
    bool have_samplers = false;
    int prior_kind = 0; double tau_sq = 2.0, sigma_sq = 0.1;
    double *mu = nullptr, *sigma = nullptr; double lmbda = 0.0; int32_t *z = nullptr;
    dlsm::HdpDeviceState *hdp = nullptr;    // device; always allocated (holds lmbda)
    // device-resident HDP-LPCM loop: auxiliary buffers and traces
    double *hdp_buf = nullptr; size_t hdp_buf_cap = 0; int hdp_K = 0;
    bool hdp_configured = false;
    dlsm_hdp_config hdp_cfg{};
    double *htr_mu = nullptr, *htr_sigma = nullptr, *htr_beta = nullptr, *htr_w = nullptr,
           *htr_lambda = nullptr, *htr_hyper = nullptr;
    uint8_t *htr_z = nullptr;
    int htr_n = 0, htr_K = 0;
    int K = 0; bool have_prior = false;
    // scratch
    double *partials = nullptr; size_t partials_cap = 0;   // doubles
    double *xr = nullptr; size_t xr_cap = 0;               // packed (x, r, r') records (case-control)
    double *dsmall = nullptr;      // 128 doubles of device scratch ([64, 128): a d x d rotation, d <= 8)
    double *hsmall = nullptr;      // 64 doubles of pinned host scratch
    double *xref = nullptr;        // T*N*D (procrustes reference staging)
    int32_t *lab_n = nullptr, *lab_nk = nullptr; double *lab_w = nullptr;
    size_t lab_cap = 0;                                    // T*K*K the three were sized for
    // sweep v2 scratch
    double *spec = nullptr; size_t spec_cap = 0;
    double *pipe = nullptr; size_t pipe_cap = 0;        // pipelined sweep (algo 4) buffers
    // the proposal buffers of the sweep enqueued last (valid for a pipelined sweep only) and the
    // iteration whose proposals the previous iteration's last launch has already drawn into them
    // the centring sums riding in the pipelined sweep's last launch (k_pipe_last_ride): asked for by
    // the undirected loops before they enqueue the sweep, granted (done) by launch_sweep_pipe
    bool post_ride_want = false, post_ride_done = false;
    bool loop_draws_intercept = false;      // set by the undirected LSM loop around its sweep
    const double *post_ride_xref = nullptr;
    int post_ride_jl = -1, post_ride_par = 0, post_ride_nwg = 0;
    dlsm::ProposeBuf next_prop{}; bool next_prop_ok = false, pipe_touched = false; long prop_drawn_for = -1;
    int n_cu = 256;
    int32_t *nctrl = nullptr; size_t nctrl_cap = 0;     // valid controls per (t, i, dir)
    bool nctrl_valid = false;
    // case-control model: a node's counts, control weights and its four index lists as one row (cc_rows.hpp)
    int32_t *cc_terms = nullptr; size_t cc_terms_cap = 0; bool cc_terms_valid = false; int cc_tw = 0;
    // the likelihood pass's walking order (cc_rows.hpp, k_cc_order): entries, then a count per slice
    int32_t *cc_order = nullptr, *cc_order_cnt = nullptr; size_t cc_order_cap = 0; int cc_emax = 0;
    int32_t *cc_pos = nullptr; size_t cc_pos_cap = 0;        // where a node's row lies (cc_rows.hpp: k_cc_pos)
    unsigned long long *stamps = nullptr;               // in-kernel timestamps (profiling)
    size_t stamps_cap = 0, stamps_used = 0;             // in [start, end] pairs
    std::vector<std::pair<size_t, size_t>> stamp_launches;   // (first pair, pairs) per launch
    // initialisation pipeline: hop matrices [T][N][N] and the per-slice maximum
    uint16_t *hops = nullptr; int *hops_max = nullptr; bool have_hops = false;
    // post-loop processing: labels of the kept samples (bytes, [T][N][Spad]) and the
    // co-occurrence probabilities [T][N][N]
    uint8_t *post_zt = nullptr; double *post_cooc = nullptr; int post_S = 0, post_Spad = 0;
    // LSM device-resident chain
    dlsm::LsmDeviceState *lsm = nullptr;
    dlsm_lsm_config lsm_cfg{};
    bool lsm_configured = false;
    double *trace_X = nullptr, *trace_ic = nullptr, *trace_logp = nullptr;
    double *trace_radii = nullptr;
    int trace_n = 0;
    // one captured Gibbs iteration (hipGraph), replayed by dlsm_lsm_run
    hipGraph_t graph = nullptr; hipGraphExec_t graph_exec = nullptr;
    int graph_ref = -2, graph_algo = -1; bool graph_failed = false;
    // measurement
    bool profiling = false;
    dlsm::ProfileSlot prof[DLSM_K_COUNT];
    hipEvent_t timer0 = nullptr, timer1 = nullptr;
    std::string err;

    dlsm::ChainView view() const {
        dlsm::ChainView v;
        v.T = T; v.N = N; v.D = D; v.model = model; v.squared = squared; v.W = W;
        v.ybits = ybits; v.ytbits = ytbits;
        v.ycm = ycm; v.Ncm = (N + 127) / 128 * 128;
        v.in_edges = in_edges; v.Din = Din; v.out_edges = out_edges; v.Dout = Dout;
        v.degree = degree; v.ctrl_in = ctrl_in; v.ctrl_out = ctrl_out; v.C = C;
        v.X = X; v.intercept = intercept; v.radii = radii;
        v.prior_kind = prior_kind; v.tau_sq = tau_sq; v.sigma_sq = sigma_sq;
        v.mu = mu; v.sigma = sigma; v.lmbda_p = hdp ? &hdp->lmbda : nullptr; v.z = z; v.K = K;
        v.step = step; v.nacc = nacc; v.nsteps = nsteps; v.until = until;
        v.tune = tune; v.tune_interval = tune_interval;
        v.seed = seed; v.chain = chain;
        return v;
    }
};
