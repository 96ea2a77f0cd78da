// Device-resident iteration of the directed LSM (lsm.py:474-572, directed branch): the two
// intercept random-walk MH steps (sample_coefficients.py:12-75) and the radii step with its
// scaled-Dirichlet proposal (metropolis.py:57-82, sample_coefficients.py:91-121), each around
// one fused two-candidate log-likelihood pass, without a host round trip.
// Draws: Philox stream INTERCEPT, counter (which, 0 | 1, iter) for the proposal normal / the
// accept uniform; stream RADII, counter (i, 2 a | 2 a + 1, iter) for attempt a of node i's
// gamma variate (Marsaglia-Tsang) and (0xFFFFFFFF, 0, iter) for the accept uniform.
#pragma once
#include "chain.hpp"
#include "device_common.hpp"
#include "kernels_sweep.hpp"      // dir_propose_intercept, reduce_records
#include "kernels_spec_pipe.hpp"  // the sweep's proposal pass

namespace dlsm {

constexpr uint32_t STREAM_RADII = 5;

// metropolis.py:23-37 (the Dirichlet proposal's concentration is tuned the other way round)
__device__ __forceinline__ double tune_dirichlet(double step, double rate) {
    if (rate < 0.001) step *= 10.0;
    else if (rate < 0.05) step *= 2.0;
    else if (rate < 0.25) step *= 1.1;
    else if (rate > 0.95) step *= 0.1;
    else if (rate > 0.75) step *= 0.5;
    else if (rate > 0.4) step *= 0.9;
    return step;
}

// accept / reject of intercept `which` from ll = [at the proposal, at the current pair] (one thread)
__device__ __forceinline__ void dir_accept_intercept(const double *ll, LsmDeviceState *lsm,
                                                     double *__restrict__ intercept, int which,
                                                     int carried) {
    const double prop = lsm->cand[which], cur = lsm->cand[2 + which];
    const double pm = lsm->intercept_prior[which], v = lsm->intercept_var;
    const double ll_cur = carried ? lsm->ll_cur : ll[1];
    const double ratio = (ll[0] - (prop - pm) * (prop - pm) / (2 * v)) -
                         (ll_cur - (cur - pm) * (cur - pm) / (2 * v));
    const int accepted = !(lsm->logu >= ratio);
    if (accepted) intercept[which] = prop;
    lsm->ll_cur = accepted ? ll[0] : ll_cur;
    double st = lsm->i_step[which];
    int32_t na = lsm->i_nacc[which], ns = lsm->i_nsteps[which], un = lsm->i_until[which];
    metropolis_bookkeeping(st, na, ns, un, lsm->i_tune, lsm->i_tune_interval, accepted);
    lsm->i_step[which] = st; lsm->i_nacc[which] = na; lsm->i_nsteps[which] = ns;
    lsm->i_until[which] = un;
}
__global__ void k_dir_accept_intercept(const double *__restrict__ ll, LsmDeviceState *lsm,
                                       double *__restrict__ intercept, int which, int carried) {
    dir_accept_intercept(ll, lsm, intercept, which, carried);
}
// Gamma(a, 1) by Marsaglia & Tsang (2000); a < 1 through Gamma(a + 1) U^(1 / a)
__device__ __forceinline__ double philox_gamma(uint64_t seed, uint32_t chain, uint32_t i,
                                               uint32_t iter, double a) {
    const double aa = a < 1.0 ? a + 1.0 : a;
    const double d = aa - 1.0 / 3.0, cc = 1.0 / sqrt(9.0 * d);
    double out = 0.0;
    for (uint32_t att = 0; att < 4096; ++att) {
        double u0, u1, z0, z1, w0, w1;
        philox_uniform2(seed, i, 2 * att, iter, stream_word(chain, STREAM_RADII), u0, u1);
        box_muller(u0, u1, z0, z1);
        philox_uniform2(seed, i, 2 * att + 1, iter, stream_word(chain, STREAM_RADII), w0, w1);
        const double t = 1.0 + cc * z0;
        if (t <= 0.0) continue;
        const double v = t * t * t;
        const double x2 = z0 * z0;
        if (w0 < 1.0 - 0.0331 * x2 * x2 || log(w0) < 0.5 * x2 + d * (1.0 - v + log(v))) {
            out = d * v;
            if (a < 1.0) out *= pow(w1, 1.0 / a);
            break;
        }
    }
    return out;
}

constexpr int DR_THREADS = 1024;
constexpr int DP_THREADS = 256;
constexpr int DP_COLS = 7;         // A, B, C, E, sum x, sum r, zero flag

// x ~ Dirichlet(step * radii) into radii_alt and the proposal density ratio
//   dir_q = log Dir(radii | step x) - log Dir(x | step radii),
// in three steps: gamma variates (one node per thread) + per-workgroup sums; normalise +
// per-workgroup sums of the density terms; one workgroup puts them together (and repairs an
// exact zero, metropolis.py:65-69, the slow way: it practically never happens).
__device__ __forceinline__ void dir_radii_gamma_wg(const ChainView &c, const LsmDeviceState *lsm,
                                                   const double *__restrict__ radii,
                                                   double *__restrict__ radii_alt,
                                                   double *__restrict__ rec, IterRef ir, int wg) {
    __shared__ double buf[DP_THREADS / 64];
    const int i = wg * DP_THREADS + threadIdx.x;
    double g = 0.0;
    if (i < c.N) {
        g = philox_gamma(c.seed, c.chain, (uint32_t)i, ir.get(), lsm->r_step * radii[i]);
        radii_alt[i] = g;
    }
    g = block_sum_all<DP_THREADS / 64>(g, buf, threadIdx.x);
    if (threadIdx.x == 0) rec[wg] = g;
}

// xr (may be NULL): the case-control log-likelihood's gather records [T][N][RW]: the proposal goes
// into their second radius slot, so the pass at the proposed radii needs no packing launch
template <int D>
__device__ __forceinline__ void dir_radii_terms_wg(const ChainView &c, const LsmDeviceState *lsm,
                                                   const double *__restrict__ radii,
                                                   double *__restrict__ radii_alt,
                                                   const double *__restrict__ rec,
                                                   double *__restrict__ rec2, double *__restrict__ xr,
                                                   int wg, int nwg) {
    __shared__ double buf[DP_COLS][DP_THREADS / 64];
    const int tid = threadIdx.x, i = wg * DP_THREADS + tid;
    double total = 0.0;
    for (int q = 0; q < nwg; ++q) total += rec[q];                  // same order in every workgroup
    const double inv = 1.0 / total, step = lsm->r_step;
    double v[DP_COLS] = {0, 0, 0, 0, 0, 0, 0};
    if (i < c.N) {
        const double x = radii_alt[i] * inv, r = radii[i];
        radii_alt[i] = x;
        if (xr) {
            constexpr int RW = llcc_record_width(D);
            const double ix = 1.0 / x;            // (the records hold RECIPROCAL radii: k_pack_xr)
            for (int t = 0; t < c.T; ++t) xr[((size_t)t * c.N + i) * RW + D + 1] = ix;
        }
        v[0] = lgamma(step * x);
        v[1] = (step * x - 1.0) * log(r);
        v[2] = lgamma(step * r);
        v[3] = (step * r - 1.0) * log(x);
        v[4] = x; v[5] = r;
        v[6] = x == 0.0 ? 1.0 : 0.0;
    }
#pragma unroll
    for (int q = 0; q < DP_COLS; ++q) {
        const double sres = block_sum_all<DP_THREADS / 64>(v[q], buf[q], tid);
        if (tid == 0) rec2[(size_t)wg * DP_COLS + q] = sres;
    }
}

template <int D, int NT>
__device__ __forceinline__ void dir_radii_finish_wg(const ChainView &c, LsmDeviceState *lsm,
                                                    const double *__restrict__ radii,
                                                    double *__restrict__ radii_alt,
                                                    const double *__restrict__ rec2,
                                                    int nblk, double *__restrict__ xr,
                                                    IterRef ir) {
    __shared__ double buf[8][NT / 64];
    __shared__ double tot[DP_COLS];
    const int tid = threadIdx.x, N = c.N;
    const double step = lsm->r_step;
    if (tid < DP_COLS) {
        double sres = 0.0;
        for (int q = 0; q < nblk; ++q) sres += rec2[(size_t)q * DP_COLS + tid];
        tot[tid] = sres;
    }
    __syncthreads();
    double A = tot[0], B = tot[1], Cc = tot[2], E = tot[3], Sx = tot[4], Sr = tot[5];
    if (tot[6] > 0.0) {                            // an exact zero: regularise and redo the sums
        double s2 = 0.0;
        for (int i = tid; i < N; i += NT) { radii_alt[i] += 1e-5; s2 += radii_alt[i]; }
        s2 = block_sum_all<NT / 64>(s2, buf[0], tid);
        A = B = Cc = E = Sx = Sr = 0.0;
        for (int i = tid; i < N; i += NT) {
            const double x = radii_alt[i] / s2, r = radii[i];
            radii_alt[i] = x;
            if (xr) {
                constexpr int RW = llcc_record_width(D);
                const double ix = 1.0 / x;        // (the records hold RECIPROCAL radii: k_pack_xr)
                for (int t = 0; t < c.T; ++t) xr[((size_t)t * N + i) * RW + D + 1] = ix;
            }
            A += lgamma(step * x); B += (step * x - 1.0) * log(r);
            Cc += lgamma(step * r); E += (step * r - 1.0) * log(x);
            Sx += x; Sr += r;
        }
        A = block_sum_all<NT / 64>(A, buf[1], tid);
        B = block_sum_all<NT / 64>(B, buf[2], tid);
        Cc = block_sum_all<NT / 64>(Cc, buf[3], tid);
        E = block_sum_all<NT / 64>(E, buf[4], tid);
        Sx = block_sum_all<NT / 64>(Sx, buf[5], tid);
        Sr = block_sum_all<NT / 64>(Sr, buf[6], tid);
    }
    if (tid == 0) {
        lsm->dir_q = (lgamma(step * Sx) - A + B) - (lgamma(step * Sr) - Cc + E);
        double u0, u1;
        philox_uniform2(c.seed, 0xFFFFFFFFu, 0, ir.get(), stream_word(c.chain, STREAM_RADII), u0, u1);
        lsm->r_logu = log(u0);
    }
}

// The iteration's chain around a log-likelihood pass in ONE launch instead of three: workgroup 0
// does the fixed-order sum of the pass's records (k_reduce_loglik), the accept / reject of
// intercept `which`, and - next_which >= 0 - the proposal of the next intercept step.  The chip
// is idle meanwhile, and the radii proposal needs nothing the intercept steps produce: its steps
// ride along as workgroups 1 .. (DP_THREADS = 256 threads, as workgroup 0) - kind 1 the gamma
// variates, 2 the normalisation + density terms, 3 (one workgroup) the closing sums.  In the
// device loop the first two ride in the centring launches below and the first reduce / accept
// launch carries the third.
struct DirRider {
    int kind, nblk;
    const double *radii; double *radii_alt; double *rec; double *rec2; double *xr;
};
static_assert(DP_THREADS == 256, "riders share the launch of a 256-thread workgroup");
template <int D>
__global__ __launch_bounds__(256) void k_dir_reduce_accept_intercept(
    const double *__restrict__ partials, int nrec, int M, double *__restrict__ ll_out, ChainView c,
    LsmDeviceState *lsm, double *__restrict__ intercept, int which, int carried, int next_which,
    IterRef ir, DirRider rd) {
    if (blockIdx.x > 0) {
        const int wg = (int)blockIdx.x - 1;
        if (rd.kind == 1) dir_radii_gamma_wg(c, lsm, rd.radii, rd.radii_alt, rd.rec, ir, wg);
        else if (rd.kind == 2) dir_radii_terms_wg<D>(c, lsm, rd.radii, rd.radii_alt, rd.rec, rd.rec2, rd.xr, wg, rd.nblk);
        else dir_radii_finish_wg<D, 256>(c, lsm, rd.radii, rd.radii_alt, rd.rec2, rd.nblk, rd.xr, ir);
        return;
    }
    __shared__ double scratch[4 * 256];
    __shared__ double sums[8];
    reduce_records(partials, nrec, M, sums, scratch, threadIdx.x);
    if (threadIdx.x != 0) return;
    double ll[2] = {sums[0], M > 1 ? sums[1] : 0.0};
    ll_out[0] = ll[0];
    if (M > 1) ll_out[1] = ll[1];
    dir_accept_intercept(ll, lsm, intercept, which, carried);
    if (next_which >= 0) dir_propose_intercept(c, lsm, intercept, next_which, ir.get());
}

// Case-control loop (round 5): BOTH intercept steps behind ONE four-candidate pass (records' columns:
// (b_in', b_out), (b_in, b_out), (b_in', b_out'), (b_in, b_out'); dir_propose_both).  Workgroup 0 sums the
// four columns in the fixed order and runs the two accept / reject rules one after the other - the second
// with the pair the first one left: exactly the values, the draws and the bookkeeping of the two launches
// around two passes this replaces (one gather pass of ~40 us and one launch less per iteration).
// Workgroup 1: the radii proposal's closing sums (rider kind 3), as in the first of the two launches.
template <int D>
__global__ __launch_bounds__(256) void k_dir_reduce_accept_both(
    const double *__restrict__ partials, int nrec, double *__restrict__ ll_out, ChainView c,
    LsmDeviceState *lsm, double *__restrict__ intercept, IterRef ir, DirRider rd) {
    if (blockIdx.x > 0) {
        dir_radii_finish_wg<D, 256>(c, lsm, rd.radii, rd.radii_alt, rd.rec2, rd.nblk, rd.xr, ir);
        return;
    }
    __shared__ double scratch[4 * 256];
    __shared__ double sums[8];
    reduce_records(partials, nrec, 4, sums, scratch, threadIdx.x);
    if (threadIdx.x != 0) return;
    ll_out[0] = sums[0]; ll_out[1] = sums[1];
    // step 1: intercept_in, [at the proposal, at the current pair] = columns 0, 1
    const double ll01[2] = {sums[0], sums[1]};
    const double b_in_before = intercept[0];
    dir_accept_intercept(ll01, lsm, intercept, 0, 0);
    const bool acc0 = intercept[0] != b_in_before || lsm->cand[0] == b_in_before;
    // step 2: intercept_out at the intercept_in the first step left; its current value is carried (ll_cur)
    lsm->cand[0] = intercept[0]; lsm->cand[1] = lsm->cand8[5];
    lsm->cand[2] = intercept[0]; lsm->cand[3] = intercept[1];
    lsm->logu = lsm->logu2;
    const double ll23[2] = {acc0 ? sums[2] : sums[3], 0.0};
    dir_accept_intercept(ll23, lsm, intercept, 1, 1);
}

// The centring launches of the directed loops with the radii proposal riding in them: the gamma
// variates beside the sums (pass 1), normalisation + density terms beside the rotation / shift
// (pass 2: it leaves the records' second radius slot to the riders).  Neither reads what the other
// writes: the proposal depends on the current radii and its step size only.
template <int D>
__global__ __launch_bounds__(PS2_THREADS) void k_post_reduce_dir(
    ChainView c, const double *__restrict__ xref_in, int n_iter_procrustes, IterRef ir,
    double *__restrict__ rec, int nb_post, const LsmDeviceState *lsm, DirRider rd) {
    if ((int)blockIdx.x < nb_post) {
        post_reduce_wg<D>(c, xref_in, n_iter_procrustes, ir, rec, (int)blockIdx.x, nb_post);
        return;
    }
    dir_radii_gamma_wg(c, lsm, rd.radii, rd.radii_alt, rd.rec, ir, (int)blockIdx.x - nb_post);
}
template <int D>
__global__ __launch_bounds__(PS2_THREADS) void k_post_apply_dir(
    ChainView c, int has_ref, int n_iter_procrustes, int do_center,
    const double *__restrict__ rec, int nrec, LsmDeviceState *lsm, IterRef ir,
    double *__restrict__ trace_X, double *__restrict__ xr, int nb_post, DirRider rd) {
    if ((int)blockIdx.x < nb_post) {
        post_apply_wg<D>(c, has_ref, n_iter_procrustes, do_center, rec, nrec, lsm, ir, nullptr, trace_X, xr,
                         1, (int)blockIdx.x, nb_post);
        return;
    }
    dir_radii_terms_wg<D>(c, lsm, rd.radii, rd.radii_alt, rd.rec, rd.rec2, rd.xr, (int)blockIdx.x - nb_post,
                          rd.nblk);
}

// The last launch of a directed iteration.  Workgroup 0: the fixed-order sum of the records of
// the pass at the proposed radii (k_reduce_loglik's), the radii's accept / reject, the trace
// row (lsm.py:604-623).  The other workgroups, when `ride`: the next sweep's proposal pass
// (kernels_spec_pipe.hpp: nothing it reads is written here; workgroup 0 leaves its two constants).
template <int D>
__global__ __launch_bounds__(DR_THREADS) void k_dir_tail(
    const double *__restrict__ partials, int nrec, double *__restrict__ ll_out, ChainView c,
    LsmDeviceState *lsm, double *__restrict__ radii, const double *__restrict__ radii_alt,
    const double *__restrict__ intercept, double *__restrict__ trace_ic,
    double *__restrict__ trace_radii, double *__restrict__ trace_logp, IterRef ir, ProposeBuf nb,
    int ride, int hdp_mode = 0) {
    const int tid = threadIdx.x, N = c.N;
    if (blockIdx.x > 0) {
        const int fb = ((int)blockIdx.x - 1) * (DR_THREADS / 256) + (tid >> 8);
        if (fb < ((N + 255) / 256) * c.T) pipe_propose_rows<D>(c, nb, ir.get() + 1u, fb, tid & 255);
        return;
    }
    __shared__ double scratch[4 * 256];
    __shared__ double sums[8];
    reduce_records(partials, nrec, 1, sums, scratch, tid);
    const double ll0 = sums[0];                     // at the proposed radii
    const int it = (int)ir.get();
    // at the current ones: the value the intercept steps left
    const double ll_now = lsm->ll_cur;
    const int accepted = !(lsm->r_logu >= (ll0 - ll_now) + lsm->dir_q);
    double *row = trace_radii + (size_t)it * N;
    for (int i = tid; i < N; i += DR_THREADS) {
        const double r = accepted ? radii_alt[i] : radii[i];
        if (accepted) radii[i] = r;
        row[i] = r;
    }
    __syncthreads();                                // every thread has read lsm->ll_cur / logu
    if (tid == 0) {
        ll_out[0] = ll0;
        const double llf = accepted ? ll0 : ll_now;
        double st = lsm->r_step;
        int32_t na = lsm->r_nacc, ns = lsm->r_nsteps, un = lsm->r_until;
        na += accepted; ns += 1;
        if (lsm->r_tune >= 0) {                    // metropolis.py:110-136, Dirichlet rule
            if (ns < lsm->r_tune && un == 0) {
                st = tune_dirichlet(st, (double)na / (double)lsm->r_tune_interval);
                na = 0; un = lsm->r_tune_interval;
            } else {
                un -= 1;
            }
        }
        lsm->r_step = st; lsm->r_nacc = na; lsm->r_nsteps = ns; lsm->r_until = un;
        const double b0 = intercept[0], b1 = intercept[1], v = lsm->intercept_var;
        const double d0 = b0 - lsm->intercept_prior[0], d1 = b1 - lsm->intercept_prior[1];
        trace_ic[(size_t)it * 2] = b0;
        trace_ic[(size_t)it * 2 + 1] = b1;
        // (HDP-LPCM loop: the row's log-posterior is computed after the run from the trace; until
        // then the slot carries the network log-likelihood of the stored state)
        trace_logp[it] = hdp_mode ? llf : llf + lsm->prior_x - 0.5 * (d0 * d0 + d1 * d1) / v;   // lsm.py:604-623
        if (ride) pipe_propose_consts(c, nb.consts, intercept);
    }
}

}  // namespace dlsm
