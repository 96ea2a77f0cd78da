"""ctypes binding of include/dynetlsm_hip.h.

There is no fallback: if the HIP library is missing or no gfx950 device is
usable, every entry point raises.
"""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
# (DLSM_LIB: another build of the same sources - a profiling build with in-kernel stamps, an A / B
# experiment under tmp_timing/ - for the measurement scripts; the product loads the in-tree library)
LIB_PATH = os.environ.get('DLSM_LIB') or os.path.join(HERE, 'libdynetlsm_hip.so')

c_double_p = C.POINTER(C.c_double)
c_i64_p = C.POINTER(C.c_int64)
c_i32_p = C.POINTER(C.c_int32)
handle_t = C.c_void_p

(K_LOGLIK, K_SWEEP, K_CENTER, K_LABELS, K_FINALIZE, K_SWEEP_EVAL, K_SWEEP_RESOLVE,
 K_INIT, K_HDP_TAIL) = range(9)
UNDIRECTED, DIRECTED, DIRECTED_CASE_CONTROL = 0, 1, 2


class LsmConfig(C.Structure):
    """mirror of ``dlsm_lsm_config``"""
    _fields_ = [('intercept_prior', C.c_double * 2),
                ('intercept_variance_prior', C.c_double),
                ('i_step_size', C.c_double * 2),
                ('i_n_accepted', C.c_int32 * 2), ('i_n_steps', C.c_int32 * 2),
                ('i_steps_until_tune', C.c_int32 * 2),
                ('i_tune', C.c_int32), ('i_tune_interval', C.c_int32),
                ('n_iter_procrustes', C.c_int32), ('sweep_algo', C.c_int32),
                ('r_step_size', C.c_double), ('r_n_accepted', C.c_int32),
                ('r_n_steps', C.c_int32), ('r_steps_until_tune', C.c_int32),
                ('r_tune', C.c_int32), ('r_tune_interval', C.c_int32), ('r_pad', C.c_int32)]


class HdpConfig(C.Structure):
    """mirror of ``dlsm_hdp_config``"""
    _fields_ = [('gamma', C.c_double), ('alpha_init', C.c_double), ('alpha', C.c_double),
                ('kappa', C.c_double), ('mean_variance_prior', C.c_double), ('b', C.c_double),
                ('a', C.c_double), ('a0', C.c_double), ('b0', C.c_double), ('c0', C.c_double),
                ('d0', C.c_double), ('has_a0', C.c_int32), ('has_c0', C.c_int32),
                ('lambda_prior', C.c_double), ('lambda_variance_prior', C.c_double),
                ('gamma_prior_shape', C.c_double), ('gamma_prior_rate', C.c_double),
                ('alpha_init_shape', C.c_double), ('alpha_init_rate', C.c_double),
                ('alpha_kappa_shape', C.c_double), ('alpha_kappa_rate', C.c_double),
                ('intercept_prior', C.c_double), ('intercept_variance_prior', C.c_double),
                ('i_step_size', C.c_double), ('i_n_accepted', C.c_int32),
                ('i_n_steps', C.c_int32), ('i_steps_until_tune', C.c_int32),
                ('i_tune', C.c_int32), ('i_tune_interval', C.c_int32),
                ('sweep_algo', C.c_int32),
                # directed models: intercept_out and the radii sampler
                ('intercept_prior_out', C.c_double), ('i_step_size_out', C.c_double),
                ('i_n_accepted_out', C.c_int32), ('i_n_steps_out', C.c_int32),
                ('i_steps_until_tune_out', C.c_int32), ('r_tune', C.c_int32),
                ('r_step_size', C.c_double), ('r_n_accepted', C.c_int32), ('r_n_steps', C.c_int32),
                ('r_steps_until_tune', C.c_int32), ('r_tune_interval', C.c_int32)]


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, 'dynetlsm_hip error %d: %s' % (code, msg))
        self.code = code


# every exported symbol of include/dynetlsm_hip.h : (restype, argtypes)
SIGNATURES = {
    'dlsm_abi_version': (C.c_int, []),
    'dlsm_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'dlsm_last_error': (C.c_char_p, [handle_t]),
    'dlsm_create': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                              C.c_uint64, C.c_uint32, C.POINTER(handle_t)]),
    'dlsm_destroy': (None, [handle_t]),
    'dlsm_synchronize': (C.c_int, [handle_t]),
    'dlsm_upload_network': (C.c_int, [handle_t, c_double_p]),
    'dlsm_network_packed_words': (C.c_int, [handle_t, c_i64_p]),
    'dlsm_get_network_packed': (C.c_int, [handle_t, C.c_void_p, C.c_int64]),
    'dlsm_set_network_packed': (C.c_int, [handle_t, C.c_void_p, C.c_int64]),
    'dlsm_upload_edges': (C.c_int, [handle_t, c_i64_p, C.c_int, c_i64_p, C.c_int,
                                    c_i64_p]),
    'dlsm_set_controls': (C.c_int, [handle_t, c_i64_p, c_i64_p, C.c_int]),
    'dlsm_get_controls': (C.c_int, [handle_t, c_i64_p, c_i64_p]),
    'dlsm_resample_controls': (C.c_int, [handle_t, C.c_uint32, C.c_int]),
    'dlsm_set_positions': (C.c_int, [handle_t, c_double_p]),
    'dlsm_get_positions': (C.c_int, [handle_t, c_double_p]),
    'dlsm_set_intercepts': (C.c_int, [handle_t, c_double_p, C.c_int]),
    'dlsm_get_intercepts': (C.c_int, [handle_t, c_double_p, C.c_int]),
    'dlsm_set_radii': (C.c_int, [handle_t, c_double_p]),
    'dlsm_get_radii': (C.c_int, [handle_t, c_double_p]),
    'dlsm_set_squared': (C.c_int, [handle_t, C.c_int]),
    'dlsm_set_samplers': (C.c_int, [handle_t, c_double_p, c_i32_p, c_i32_p, c_i32_p,
                                    C.c_int, C.c_int]),
    'dlsm_get_samplers': (C.c_int, [handle_t, c_double_p, c_i32_p, c_i32_p, c_i32_p]),
    'dlsm_set_prior_random_walk': (C.c_int, [handle_t, C.c_double, C.c_double]),
    'dlsm_set_prior_mixture': (C.c_int, [handle_t, c_double_p, c_double_p, C.c_double,
                                         c_i64_p, C.c_int]),
    'dlsm_loglik_full': (C.c_int, [handle_t, C.c_int, c_double_p, c_double_p]),
    'dlsm_loglik_full_radii': (C.c_int, [handle_t, c_double_p, c_double_p]),
    'dlsm_loglik_partial': (C.c_int, [handle_t, C.c_int, C.c_int, c_double_p, C.c_int,
                                      c_double_p]),
    'dlsm_loglik_partial_all': (C.c_int, [handle_t, C.c_int, c_double_p]),
    'dlsm_sweep_positions': (C.c_int, [handle_t, C.c_uint32, C.c_int]),
    'dlsm_resolve_sweep_algo': (C.c_int, [handle_t, C.c_int]),
    'dlsm_center': (C.c_int, [handle_t]),
    'dlsm_procrustes': (C.c_int, [handle_t, c_double_p, c_double_p]),
    'dlsm_gaussian_likelihood': (C.c_int, [handle_t, C.c_int, C.c_int, c_double_p]),
    'dlsm_sample_labels': (C.c_int, [handle_t, C.c_uint32, c_double_p, c_i64_p,
                                     c_double_p, c_i64_p]),
    'dlsm_hdp_label_sums': (C.c_int, [handle_t, C.c_int, c_double_p, c_double_p, C.c_double,
                                      c_double_p, C.c_double, C.c_double, c_double_p]),
    'dlsm_lsm_configure': (C.c_int, [handle_t, C.POINTER(LsmConfig)]),
    'dlsm_lsm_get_config': (C.c_int, [handle_t, C.POINTER(LsmConfig)]),
    'dlsm_trace_alloc': (C.c_int, [handle_t, C.c_int, C.c_double]),
    'dlsm_lsm_run': (C.c_int, [handle_t, C.c_int, C.c_int, C.c_int]),
    'dlsm_trace_read': (C.c_int, [handle_t, C.c_int, C.c_int, c_double_p, c_double_p,
                                  c_double_p]),
    'dlsm_trace_read_radii': (C.c_int, [handle_t, C.c_int, C.c_int, c_double_p]),
    'dlsm_hdp_configure': (C.c_int, [handle_t, C.POINTER(HdpConfig), c_double_p, c_double_p]),
    'dlsm_hdp_get_config': (C.c_int, [handle_t, C.POINTER(HdpConfig)]),
    'dlsm_hdp_trace_alloc': (C.c_int, [handle_t, C.c_int, C.c_double]),
    'dlsm_hdp_run': (C.c_int, [handle_t, C.c_int, C.c_int]),
    'dlsm_hdp_trace_read': (C.c_int, [handle_t, C.c_int, C.c_int, c_double_p, c_double_p,
                                      c_double_p, c_double_p, c_double_p, c_i64_p, c_double_p,
                                      c_double_p, c_double_p, c_double_p]),
    'dlsm_hdp_trace_write': (C.c_int, [handle_t, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p,
                                       c_double_p, c_double_p, c_i64_p, c_double_p, c_double_p,
                                       c_double_p]),
    'dlsm_hdp_queues': (C.c_int, [handle_t, C.POINTER(C.c_int)]),
    'dlsm_hdp_get_aux': (C.c_int, [handle_t, c_i64_p, c_double_p, c_i64_p, c_i64_p, c_i64_p]),
    'dlsm_init_shortest_paths': (C.c_int, [handle_t]),
    'dlsm_init_get_dissimilarity': (C.c_int, [handle_t, C.c_int, c_double_p]),
    'dlsm_init_smacof': (C.c_int, [handle_t, C.c_int, C.c_int, c_double_p, C.c_int,
                                   C.c_double, c_double_p, c_double_p, c_i32_p]),
    'dlsm_init_gmds_step': (C.c_int, [handle_t, C.c_int, c_double_p, C.c_double, C.c_int,
                                      C.c_double, c_double_p, c_double_p, c_i32_p,
                                      c_double_p]),
    'dlsm_init_mle_sums': (C.c_int, [handle_t, C.c_double, C.c_double, c_double_p]),
    'dlsm_init_kmeans_lloyd': (C.c_int, [handle_t, c_double_p, C.c_int, C.c_int, C.c_int, c_double_p,
                                         C.c_int, C.c_double, c_double_p, c_i32_p, c_i32_p, c_i32_p]),
    'dlsm_init_release': (C.c_int, [handle_t]),
    'dlsm_post_cooccurrence': (C.c_int, [handle_t, c_i64_p, C.c_int, C.c_int, c_double_p]),
    'dlsm_post_expected_vi_sums': (C.c_int, [handle_t, c_double_p]),
    'dlsm_post_release': (C.c_int, [handle_t]),
    'dlsm_post_trace_label_counts': (C.c_int, [handle_t, C.c_int, C.c_int, c_i32_p]),
    'dlsm_post_trace_cooccurrence': (C.c_int, [handle_t, C.c_int, C.c_int, c_double_p, c_double_p]),
    'dlsm_post_get_cooccurrence': (C.c_int, [handle_t, c_double_p]),
    'dlsm_post_trace_align': (C.c_int, [handle_t, C.c_int, C.c_int, C.c_int]),
    'dlsm_post_trace_mean': (C.c_int, [handle_t, C.c_int, C.c_int, c_double_p]),
    'dlsm_post_latent_marginal_loglik': (C.c_int, [handle_t, C.c_int, c_double_p, c_double_p,
                                                   c_double_p, c_double_p, C.c_double, C.c_int,
                                                   c_double_p]),
    'dlsm_forecast_mean_probas': (C.c_int, [handle_t, c_double_p, c_double_p, C.c_int, C.c_int,
                                            c_double_p]),
    'dlsm_forecast_marginal': (C.c_int, [handle_t, c_double_p, c_double_p, c_double_p, C.c_int,
                                         c_double_p]),
    'dlsm_host_sample_tables': (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_double_p, c_double_p,
                                          C.c_double, C.c_double, C.c_double, c_i64_p]),
    'dlsm_profile_enable': (C.c_int, [handle_t, C.c_int]),
    'dlsm_profile_read': (C.c_int, [handle_t, C.c_int, c_double_p, C.POINTER(C.c_int)]),
    'dlsm_profile_read_eval_stamps': (C.c_int, [handle_t, c_double_p, C.POINTER(C.c_int)]),
    'dlsm_timer_start': (C.c_int, [handle_t]),
    'dlsm_timer_stop': (C.c_int, [handle_t, c_double_p]),
}

_LIB = None


def load():
    """Load the engine; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                'dynetlsm_amd: %s is missing. Build it with '
                '`python -m dynetlsm_amd.build` (needs hipcc); there is no CPU '
                'fallback.' % LIB_PATH)
        # PyTorch-ROCm wheels carry their own libamdhip64 / libhsa-runtime64 under the same
        # sonames as /opt/rocm's, and a process gets ONE copy: the first one loaded.  When the
        # engine's library came first, torch found no GPU afterwards (measured on the MI355X
        # box), so if torch is already imported its runtime is initialised before ours binds.
        # (Processes that use both should import torch first; without torch the engine uses
        # the system runtime.)
        if 'torch' in sys.modules:
            torch = sys.modules['torch']
            try:
                if torch.cuda.is_available():
                    torch.cuda.init()
            except Exception:       # a torch without a usable device must not mask our own error
                pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.dlsm_abi_version() != 1:
            raise ImportError('dynetlsm_amd: ABI version mismatch')
        _LIB = L
    return _LIB


def device_count():
    n = C.c_int(0)
    load().dlsm_device_count(C.byref(n))
    return n.value
