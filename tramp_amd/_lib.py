"""ctypes binding of libtramp_hip.so (the C ABI declared in include/tramp_hip.h).

There is no CPU fallback: if the HIP library is missing or a call fails, the
caller gets an exception.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRAMP_HIP_LIB") or os.path.join(_HERE, "libtramp_hip.so")      # TRAMP_HIP_LIB: an experimental build

OK, EINVAL, EHIP, ENOMEM, ENOTSUP, ENODEV = range(6)
F32, F64 = 0, 1
SGN, ABS, RELU = 0, 1, 2
FWD, BWD = 0, 1
GEMV_MAX_BATCH = 8
REUSE_T, REUSE_U = 1, 2
(DIAG_GB, DIAG_L1, DIAG_SGN_LIK, DIAG_ABS_LIK, DIAG_GAUSS_PRIOR, DIAG_GAUSS_LIK, DIAG_BINARY, DIAG_RELU_LIK,
 DIAG_PWL3_LIK, DIAG_HALFLINE, DIAG_GM) = range(11)
(LOGZ_GB_PRIOR, LOGZ_GAUSS_PRIOR, LOGZ_L1_PRIOR, LOGZ_GAUSS_LIK, LOGZ_SGN_LIK, LOGZ_ABS_LIK, LOGZ_PWL_SGN,
 LOGZ_PWL_ABS, LOGZ_PWL_RELU, LOGZ_VARIABLE) = range(10)

_STATUS = {EINVAL: "invalid argument", EHIP: "HIP runtime error", ENOMEM: "out of device memory",
           ENOTSUP: "not supported", ENODEV: "no MI355X / HIP device"}


class TrampHipError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__(f"{where}: {_STATUS.get(code, code)} {detail}".strip())


_lib = None

vp, sz, dbl, i32 = C.c_void_p, C.c_size_t, C.c_double, C.c_int
u64 = C.c_ulonglong

# name -> argtypes (restype is always int unless listed in _RESTYPES)
_SIGNATURES = {
    "tramp_device_count": [C.POINTER(i32)],
    "tramp_ctx_create": [i32, C.POINTER(vp)],
    "tramp_ctx_destroy": [vp],
    "tramp_ctx_make_current": [vp],
    "tramp_sync": [vp],
    "tramp_ctx_stream": [vp, C.POINTER(vp)],
    "tramp_device_name": [vp, C.c_char_p, sz],
    "tramp_mem_info": [vp, C.POINTER(sz), C.POINTER(sz)],
    "tramp_buf_alloc": [vp, sz, C.POINTER(vp)],
    "tramp_buf_free": [vp, vp],
    "tramp_h2d": [vp, vp, vp, sz],
    "tramp_d2h": [vp, vp, vp, sz],
    "tramp_d2d": [vp, vp, vp, sz],
    "tramp_memset": [vp, vp, i32, sz],
    "tramp_fill": [vp, i32, vp, sz, dbl],
    "tramp_fill_f64": [vp, vp, sz, dbl],
    "tramp_convert": [vp, i32, vp, i32, vp, sz],
    "tramp_event_create": [vp, C.POINTER(vp)],
    "tramp_event_destroy": [vp, vp],
    "tramp_event_record": [vp, vp],
    "tramp_event_elapsed_ms": [vp, vp, vp, C.POINTER(C.c_float)],
    "tramp_gb_prior_fwd": [vp, i32, sz, sz, vp, vp, dbl, dbl, dbl, dbl, vp, vp, vp, vp, vp, vp],
    "tramp_l1_prior_fwd": [vp, i32, sz, sz, vp, vp, dbl, dbl, vp, vp, vp, vp, vp, vp, vp, vp],
    "tramp_gauss_prior_fwd": [vp, i32, sz, sz, dbl, dbl, vp, vp],
    "tramp_gauss_lik_bwd": [vp, i32, sz, sz, vp, dbl, vp, vp],
    "tramp_sgn_lik_bwd": [vp, i32, sz, sz, vp, vp, vp, dbl, vp, vp, vp, vp, vp, vp],
    "tramp_sgn_pack": [vp, i32, sz, sz, vp, vp, C.POINTER(u64)],
    "tramp_sgn_lik_bwd_bits": [vp, i32, sz, sz, vp, vp, vp, vp, dbl, vp, vp, vp, vp, vp, vp],
    "tramp_abs_lik_bwd": [vp, i32, sz, sz, vp, vp, vp, dbl, vp, vp, vp, vp, vp, vp],
    "tramp_relu_lik_bwd": [vp, i32, sz, sz, dbl, vp, vp, vp, dbl, vp, vp, vp, vp, vp, vp],
    "tramp_se_relu_lik": [vp, sz, vp, vp, vp],
    "tramp_pwl_channel": [vp, i32, i32, i32, sz, sz, vp, vp, vp, vp, dbl, vp, vp, vp, vp, vp, vp, vp],
    "tramp_pwl2_channel": [vp, i32, i32, sz, sz, dbl, dbl, dbl, dbl, dbl, vp, vp, vp, vp, dbl, vp, vp, vp, vp, vp, vp, vp],
    "tramp_se_pwl2_channel": [vp, i32, sz, dbl, dbl, dbl, dbl, dbl, vp, vp, vp, vp],
    "tramp_pwl3_channel": [vp, i32, i32, sz, sz, dbl, dbl, vp, vp, vp, vp, vp, vp, dbl, vp, vp, vp, vp, vp, vp, vp],
    "tramp_se_pwl3_channel": [vp, i32, sz, dbl, dbl, vp, vp, vp, vp, vp, vp],
    "tramp_pwl3_lik_bwd": [vp, i32, sz, sz, dbl, dbl, vp, vp, vp, vp, vp, dbl, vp, vp, vp, vp, vp, vp],
    "tramp_se_pwl3_lik": [vp, sz, dbl, dbl, vp, vp, vp, vp, vp],
    "tramp_diag_factor": [vp, i32, i32, sz, vp, vp, vp, dbl, dbl, dbl, vp, vp, vp, vp],
    "tramp_diag_factor_p": [vp, i32, i32, sz, vp, vp, vp, vp, i32, vp, vp, vp, vp],
    "tramp_linear_create": [vp, i32, sz, sz, sz, vp, vp, vp, C.POINTER(vp)],
    "tramp_linear_destroy": [vp, vp],
    "tramp_linear_factorize": [vp, i32, sz, sz, vp, vp, vp, vp, C.POINTER(sz)],
    "tramp_eigh": [vp, sz, vp, vp],
    "tramp_factorize_peak_bytes": [vp, C.POINTER(sz)],
    "tramp_prior_hint": [vp, i32, dbl, dbl, dbl, sz, sz, vp, vp],
    "tramp_lik_hint": [vp, i32, sz, sz, vp, vp, vp],
    "tramp_prior_hint_stats": [vp, C.POINTER(u64), C.POINTER(u64)],
    "tramp_region_log_partition": [vp, i32, i32, i32, vp, sz, sz, vp, vp, vp, vp, vp],
    "tramp_resident_stats": [vp, C.POINTER(u64)],
    "tramp_resident_errors": [vp, C.POINTER(u64)],
    "tramp_coop_stats": [vp, C.POINTER(u64)],
    "tramp_resident_covers": [vp, i32, sz, sz, C.POINTER(i32)],
    "tramp_linear_variance": [vp, vp, i32, sz, vp, vp, vp, vp, vp],
    "tramp_linear_message": [vp, vp, i32, sz, vp, vp, vp, vp, dbl, vp, vp, vp, vp, vp, vp, i32],
    "tramp_conv_create": [vp, i32, i32, C.POINTER(sz), vp, C.POINTER(vp)],
    "tramp_conv_create_complex": [vp, i32, i32, C.POINTER(sz), vp, C.POINTER(vp)],
    "tramp_conv_destroy": [vp, vp],
    "tramp_conv_spectrum": [vp, vp, vp, C.POINTER(sz), C.POINTER(dbl)],
    "tramp_conv_variance": [vp, vp, i32, sz, vp, vp, vp, vp, vp],
    "tramp_conv_apply": [vp, vp, sz, vp, vp],
    "tramp_conv_message": [vp, vp, i32, sz, vp, vp, vp, vp, dbl, vp, vp, vp, vp, vp, vp, i32],
    "tramp_grad_create": [vp, i32, i32, C.POINTER(sz), vp, C.POINTER(vp)],
    "tramp_grad_destroy": [vp, vp],
    "tramp_grad_spectrum": [vp, vp, vp, C.POINTER(sz), C.POINTER(dbl)],
    "tramp_grad_variance": [vp, vp, i32, sz, vp, vp, vp, vp, vp],
    "tramp_grad_apply": [vp, vp, sz, vp, vp],
    "tramp_grad_message": [vp, vp, i32, sz, vp, vp, vp, vp, dbl, vp, vp, vp, vp, vp, vp, i32],
    "tramp_linear_set_rotated": [vp, vp, i32],
    "tramp_lsolve_create": [vp, i32, sz, sz, vp, C.POINTER(vp)],
    "tramp_lsolve_destroy": [vp, vp],
    "tramp_lsolve_spectrum": [vp, vp, vp],
    "tramp_lsolve_set_rank": [vp, vp, sz],
    "tramp_lsolve_message": [vp, vp, i32, sz, vp, vp, vp, vp, dbl, vp, vp, vp, vp, vp, vp],
    "tramp_lsolve_log_partition": [vp, vp, sz, vp, vp, vp, vp, vp],
    "tramp_linear_project_affine": [vp, vp, i32, sz, vp, dbl, i32],
    "tramp_conv_project_affine": [vp, vp, i32, sz, vp, dbl, i32],
    "tramp_dft_create": [vp, i32, i32, C.POINTER(sz), C.POINTER(vp)],
    "tramp_dft_destroy": [vp, vp],
    "tramp_dft_set_complex_input": [vp, vp, i32],
    "tramp_dft_message": [vp, vp, i32, sz, vp, vp, dbl, vp, vp, vp, vp],
    "tramp_log_partition": [vp, i32, i32, sz, sz, dbl, dbl, dbl, vp, vp, vp, vp, vp],
    "tramp_linear_log_partition": [vp, vp, sz, vp, vp, vp, vp, i32, vp],
    "tramp_fft_len": [vp, i32, sz, sz, i32, vp, vp],
    "tramp_fft_order": [vp, sz, sz, vp],
    "tramp_fft_bench": [vp, i32, sz, sz, i32, vp, vp, C.POINTER(dbl), C.POINTER(dbl)],
    "tramp_fftn": [vp, i32, i32, C.POINTER(sz), sz, i32, vp, vp],
    "tramp_fftn_order": [vp, i32, i32, C.POINTER(sz), i32, vp],
    "tramp_fftn_supported": [i32, i32, C.POINTER(sz)],
    "tramp_matvec_nt": [vp, i32, i32, sz, sz, sz, vp, vp, vp],
    "tramp_matvec_nn": [vp, i32, i32, sz, sz, sz, vp, vp, vp],
    "tramp_transpose": [vp, i32, sz, sz, vp, vp],
    "tramp_matvec_last_launch": [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)],
    "tramp_matrix_attach_digits": [vp, i32, sz, sz, vp, i32, i32],
    "tramp_matrix_detach_digits": [vp, vp],
    "tramp_matrix_digits_info": [vp, vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(sz)],
    "tramp_sliced_stats": [vp, C.POINTER(u64)],
    "tramp_variable_posterior": [vp, i32, sz, sz, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "tramp_damp": [vp, i32, sz, sz, dbl, vp, vp, vp, vp],
    "tramp_damp_to": [vp, i32, sz, sz, dbl, vp, vp, vp, vp, vp, vp],
    "tramp_binary_prior_fwd": [vp, i32, sz, sz, vp, vp, dbl, dbl, vp, vp, vp, vp, vp, vp],
    "tramp_halfline_prior_fwd": [vp, i32, sz, sz, vp, vp, dbl, dbl, dbl, vp, vp, vp, vp, vp, vp],
    "tramp_gm_prior_fwd": [vp, i32, sz, sz, vp, vp, i32, vp, vp, vp, dbl, vp, vp, vp, vp, vp, vp],
    "tramp_modulus_lik_bwd": [vp, i32, sz, sz, vp, vp, vp, dbl, vp, vp, vp, vp, vp, vp],
    "tramp_se_gm_prior": [vp, sz, vp, i32, vp, vp, vp, vp],
    "tramp_prior2_log_partition": [vp, i32, i32, sz, sz, vp, i32, vp, vp, vp],
    "tramp_modulus_log_partition": [vp, i32, sz, sz, vp, vp, vp, vp],
    "tramp_l21_prior_fwd": [vp, i32, i32, sz, sz, dbl, vp, vp, dbl, vp, vp, vp, vp, vp, vp, vp, vp],
    "tramp_gauss_channel": [vp, i32, sz, sz, dbl, vp, vp, dbl, vp, vp, vp, vp],
    "tramp_bias_channel": [vp, i32, i32, sz, sz, vp, vp, vp, dbl, vp, vp, vp, vp],
    "tramp_sum_channel": [vp, i32, sz, sz, i32, C.POINTER(vp), C.POINTER(vp), vp, vp, i32, dbl, vp, vp, vp, vp],
    "tramp_duplicate_channel": [vp, i32, sz, sz, i32, vp, vp, C.POINTER(vp), C.POINTER(vp), i32, dbl, vp, vp, vp, vp],
    "tramp_message_sum": [vp, i32, sz, sz, i32, C.POINTER(vp), C.POINTER(vp), i32, vp, vp],
    "tramp_sq_dist": [vp, i32, sz, sz, vp, vp, vp],
    "tramp_se_gb_prior": [vp, sz, vp, vp, vp, vp, vp],
    "tramp_se_last_evals": [vp, C.POINTER(u64)],
    "tramp_se_binary_prior": [vp, sz, vp, vp, vp],
    "tramp_se_mp_channel": [vp, i32, sz, vp, vp, vp, vp, vp, vp],
    "tramp_se_sgn_lik": [vp, sz, vp, vp, vp],
    "tramp_se_abs_lik": [vp, sz, vp, vp, vp],
    "tramp_se_modulus_lik": [vp, sz, vp, vp, vp],
    "tramp_se_pwl_channel": [vp, i32, i32, sz, vp, vp, vp, vp],
    "tramp_randn": [vp, i32, vp, sz, u64, dbl],
    "tramp_ep_monitor": [vp, i32, i32, sz, vp, vp, dbl, dbl, i32, vp, vp, vp, vp, vp],
    "tramp_ep_restore": [vp, i32, sz, sz, vp, vp, vp, i32],
    "tramp_ep_restore_a": [vp, i32, sz, vp, vp, vp, i32],
    "tramp_ep_freeze": [vp, i32, sz, sz, vp, vp, vp, vp],
    "tramp_comm_available": [vp],
    "tramp_comm_unique_id": [vp, vp],
    "tramp_comm_init": [vp, i32, i32, vp, C.POINTER(vp)],
    "tramp_comm_destroy": [vp, vp],
    "tramp_comm_rank": [vp, C.POINTER(i32), C.POINTER(i32)],
    "tramp_bcast_matrix": [vp, vp, vp, sz, i32],
    "tramp_gather_scalars": [vp, vp, vp, sz, vp],
    "tramp_comm_max": [vp, vp, C.POINTER(dbl)],
    "tramp_comm_barrier": [vp, vp],
    "tramp_graph_begin": [vp],
    "tramp_graph_end": [vp, C.POINTER(vp)],
    "tramp_graph_launch": [vp, vp],
    "tramp_graph_destroy": [vp, vp],
    "tramp_profile_enable": [vp, i32],
    "tramp_profile_report": [vp, vp, vp, vp, vp],
}
_RESTYPES = {"tramp_version": C.c_char_p, "tramp_last_error": C.c_char_p}
EXPORTS = sorted(list(_SIGNATURES) + list(_RESTYPES))


def load():
    """Load libtramp_hip.so once; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C tramp_amd/csrc). "
            "tramp_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.tramp_version.argtypes = []
    lib.tramp_version.restype = C.c_char_p
    lib.tramp_last_error.argtypes = [vp]
    lib.tramp_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def check(code, where, ctx=None):
    if code == OK:
        return
    detail = ""
    if ctx:
        detail = load().tramp_last_error(ctx).decode(errors="replace")
    raise TrampHipError(code, where, detail)


def device_count():
    n = i32(0)
    load().tramp_device_count(C.byref(n))
    return n.value


def version():
    return load().tramp_version().decode()
