"""ctypes binding of libozk_hip.so (include/ozk.h).  No CPU path: if the library is
missing or cannot be loaded, importing callers get a RuntimeError."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# OZK_LIB_PATH: another build of the same library (tools/ A/B runs of two builds inside one gpurun call)
LIB_PATH = os.environ.get("OZK_LIB_PATH") or os.path.join(HERE, "libozk_hip.so")

_lib = None

c_u8p = ctypes.POINTER(ctypes.c_uint8)
i32 = ctypes.c_int32
vp = ctypes.c_void_p
sz = ctypes.c_size_t

_SIGS = {
    "ozk_last_error": (ctypes.c_char_p, []),
    "ozk_version": (ctypes.c_int, []),
    "ozk_device_count": (ctypes.c_int, []),
    "ozk_var_msm_host": (ctypes.c_int, [vp, vp, i32, i32, i32, vp]),
    "ozk_var_double_msm_host": (ctypes.c_int, [vp, vp, vp, i32, i32, vp]),
    "ozk_var_msm_sharded_host": (ctypes.c_int, [vp, vp, i32, i32, i32, vp]),
    "ozk_var_msm_auto_host": (ctypes.c_int, [vp, vp, i32, i32, i32, vp]),
    "ozk_var_double_msm_sharded_host": (ctypes.c_int, [vp, vp, vp, i32, i32, vp]),
    "ozk_var_double_msm_auto_host": (ctypes.c_int, [vp, vp, vp, i32, i32, vp]),
    "ozk_shard_last_exchange": (ctypes.c_int, []),
    "ozk_shard_comms_release": (None, []),
    "ozk_var_msm_workspace_bytes": (sz, [i32, i32]),
    "ozk_var_msm_dev": (ctypes.c_int, [vp, vp, i32, i32, vp, vp, sz, vp]),
    "ozk_prof_enable": (ctypes.c_int, [ctypes.c_int]),
    "ozk_prof_dominant_kernel_ms": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]),
    "ozk_prof_dominant_kernel_stats": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]),
    "ozk_prof_clock_khz": (ctypes.c_double, []),
    "ozk_prof_dominant_kernel_clock_mhz": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]),
    "ozk_var_msm_plan": (ctypes.c_int, [i32, ctypes.POINTER(i32), ctypes.POINTER(i32)]),
    "ozk_var_msm_glv": (ctypes.c_int, [i32]),
    "ozk_gen_bases_dev": (ctypes.c_int, [ctypes.c_uint64, i32, i32, vp, vp]),
    "ozk_var_msm_stage_bytes": (ctypes.c_int, [i32, i32, ctypes.POINTER(sz), ctypes.POINTER(sz), ctypes.POINTER(sz)]),
    "ozk_var_msm_sort_dev": (ctypes.c_int, [vp, vp, i32, i32, vp, sz, vp, sz, vp]),
    "ozk_var_msm_accum_dev": (ctypes.c_int, [i32, i32, vp, sz, vp, sz, vp, sz, vp]),
    "ozk_var_msm_sort_prepared_dev": (ctypes.c_int, [vp, vp, i32, i32, vp, sz, vp, sz, vp]),
    "ozk_var_msm_accum_prepared_dev": (ctypes.c_int, [vp, i32, i32, vp, sz, vp, sz, vp, sz, vp]),
    "ozk_device_cu_count": (ctypes.c_int, []),
    "ozk_stream_create_cu_range": (ctypes.c_int, [i32, i32, ctypes.POINTER(ctypes.c_void_p)]),
    "ozk_stream_destroy": (ctypes.c_int, [vp]),
    "ozk_var_msm_accum_part_dev": (ctypes.c_int, [vp, i32, i32, vp, sz, vp, sz, vp, sz, vp, i32]),
    "ozk_var_msm_head_workspace_bytes": (sz, [i32, i32]),
    "ozk_var_msm_tail_bytes": (sz, [i32, i32]),
    "ozk_var_msm_head_dev": (ctypes.c_int, [vp, vp, i32, i32, vp, sz, vp, sz, vp]),
    "ozk_var_msm_tail_dev": (ctypes.c_int, [i32, i32, vp, sz, vp, vp]),
    "ozk_bases_create_host": (ctypes.c_int, [vp, i32, i32, i32, ctypes.POINTER(vp)]),
    "ozk_var_msm_bases_host": (ctypes.c_int, [vp, vp, i32, vp]),
    "ozk_bases_destroy": (ctypes.c_int, [vp]),
    "ozk_var_msm_prepared_bytes": (sz, [i32, i32]),
    "ozk_var_msm_prepare_dev": (ctypes.c_int, [vp, i32, i32, vp, sz, vp]),
    "ozk_var_msm_prepared_dev": (ctypes.c_int, [vp, vp, i32, i32, vp, vp, sz, vp]),
    "ozk_var_msm_head_prepared_dev": (ctypes.c_int, [vp, vp, i32, i32, vp, sz, vp, sz, vp, vp]),
    "ozk_order_event_create": (ctypes.c_int, [ctypes.POINTER(vp)]),
    "ozk_order_event_destroy": (ctypes.c_int, [vp]),
    "ozk_var_msm_head_ordered_dev": (ctypes.c_int, [vp, vp, i32, i32, vp, sz, vp, sz, vp, vp]),
    "ozk_var_msm_tail_ordered_dev": (ctypes.c_int, [i32, i32, vp, sz, vp, vp, vp]),
    "ozk_var_msm_tail_mode_dev": (ctypes.c_int, [i32, i32, vp, sz, vp, vp, vp, i32]),
    "ozk_points_sum_dev": (ctypes.c_int, [vp, i32, i32, vp, vp]),
    "ozk_fixed_batch_msm_host": (ctypes.c_int, [i32, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp]),
    "ozk_fixed_double_batch_msm_host": (ctypes.c_int, [i32] * 9 + [vp, vp, vp, i32, vp]),
    "ozk_field_batch_mul_host": (ctypes.c_int, [vp, i32, i32, vp]),
    "ozk_fixed_batch_msm_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "ozk_fixed_batch_msm_dev": (ctypes.c_int, [i32, i32, i32, vp, vp, i32, vp, vp, sz, vp]),
    "ozk_field_batch_mul_dev": (ctypes.c_int, [vp, i32, vp, vp]),
    "ozk_fixed_batch_msm_compact_dev": (ctypes.c_int, [i32, i32, i32, vp, vp, i32, vp, vp, sz, vp]),
    "ozk_fixed_batch_msm_compact_host": (ctypes.c_int, [i32, i32, i32, vp, vp, i32, i32, vp]),
    "ozk_fixed_batch_msm_base_dev": (ctypes.c_int, [i32, i32, i32, vp, vp, i32, vp, i32, vp, sz, vp]),
    "ozk_fft_compact_host": (ctypes.c_int, [vp, i32, vp, i32, vp]),
    "ozk_fft_compact_dev": (ctypes.c_int, [vp, i32, vp, vp, vp, sz, vp]),
    "ozk_tuning_reload": (ctypes.c_int, []),
    "ozk_host_cache_release": (ctypes.c_int, []),
    "ozk_host_call_stats": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double)]),
    "ozk_bases_type": (ctypes.c_int, [vp]),
    "ozk_fft_host": (ctypes.c_int, [vp, i32, vp, i32, vp]),
    "ozk_fft_workspace_bytes": (sz, [i32]),
    "ozk_fft_dev": (ctypes.c_int, [vp, i32, vp, vp, vp, sz, vp]),
    "ozk_r1cs_evaluate_workspace_bytes": (sz, [i32]),
    "ozk_r1cs_evaluate_dev": (ctypes.c_int, [vp, vp, vp, vp, i32, vp, i32, vp, vp, sz, vp]),
    "ozk_qap_lagrange_workspace_bytes": (sz, [i32]),
    "ozk_qap_lagrange_dev": (ctypes.c_int, [vp, vp, i32, vp, vp, vp, sz, vp]),
    "ozk_sparse_mat_vec_dev": (ctypes.c_int, [vp, vp, vp, vp, i32, vp, i32, vp, vp, sz, vp]),
    "ozk_fr_powers_workspace_bytes": (sz, [i32]),
    "ozk_fr_powers_dev": (ctypes.c_int, [vp, vp, i32, vp, vp, sz, vp]),
    "ozk_fr_lincomb3_dev": (ctypes.c_int, [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp]),
    "ozk_qap_witness_host": (ctypes.c_int, [vp, vp, vp, i32, vp, vp, i32, vp]),
    "ozk_qap_witness_workspace_bytes": (sz, [i32]),
    "ozk_qap_witness_dev": (ctypes.c_int, [vp, vp, vp, i32, vp, vp, vp, vp, sz, vp]),
}


class OzkError(RuntimeError):
    pass


def exported_symbols():
    return sorted(_SIGS)


def load():
    """Load libozk_hip.so; raise if absent (the product has no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OzkError(
            "libozk_hip.so is not built (%s). Run `python -m octopuszk_amd.build`; "
            "there is no CPU fallback." % LIB_PATH)
    # Device pointers handed over by torch (bench.py, torch.distributed ranks) are only valid
    # in the HIP runtime instance torch itself loaded, so when torch is installed it must be
    # imported BEFORE libozk_hip.so pulls in libamdhip64 (else two runtimes coexist and the
    # second one reports "no ROCm-capable device").  A JVM loads the library without torch.
    try:
        import torch  # noqa: F401  (plumbing only)
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    missing = [name for name in _SIGS if not hasattr(lib, name)]
    if missing:
        raise OzkError("libozk_hip.so lacks symbols declared in include/ozk.h: %s" % ", ".join(missing))
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().ozk_last_error()
        raise OzkError("ozk call failed (%d): %s" % (rc, msg.decode() if msg else "?"))
