"""ctypes loader for libezkl_hip.so.  Fails loudly: no CPU fallback exists (and none may be added)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# every symbol include/ezkl_hip.h declares (tests assert the .so exports all of them)
SYMBOLS = [
    "ezkl_hip_init", "ezkl_hip_contexts_configure", "ezkl_hip_context_count", "ezkl_hip_set_context", "ezkl_hip_context_device", "ezkl_hip_memcpy_peer", "ezkl_hip_warmup", "ezkl_hip_device_count", "ezkl_hip_mem_info", "ezkl_hip_pool_stats", "ezkl_hip_pool_trim", "ezkl_hip_synchronize", "ezkl_hip_stream_create", "ezkl_hip_stream_synchronize", "ezkl_hip_stream_destroy", "ezkl_hip_context_stream", "ezkl_hip_strerror",
    "ezkl_hip_last_hip_error", "ezkl_hip_version", "ezkl_hip_enabled", "ezkl_hip_malloc", "ezkl_hip_free", "ezkl_hip_memcpy_h2d",
    "ezkl_hip_memcpy_d2h", "ezkl_hip_bases_upload", "ezkl_hip_bases_prepare", "ezkl_hip_bases_free", "ezkl_hip_bases_len", "ezkl_hip_bases_generate",
    "ezkl_hip_bases_download", "ezkl_hip_bases_from_scalars", "ezkl_hip_bases_downsize", "ezkl_hip_msm_g2", "ezkl_hip_msm_g1",
    "ezkl_hip_msm_g1_dev", "ezkl_hip_msm_g1_start_dev", "ezkl_hip_msm_g1_finish", "ezkl_hip_msm_g1_batch", "ezkl_hip_msm_g1_batch_dev", "ezkl_hip_msm_g1_batch_small_dev", "ezkl_hip_g1_add_affine", "ezkl_hip_ntt", "ezkl_hip_ntt_dev",
    "ezkl_hip_coset_ntt_batch", "ezkl_hip_coset_ntt_dev", "ezkl_hip_coeff_to_cosets_dev", "ezkl_hip_coeff_to_cosets_range_dev", "ezkl_hip_cosets_transpose_dev", "ezkl_hip_vec_op_dev", "ezkl_hip_vec_scale_dev", "ezkl_hip_permutation_sigma_dev", "ezkl_hip_vec_fill_dev", "ezkl_hip_host_malloc", "ezkl_hip_host_free", "ezkl_hip_upload_commit_batch", "ezkl_hip_upload_begin", "ezkl_hip_upload_begin_fmt", "ezkl_hip_upload_wait", "ezkl_hip_upload_commit", "ezkl_hip_upload_end", "ezkl_hip_msm_batch_begin", "ezkl_hip_msm_batch_push_dev", "ezkl_hip_msm_batch_push_many_dev", "ezkl_hip_msm_batch_finish", "ezkl_hip_eval_poly_batch_dev", "ezkl_hip_lincomb_dev", "ezkl_hip_kate_division_dev", "ezkl_hip_chacha20_fr_dev",
    "ezkl_hip_divide_by_vanishing_dev", "ezkl_hip_prefix_scan_dev", "ezkl_hip_eval_poly_dev", "ezkl_hip_lookup_multiplicity_dev", "ezkl_hip_lookup_multiplicity_acc_dev", "ezkl_hip_lookup_multiplicity_batch_dev", "ezkl_hip_batch_invert_dev", "ezkl_hip_eval_h_dev", "ezkl_hip_eval_h_check", "ezkl_hip_eval_h_schedule", "ezkl_hip_eval_h_prepare", "ezkl_hip_eval_h_jit_stats", "ezkl_hip_set_async", "ezkl_hip_stream_wait_library",
    "ezkl_hip_last_kernel_ms", "ezkl_hip_kernel_ms_stats", "ezkl_hip_ubench",
    "ezkl_hip_comm_available", "ezkl_hip_comm_unique_id", "ezkl_hip_comm_init", "ezkl_hip_comm_info", "ezkl_hip_comm_destroy", "ezkl_hip_comm_allgather_dev",
    "ezkl_hip_comm_fold_points", "ezkl_hip_comm_broadcast_host", "ezkl_hip_comm_alltoall_dev", "ezkl_hip_comm_allgather_host", "ezkl_hip_comm_alltoallv_dev", "ezkl_hip_comm_stats", "ezkl_hip_comm_selftest",
]


class EzklHipError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        msg = load().ezkl_hip_strerror(code).decode() if _LIB is not None else "library not loaded"
        super().__init__("%s failed: %d (%s)" % (what, code, msg))


def lib_path():
    return os.environ.get("EZKL_HIP_LIB") or os.path.join(_HERE, "libezkl_hip.so")      # override: A/B builds of the kernels


def load():
    """Load the in-tree HIP library.  Raises if it has not been built (python __graft_entry__.py)."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError("libezkl_hip.so missing at %s -- build it with `make -C ezkl_amd/csrc` "
                               "(there is no CPU fallback)" % path)
        lib = C.CDLL(path)
        lib.ezkl_hip_strerror.restype = C.c_char_p
        lib.ezkl_hip_version.restype = C.c_char_p
        lib.ezkl_hip_bases_len.restype = C.c_size_t
        _LIB = lib
    return _LIB


def check(rc, what):
    if rc != 0:
        raise EzklHipError(rc, what)
