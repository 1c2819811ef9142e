"""ctypes binding of libsumma_gpu.so (C ABI: include/summa_gpu.h).

This is the binding a maintainer of the reference would write in Rust; the
Python flavour exists because this pipeline has no Rust toolchain.  Loading fails loudly if
the library is missing -- there is no fallback implementation.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SG_OK = 0
EXPORTS = [
    "sg_init", "sg_shutdown", "sg_collect_retired", "sg_host_register", "sg_host_unregister", "sg_stream_wait", "sg_last_error", "sg_device_count", "sg_device", "sg_bind_thread", "sg_version",
    "sg_msm_g1", "sg_msm_g1_dev", "sg_msm_g1_batch", "sg_msm_g1_batch_dev", "sg_g1_sum_affine", "sg_srs_upload", "sg_srs_upload_dev", "sg_srs_copy_dev", "sg_srs_free", "sg_srs_check", "sg_commit", "sg_commit_dev",
    "sg_srs_device_ptrs", "sg_srs_precompute", "sg_commit_batch_dev", "sg_commit_batch_mixed_dev", "sg_commit_combine_begin", "sg_commit_combine_end", "sg_commit_combining", "sg_commit_combine_stats", "sg_ntt_fr", "sg_ntt_fr_dev", "sg_ntt_fr_batch_dev", "sg_ntt_fr_batch_oop_dev", "sg_intt_fr", "sg_intt_fr_dev",
    "sg_lagrange_to_coeff", "sg_lagrange_to_coeff_dev", "sg_coeff_to_extended", "sg_coeff_to_extended_dev", "sg_coeff_to_extended_batch_dev",
    "sg_extended_to_coeff", "sg_extended_to_coeff_dev", "sg_coeff_to_cosets_batch_dev", "sg_cosets_to_pieces_dev", "sg_quotient_permutation_cosets_dev", "sg_quotient_lookup_cosets_dev", "sg_quotient_gates_cosets_dev", "sg_gates_program_info", "sg_gates_program_words", "sg_mst_inclusion_keygen_columns", "sg_divide_by_vanishing_poly",
    "sg_divide_by_vanishing_poly_dev", "sg_domain_constant", "sg_g1_fixed_base_mul", "sg_g1_fixed_base_mul_dev", "sg_g2_generator_mul", "sg_pairing_check", "sg_pairing_check_slow", "sg_keccak256", "sg_kzg_setup", "sg_kzg_setup_dev", "sg_g1_fft_dev", "sg_g1_to_lagrange",
    "sg_fr_to_montgomery_dev", "sg_fr_from_montgomery_dev", "sg_fr_random_dev", "sg_fr_random_batch_dev", "sg_lookup_permute_small_dev", "sg_fr_eval_poly", "sg_fr_eval_poly_dev", "sg_fr_eval_poly_batch_dev",
    "sg_fr_batch_invert_dev", "sg_fr_prefix_product_dev", "sg_fr_mul_dev", "sg_fr_kate_division_dev", "sg_fr_kate_division_batch_dev", "sg_fr_count_noncanonical_dev", "sg_fr_lincomb_dev", "sg_fr_lincomb_low_dev", "sg_permutation_product_dev",
    "sg_lookup_product_dev", "sg_grand_products_dev", "sg_quotient_permutation_dev", "sg_quotient_lookup_dev", "sg_quotient_gates_dev", "sg_mst_leaves_dev", "sg_mst_level_dev", "sg_mst_build_dev", "sg_mst_inclusion_witness_dev", "sg_msm_g1_dev_timed", "sg_commit_dev_timed", "sg_fr_lincomb_sets_dev", "sg_quotient_numerator_cosets_dev", "sg_fr_flag_noncanonical_dev", "sg_lookup_permute_small_async_dev", "sg_grand_products_closing_dev", "sg_fr_kate_division_rem_dev", "sg_set_param", "sg_get_param", "sg_msm_launch_log", "sg_abi_version", "sg_time_ntt_dev",
]


class SummaGpuError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"summa_gpu error {code}: {msg}")
        self.code = code


class MsmTimings(C.Structure):
    _fields_ = [("digits_ms", C.c_float), ("sort_ms", C.c_float), ("accumulate_ms", C.c_float),
                ("reduce_ms", C.c_float), ("total_ms", C.c_float), ("window_bits", C.c_uint32),
                ("windows", C.c_uint32), ("tasks", C.c_uint32), ("max_bucket", C.c_uint32),
                ("accumulate_threads", C.c_uint32), ("order_ms", C.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def library_path() -> str:
    return os.path.join(_HERE, "libsumma_gpu.so")


def lib():
    """Load the HIP library (built by __graft_entry__.build() / csrc/Makefile)."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise SummaGpuError(-2, f"{path} not built: run `make -C circuits_halo2_amd/csrc` "
                                    "(there is no CPU fallback for the MSM/NTT path)")
        L = C.CDLL(path)
        L.sg_last_error.restype = C.c_char_p
        L.sg_version.restype = C.c_char_p
        for name in EXPORTS:
            getattr(L, name)  # fail now if a declared symbol is missing
        _LIB = L
    return _LIB


def check(rc: int):
    if rc != SG_OK:
        raise SummaGpuError(rc, lib().sg_last_error().decode())


def set_param(name: str, value: int):
    check(lib().sg_set_param(name.encode(), int(value)))


def get_param(name: str) -> int:
    out = C.c_int(0)
    check(lib().sg_get_param(name.encode(), C.byref(out)))
    return out.value


def msm_launch_log():
    """records of the msm_accumulate launches since parameter "msm.acc_log" was set to 1 (include/summa_gpu.h: sg_msm_launch_log)"""
    n = C.c_size_t(0)
    check(lib().sg_msm_launch_log(None, 0, C.byref(n)))
    buf = np.zeros((max(1, n.value), 8), dtype=np.uint32)
    check(lib().sg_msm_launch_log(buf.ctypes.data_as(C.POINTER(C.c_uint32)), buf.shape[0], C.byref(n)))
    keys = ("n", "M", "threads", "fixed", "jobs_in_flight", "task_len")
    return [dict(entries=int(r[0]) | (int(r[1]) << 32), **{k: int(v) for k, v in zip(keys, r[2:])}) for r in buf[:min(n.value, buf.shape[0])]]


def bind_thread():
    """Make the library's device the calling thread's current device, for HIP and for torch.  The current device is a
    property of the host thread and starts at 0 in every new thread: on a multi-GPU host (one process per GPU, rank r on
    device r) a worker thread's `device="cuda"` allocations and `torch.cuda.Stream()`s would land on device 0.  Every
    thread pool of this package starts its threads with this; callers with threads of their own do the same.  No-op
    before the library is bound to a device (sg_init or its first call)."""
    dev = lib().sg_device()
    if dev >= 0:
        import torch
        torch.cuda.set_device(dev)          # hipSetDevice on this thread + torch's notion of the current device
    return dev


def cuda_device():
    """torch.device of the GPU this process is bound to (what a bare "cuda" means on the thread that called sg_init)"""
    import torch
    dev = lib().sg_device()
    return torch.device("cuda", dev if dev >= 0 else torch.cuda.current_device())


def u8(a) -> np.ndarray:
    a = np.ascontiguousarray(a)
    if a.dtype != np.uint8:
        a = a.view(np.uint8)
    return a.reshape(-1)


def ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def dev_ptr(t):
    """torch CUDA tensor -> raw device pointer.  A tensor on another GPU than the library's is refused here: handed to a
    kernel it would be a fault on the device, not an error code (see bind_thread)."""
    assert t.is_cuda and t.is_contiguous()
    bound = lib().sg_device()
    if bound >= 0 and t.device.index != bound:
        raise SummaGpuError(-2, f"tensor on cuda:{t.device.index}, library bound to device {bound}: "
                                "call ffi.bind_thread() in the thread that allocates")
    return C.c_void_p(t.data_ptr())


def current_stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)

# C ABI of the compiled-host prover (include/summa_prover.h), exported by the same library
PROVER_EXPORTS = ["sp_key_create", "sp_key_destroy", "sp_create_proof", "sp_last_error", "sp_verify_proof", "sp_verify_last_error"]


def prover_lib():
    L = lib()
    if not getattr(L, "_sp_ready", False):
        for name in PROVER_EXPORTS:
            getattr(L, name)
        L.sp_last_error.restype = C.c_char_p
        L.sp_verify_last_error.restype = C.c_char_p
        L._sp_ready = True
    return L


def check_prover(rc: int):
    if rc != SG_OK:
        raise SummaGpuError(rc, prover_lib().sp_last_error().decode())
