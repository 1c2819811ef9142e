"""MI355X-native back-end for the Halo2/KZG hot path of Summa's zk_prover.

Host-side mirror of the interface the reference reaches the path through
(halo2_proofs::arithmetic::{best_multiexp, best_fft}, poly::EvaluationDomain,
poly::kzg::commitment::ParamsKZG -- used by zk_prover/src/circuits/utils.rs:37-107), bound
over the C ABI in include/summa_gpu.h to hand-written HIP kernels (csrc/).  There is no CPU
implementation of the path in this package: without the HIP library or a GPU every call
raises.
"""
from .ffi import SummaGpuError, lib, library_path  # noqa: F401
from .arithmetic import best_fft, best_multiexp, best_multiexp_batch  # noqa: F401
from .domain import EvaluationDomain  # noqa: F401
from .params import ParamsKZG  # noqa: F401

__all__ = ["best_multiexp", "best_multiexp_batch", "best_fft", "EvaluationDomain", "ParamsKZG", "SummaGpuError", "lib", "library_path"]
