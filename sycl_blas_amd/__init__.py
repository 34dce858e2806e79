"""MI355X-native GEMM engine behind the portBLAS `_gemm` interface.

`from sycl_blas_amd import SB_Handle, _gemm, _gemm_batched, _gemm_strided_batched`
gives the reference's call signatures over hand-written gfx950 HIP kernels
(loaded from sycl_blas_amd/lib/libsbamd.so through the C ABI of
include/sbamd_gemm.h).
"""
from . import _lib
from .interface import (DeviceBuffer, Event, SB_Handle, SbamdError, _gemm, _gemm_batched,
                        _gemm_strided_batched, _symm, _trsm, gemm_batch_type_t, helper, uniform_reference)

__all__ = [
    "DeviceBuffer", "Event", "SB_Handle", "SbamdError", "_gemm", "_gemm_batched",
    "_gemm_strided_batched", "_symm", "_trsm", "gemm_batch_type_t", "helper", "uniform_reference", "_lib",
]
