"""ctypes binding of libsbamd.so (the C ABI declared in include/sbamd_gemm.h and include/sbamd_dist.h).

The library is loaded from the source tree (sycl_blas_amd/lib/libsbamd.so, built
by sycl_blas_amd/csrc/Makefile).  There is no fallback: if the shared object is
missing the import fails loudly, because every GEMM in this package runs in the
HIP kernels and nowhere else.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsbamd.so")

# status codes / enums (include/sbamd_gemm.h)
OK = 0
INVALID_TRANSA, INVALID_TRANSB, INVALID_STRIDEC, INVALID_STRIDEA, INVALID_STRIDEB = 1, 2, 3, 4, 5
INVALID_HANDLE, INVALID_VALUE, HIP_ERROR, ALLOC_FAILED, NOT_SUPPORTED = 6, 7, 8, 9, 10
INVALID_UPLO, INVALID_SIDE = 11, 12
TRSM_INVALID_SIZE, TRSM_INVALID_SIDE, TRSM_INVALID_UPLO, TRSM_INVALID_TRANS, TRSM_INVALID_DIAG = 13, 14, 15, 16, 17
SCRATCH_TOO_SMALL = 18
RCCL_ERROR, DIST_TIMEOUT, DIST_BOOTSTRAP, DIST_NO_TRANSPORT = 19, 20, 21, 22
DIST_MESH, DIST_RCCL = 1, 2
DIST_MAX_RANKS, DIST_MAX_BANDS = 16, 16
NOALLOC_SCRATCH_BYTES = 64 << 20
BATCH_STRIDED, BATCH_INTERLEAVED = 0, 1
KERNEL_AUTO, KERNEL_LARGE, KERNEL_SMALL, KERNEL_SPLITK, KERNEL_NAIVE, KERNEL_HUGE = 0, 1, 2, 3, 4, 5
KERNEL_INTERLEAVED = 6
KERNEL_TINY = 7
KERNEL_STREAM = 8
KERNEL_MEDIUM = 9

_i64, _int, _f32, _f64, _vp, _ch = C.c_int64, C.c_int, C.c_float, C.c_double, C.c_void_p, C.c_char


def _gemm_sig(scalar):
    return [_vp, _ch, _ch, _i64, _i64, _i64, scalar, _vp, _i64, _vp, _i64, scalar, _vp, _i64]


def _gemm_batched_sig(scalar):
    return _gemm_sig(scalar) + [_i64, _int]


def _gemm_strided_sig(scalar):
    return [_vp, _ch, _ch, _i64, _i64, _i64, scalar, _vp, _i64, _i64, _vp, _i64, _i64, scalar, _vp,
            _i64, _i64, _i64]


# name -> (restype, argtypes); every symbol include/sbamd_gemm.h declares
SIGNATURES = {
    "sbamd_status_string": (C.c_char_p, [_int]),
    "sbamd_create": (_int, [C.POINTER(_vp), _int, _vp]),
    "sbamd_destroy": (_int, [_vp]),
    "sbamd_wait": (_int, [_vp]),
    "sbamd_get_num_compute_units": (_int, [_vp, C.POINTER(_int)]),
    "sbamd_get_stream": (_int, [_vp, C.POINTER(_vp)]),
    "sbamd_last_error": (C.c_char_p, [_vp]),
    "sbamd_set_kernel": (_int, [_vp, _int, _int]),
    "sbamd_last_kernel": (_int, [_vp]),
    "sbamd_last_splitk": (_int, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "sbamd_last_tile": (_int, [_vp, C.POINTER(_int), C.POINTER(_int), C.POINTER(_int)]),
    "sbamd_last_trsm_launches": (_int, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "sbamd_event_record": (_int, [_vp, C.POINTER(_vp)]),
    "sbamd_stream_wait_event": (_int, [_vp, _vp]),
    "sbamd_event_synchronize": (_int, [_vp]),
    "sbamd_event_elapsed_ms": (_int, [_vp, _vp, C.POINTER(_f32)]),
    "sbamd_event_destroy": (_int, [_vp]),
    "sbamd_malloc": (_int, [_vp, C.POINTER(_vp), C.c_size_t]),
    "sbamd_free": (_int, [_vp, _vp]),
    "sbamd_device_free": (_int, [_int, _vp]),
    "sbamd_copy_to_device": (_int, [_vp, _vp, _vp, C.c_size_t]),
    "sbamd_copy_to_host": (_int, [_vp, _vp, _vp, C.c_size_t]),
    "sbamd_memset": (_int, [_vp, _vp, _int, C.c_size_t]),
    "sbamd_fill_uniform": (_int, [_vp, _vp, _i64, _int, C.c_uint64, _f64, _f64]),
    "sbamd_sgemm": (_int, _gemm_sig(_f32)),
    "sbamd_sgemm_noalloc": (_int, _gemm_sig(_f32)),
    "sbamd_dgemm": (_int, _gemm_sig(_f64)),
    "sbamd_hgemm": (_int, _gemm_sig(_f32)),
    "sbamd_hsgemm": (_int, _gemm_sig(_f32)),
    "sbamd_bsgemm": (_int, _gemm_sig(_f32)),
    "sbamd_sgemm_batched": (_int, _gemm_batched_sig(_f32)),
    "sbamd_dgemm_batched": (_int, _gemm_batched_sig(_f64)),
    "sbamd_hgemm_batched": (_int, _gemm_batched_sig(_f32)),
    "sbamd_hsgemm_batched": (_int, _gemm_batched_sig(_f32)),
    "sbamd_bsgemm_batched": (_int, _gemm_batched_sig(_f32)),
    "sbamd_sgemm_strided_batched": (_int, _gemm_strided_sig(_f32)),
    "sbamd_dgemm_strided_batched": (_int, _gemm_strided_sig(_f64)),
    "sbamd_hgemm_strided_batched": (_int, _gemm_strided_sig(_f32)),
    "sbamd_hsgemm_strided_batched": (_int, _gemm_strided_sig(_f32)),
    "sbamd_bsgemm_strided_batched": (_int, _gemm_strided_sig(_f32)),
}

for _p, _sc in (("s", _f32), ("d", _f64)):
    SIGNATURES[f"sbamd_{_p}symm"] = (_int, [_vp, _ch, _ch, _i64, _i64, _sc, _vp, _i64, _vp, _i64, _sc, _vp, _i64])
    SIGNATURES[f"sbamd_{_p}trsm"] = (_int, [_vp, _ch, _ch, _ch, _ch, _i64, _i64, _sc, _vp, _i64, _vp, _i64])

# ---- the multi-GPU layer (include/sbamd_dist.h)
ALLGATHER_FN = C.CFUNCTYPE(_int, _vp, _vp, _vp, C.c_size_t)
_pi64 = C.POINTER(_i64)


def _dist_gemm_sig(scalar):
    return [_vp, _ch, _ch, _i64, _i64, _i64, scalar, _vp, _i64, _vp, _i64, scalar, _vp, _i64]


DIST_SIGNATURES = {
    "sbamd_dist_create": (_int, [C.POINTER(_vp), _vp, _int, _int, ALLGATHER_FN, _vp, _int, C.c_size_t]),
    "sbamd_dist_destroy": (_int, [_vp]),
    "sbamd_dist_rank": (_int, [_vp, C.POINTER(_int), C.POINTER(_int), C.POINTER(_int)]),
    "sbamd_dist_register": (_int, [_vp, _vp, C.c_size_t, C.POINTER(_int)]),
    "sbamd_dist_unregister": (_int, [_vp, _int]),
    "sbamd_dist_barrier": (_int, [_vp]),
    "sbamd_dist_status": (_int, [_vp]),
    "sbamd_dist_grid": (None, [_int, C.POINTER(_int), C.POINTER(_int)]),
    "sbamd_dist_split_range": (None, [_i64, _int, _int, _pi64, _pi64]),
    "sbamd_dist_block_of": (_int, [_int, _int, _int, _i64, _i64, _pi64, _pi64, _pi64, _pi64]),
    "sbamd_dist_band_widths": (_int, [_i64, _int, _int, _pi64]),
    "sbamd_dist_all_gather_c": (_int, [_vp, _int, _int, _i64, _i64, _vp, _i64, _int, _int]),
    "sbamd_dist_last_bands": (_int, [_vp, _pi64, _pi64, _pi64, _pi64]),
}
for _p, _sc in (("s", _f32), ("d", _f64), ("bs", _f32)):
    SIGNATURES_BLOCK = [_vp, _int, _int, _ch, _ch, _i64, _i64, _i64, _sc, _vp, _i64, _vp, _i64, _sc, _vp, _i64]
    DIST_SIGNATURES[f"sbamd_dist_block_{_p}gemm"] = (_int, SIGNATURES_BLOCK)
for _p, _sc in (("s", _f32), ("d", _f64)):
    DIST_SIGNATURES[f"sbamd_dist_gathered_{_p}gemm"] = (_int, _dist_gemm_sig(_sc) + [_int, _int])
    DIST_SIGNATURES[f"sbamd_dist_splitk_{_p}gemm"] = (_int, _dist_gemm_sig(_sc) + [_int, _int])

_lib = None


def load():
    """Load libsbamd.so once and attach the prototypes.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C sycl_blas_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback for the GEMM path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in list(SIGNATURES.items()) + list(DIST_SIGNATURES.items()):
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def status_string(status):
    return load().sbamd_status_string(int(status)).decode()
