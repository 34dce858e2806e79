"""Python face of the CPU oracle (oracle/gemm_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by the smoke check and by the
CPU-baseline leg of bench.py.  The product package (sycl_blas_amd) never imports
it.  Parity is "unpinned" in the strict sense -- the reference ships no golden
vectors for GEMM and cannot be built here -- so the oracle is pinned the way the
reference pins itself: against OpenBLAS within the reference's own tolerance
predicate (see gemm_oracle.c header and tests/test_oracle.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

FORM_REF, FORM_LOCAL, FORM_SPLITK, FORM_SPLITK_TREE = 0, 1, 2, 3

# Margins of common/include/common/float_comparison.hpp:100-158 (abs, rel).
MARGINS = {"float32": (1e-3, 5e-3), "float64": (1e-10, 1e-10), "float16": (1.0, 0.05)}

_lib = None


def build():
    """Compile liboracle.so with the Makefile next to this file."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    i64, vp = C.c_int64, C.c_void_p
    for suf, sc in (("f32", C.c_float), ("f64", C.c_double)):
        f = getattr(lib, "oracle_gemm_" + suf)
        f.restype = C.c_int
        f.argtypes = [C.c_int, C.c_int, C.c_int, i64, i64, i64, sc, vp, i64, vp, i64, sc, vp, i64, i64, i64]
        f = getattr(lib, "oracle_gemm_backend_" + suf)
        f.restype = C.c_int
        f.argtypes = [C.c_int, C.c_char, C.c_char, i64, i64, i64, sc, vp, i64, i64, vp, i64, i64, sc, vp,
                      i64, i64, i64, C.c_int, i64, i64]
        f = getattr(lib, "oracle_gemm_default_tile_" + suf)
        f.restype = C.c_int
        f.argtypes = [C.c_int, C.c_int, i64, i64, i64, sc, vp, i64, vp, i64, sc, vp, i64]
    lib.oracle_gemm_truth.restype = C.c_int
    lib.oracle_gemm_truth.argtypes = [C.c_int, C.c_int, i64, i64, i64, C.c_double, vp, i64, vp, i64,
                                      C.c_double, vp, i64, vp, vp, i64]
    lib.oracle_compare_f64.restype = i64
    lib.oracle_compare_f64.argtypes = [vp, vp, i64, C.c_double, C.c_double]
    lib.oracle_gemm_half_chain.restype = C.c_int
    lib.oracle_gemm_half_chain.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, i64, i64, i64, C.c_uint16, vp, i64, vp,
                                           i64, C.c_uint16, vp, i64]
    lib.oracle_half_from_double.restype = None
    lib.oracle_half_from_double.argtypes = [vp, vp, i64]
    lib.oracle_half_to_double.restype = None
    lib.oracle_half_to_double.argtypes = [vp, vp, i64]
    lib.oracle_default_strides.restype = None
    lib.oracle_default_strides.argtypes = [C.c_char, C.c_char, i64, i64, i64, i64, i64, i64,
                                           C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
    _lib = lib
    return lib


def _suffix(arr):
    if arr.dtype == np.float32:
        return "f32", C.c_float
    if arr.dtype == np.float64:
        return "f64", C.c_double
    raise TypeError("oracle works on float32 / float64 arrays")


def _ch(x):
    return x.encode()[:1] if isinstance(x, str) else bytes([x])[:1]


def gemm_backend(transa, transb, m, n, k, alpha, a, lda, stride_a, b, ldb, stride_b, beta, c, ldc,
                 stride_c, batch=1, batch_type=0, form=FORM_REF, depth=1, k_chunk=0, a_off=0,
                 b_off=0, c_off=0):
    """Front end + arithmetic of the reference on flat 1-D arrays (c is updated in place).

    a_off / b_off / c_off are element offsets into the flat buffers (the
    reference's `buffer + offset`).  Returns the status code (0 = ok, 1..5 = the
    reference's invalid_argument cases).
    """
    lib = load()
    suf, sc = _suffix(c)
    assert a.dtype == c.dtype and b.dtype == c.dtype
    assert a.flags["C_CONTIGUOUS"] and b.flags["C_CONTIGUOUS"] and c.flags["C_CONTIGUOUS"]
    fn = getattr(lib, "oracle_gemm_backend_" + suf)
    isz = c.itemsize
    return fn(form, _ch(transa), _ch(transb), m, n, k, sc(alpha), a.ctypes.data + a_off * isz, lda,
              stride_a, b.ctypes.data + b_off * isz, ldb, stride_b, sc(beta), c.ctypes.data + c_off * isz,
              ldc, stride_c, batch, int(batch_type), depth, k_chunk if k_chunk > 0 else max(k, 1))


def gemm_default_tile(transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc):
    """One GEMM through the restatement of the kernel the reference's CPU back end runs for M*N >= 524288
    (no_local, partial, Tile<4,4,4,4>: src/interface/blas3/backend/default.hpp:98-112); c is updated in place."""
    lib = load()
    suf, sc = _suffix(c)
    assert a.dtype == c.dtype and b.dtype == c.dtype
    return getattr(lib, "oracle_gemm_default_tile_" + suf)(int(transa.lower() != "n"), int(transb.lower() != "n"), m, n, k,
                                                           sc(alpha), a.ctypes.data, lda, b.ctypes.data, ldb, sc(beta),
                                                           c.ctypes.data, ldc)


def default_strides(transa, transb, m, n, k, lda, ldb, ldc):
    sa, sb, sc = C.c_int64(), C.c_int64(), C.c_int64()
    load().oracle_default_strides(_ch(transa), _ch(transb), m, n, k, lda, ldb, ldc, C.byref(sa),
                                  C.byref(sb), C.byref(sc))
    return sa.value, sb.value, sc.value


def half_from_double(x):
    """The oracle's own round-to-nearest-even double -> binary16 conversion (bit patterns)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty(x.shape, dtype=np.uint16)
    load().oracle_half_from_double(x.ctypes.data, out.ctypes.data, x.size)
    return out


def half_to_double(bits):
    bits = np.ascontiguousarray(bits, dtype=np.uint16)
    out = np.empty(bits.shape, dtype=np.float64)
    load().oracle_half_to_double(bits.ctypes.data, out.ctypes.data, bits.size)
    return out


def gemm_half_chain(transa, transb, m, n, k, alpha, a_bits, lda, b_bits, ldb, beta, c_bits, ldc, form=FORM_LOCAL,
                    fused=True, a_off=0, b_off=0, c_off=0):
    """One half -> half GEMM in the reference's own arithmetic for that pair: accumulator, alpha, beta and every
    operation in binary16 (src/operations/blas3/gemm_common.hpp:45-51).  Arrays are uint16 bit patterns; c_bits is
    updated in place.  form: FORM_LOCAL (the tiled kernels, the default) or FORM_REF (the naive kernel)."""
    for x in (a_bits, b_bits, c_bits):
        assert x.dtype == np.uint16 and x.flags["C_CONTIGUOUS"]
    al = int(np.float16(alpha).view(np.uint16))
    be = int(np.float16(beta).view(np.uint16))
    return load().oracle_gemm_half_chain(int(form), int(bool(fused)), int(transa.lower() != "n"),
                                         int(transb.lower() != "n"), m, n, k, al, a_bits.ctypes.data + 2 * a_off, lda,
                                         b_bits.ctypes.data + 2 * b_off, ldb, be, c_bits.ctypes.data + 2 * c_off, ldc)


def gemm_truth(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc):
    """Long-double result and magnitude sum for an m x n problem (float64 arrays in, packed out)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    c = np.ascontiguousarray(c, dtype=np.float64)
    out = np.empty(m * n, dtype=np.float64)
    mag = np.empty(m * n, dtype=np.float64)
    load().oracle_gemm_truth(int(ta), int(tb), m, n, k, float(alpha), a.ctypes.data, lda, b.ctypes.data,
                             ldb, float(beta), c.ctypes.data, ldc, out.ctypes.data, mag.ctypes.data, m)
    return out, mag


def compare_vectors(vec, ref, dtype_name=None):
    """utils::compare_vectors (common/include/common/float_comparison.hpp:197-217).

    Returns -1 when every element passes, else the index of the first mismatch.
    """
    name = dtype_name or np.asarray(ref).dtype.name
    abs_m, rel_m = MARGINS[name]
    v = np.ascontiguousarray(vec, dtype=np.float64).ravel()
    r = np.ascontiguousarray(ref, dtype=np.float64).ravel()
    assert v.size == r.size
    return load().oracle_compare_f64(v.ctypes.data, r.ctypes.data, v.size, abs_m, rel_m)


def _bind_callers(lib):
    if getattr(lib, "_callers_bound", False):
        return
    i64, vp, ch = C.c_int64, C.c_void_p, C.c_char
    for suf, sc in (("f32", C.c_float), ("f64", C.c_double)):
        f = getattr(lib, "oracle_symm_" + suf)
        f.restype = C.c_int
        f.argtypes = [C.c_int, ch, ch, i64, i64, sc, vp, i64, vp, i64, sc, vp, i64, i64, i64]
        f = getattr(lib, "oracle_trsm_" + suf)
        f.restype = C.c_int
        f.argtypes = [C.c_int, ch, ch, ch, ch, i64, i64, sc, vp, i64, vp, i64]
    lib._callers_bound = True


def symm(side, uplo, m, n, alpha, a, lda, b, ldb, beta, c, ldc, form=FORM_REF, depth=1, k_chunk=0):
    """_symm of the reference on flat column-major arrays (c updated in place); returns the status code
    (0 ok, 11 "invalid _uplo", 12 "invalid _side").  depth / k_chunk give the K cut of the split-K forms."""
    lib = load()
    _bind_callers(lib)
    suf, sc = _suffix(c)
    assert a.dtype == c.dtype and b.dtype == c.dtype
    return getattr(lib, "oracle_symm_" + suf)(form, _ch(side), _ch(uplo), m, n, sc(alpha), a.ctypes.data, lda,
                                              b.ctypes.data, ldb, sc(beta), c.ctypes.data, ldc, depth,
                                              k_chunk if k_chunk > 0 else max(m if _ch(side).lower() == b"l" else n, 1))


def trsm(side, uplo, trans, diag, m, n, alpha, a, lda, b, ldb, form=FORM_REF):
    """_trsm of the reference (16x16 inverted diagonal blocks + GEMM sweeps); b is overwritten with X.
    Returns the status code (0 ok, 13..17 the reference's invalid_argument cases)."""
    lib = load()
    _bind_callers(lib)
    suf, sc = _suffix(b)
    assert a.dtype == b.dtype
    return getattr(lib, "oracle_trsm_" + suf)(form, _ch(side), _ch(uplo), _ch(trans), _ch(diag), m, n, sc(alpha),
                                              a.ctypes.data, lda, b.ctypes.data, ldb)
