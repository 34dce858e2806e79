"""Shared machinery of the parity tests.

`verify_gemm` / `verify_gemm_strided` follow the reference's test drivers
(test/unittest/blas3/blas3_gemm_common.hpp:96-230 and :288-409): same buffer
sizing, offsets and leading-dimension multipliers, inputs uniform in [-2, 5]
(test/blas_test.hpp:137-140) -- but seeded, because the reference's
std::random_device inputs cannot be replayed.

Checks per case:
  1. the WHOLE C buffer (offset area, padding rows, gaps between batch items)
     equals the CPU oracle bit for bit -- the fp32/fp64 kernels run the same
     fused-multiply-add chain over k as the oracle's restatement of
     src/operations/blas3/gemm_ref.hpp:234-251;
  2. the reference's own pass/fail predicate against OpenBLAS
     (common/include/common/float_comparison.hpp:163-217, with
     scipy's sgemm/dgemm in the role of cblas_sgemm/cblas_dgemm of
     common/include/common/system_reference_blas.hpp:402-431).
Split-K results depend on how K is cut, so there check 1 uses the oracle's
split-K form with the same cut.
"""
import zlib

import numpy as np

import oracle
import sycl_blas_amd as sb
from sycl_blas_amd import gemm_batch_type_t, helper

strided = gemm_batch_type_t.strided
interleaved = gemm_batch_type_t.interleaved


def seeded(*key):
    return np.random.default_rng(zlib.crc32(repr(key).encode()))


def fill_random(rng, count, dtype):
    return rng.uniform(-2.0, 5.0, size=count).astype(dtype)


def openblas_gemm(transa, transb, m, n, k, alpha, a, a_off, lda, b, b_off, ldb, beta, c, c_off, ldc):
    """reference_blas::gemm on one item of flat column-major buffers; updates c in place."""
    from scipy.linalg import blas

    fn = blas.sgemm if c.dtype == np.float32 else blas.dgemm
    ta, tb = transa.lower() != "n", transb.lower() != "n"
    ar, ac = (k, m) if ta else (m, k)
    br, bc = (n, k) if tb else (k, n)
    if m == 0 or n == 0:
        return
    A = np.lib.stride_tricks.as_strided(a[a_off:], shape=(ar, ac), strides=(a.itemsize, lda * a.itemsize))
    B = np.lib.stride_tricks.as_strided(b[b_off:], shape=(br, bc), strides=(b.itemsize, ldb * b.itemsize))
    Cv = np.lib.stride_tricks.as_strided(c[c_off:], shape=(m, n), strides=(c.itemsize, ldc * c.itemsize))
    if k == 0:
        Cv[...] = beta * Cv if beta != 0 else 0
        return
    out = fn(alpha, A, B, beta=beta, c=np.asfortranarray(Cv), trans_a=ta, trans_b=tb)
    Cv[...] = out


def to_interleaved(x, offset, ld_rows, ld_cols, batch):
    """strided_to_interleaved (blas3_gemm_common.hpp:54-69)."""
    out = x.copy()
    body = x[offset:offset + batch * ld_cols * ld_rows].reshape(batch, ld_cols, ld_rows)
    out[offset:offset + batch * ld_cols * ld_rows] = body.transpose(1, 2, 0).ravel()
    return out


def from_interleaved(x, offset, ld_rows, ld_cols, batch):
    """interleaved_to_strided (blas3_gemm_common.hpp:71-87)."""
    out = x.copy()
    body = x[offset:offset + batch * ld_cols * ld_rows].reshape(ld_cols, ld_rows, batch)
    out[offset:offset + batch * ld_cols * ld_rows] = body.transpose(2, 0, 1).ravel()
    return out


def run_device(h, call, a_m, b_m, c_m):
    """Upload, run `call(a_gpu, b_gpu, c_gpu, deps)`, download C."""
    a_gpu = helper.allocate(h, a_m.dtype, a_m.size)
    b_gpu = helper.allocate(h, b_m.dtype, b_m.size)
    c_gpu = helper.allocate(h, c_m.dtype, c_m.size)
    deps = [helper.copy_to_device(h, a_m, a_gpu, a_m.size), helper.copy_to_device(h, b_m, b_gpu, b_m.size),
            helper.copy_to_device(h, c_m, c_gpu, c_m.size)]
    try:
        ev = call(a_gpu, b_gpu, c_gpu, deps)
        h.wait(ev)
        out = np.empty_like(c_m)
        h.wait(helper.copy_to_host(h, c_gpu, out, out.size))
        h.wait()
    finally:
        for buf in (a_gpu, b_gpu, c_gpu):
            helper.deallocate(buf)
    return out


def assert_same_bits(got, want, what):
    gv = got.view(np.uint32 if got.dtype == np.float32 else np.uint64)
    wv = want.view(gv.dtype)
    if not np.array_equal(gv, wv):
        bad = np.flatnonzero(gv != wv)
        i = int(bad[0])
        raise AssertionError(f"{what}: {bad.size} of {got.size} elements differ from the oracle; "
                             f"first at {i}: got {got[i]!r}, oracle {want[i]!r}")


def assert_reference_predicate(got, ref, what):
    idx = oracle.compare_vectors(got, ref, got.dtype.name)
    assert idx == -1, f"{what}: reference predicate fails at {idx}: {got.ravel()[idx]} vs {ref.ravel()[idx]}"


def verify_gemm(h, dtype, offset, batch, m, n, k, transa, transb, alpha, beta, lda_mul, ldb_mul,
                ldc_mul, batch_type=strided, exact=True):
    """blas3_gemm_common.hpp:96-230 with the GPU in place of portBLAS."""
    rng = seeded(np.dtype(dtype).name, offset, batch, m, n, k, transa, transb, alpha, beta, lda_mul,
                 ldb_mul, ldc_mul, int(batch_type))
    lda = (k if transa != "n" else m) * lda_mul
    ldb = (n if transb != "n" else k) * ldb_mul
    ldc = m * ldc_mul
    size_a, size_b, size_c = m * k * lda_mul, k * n * ldb_mul, m * n * ldc_mul
    a_m = fill_random(rng, batch * size_a + offset, dtype)
    b_m = fill_random(rng, batch * size_b + offset, dtype)
    c_m = fill_random(rng, batch * size_c + offset, dtype)

    a_d, b_d, c_d = a_m, b_m, c_m
    inter = batch > 1 and batch_type == interleaved
    if inter:
        a_d = to_interleaved(a_m, offset, lda, m if transa == "t" else k, batch)
        b_d = to_interleaved(b_m, offset, ldb, k if transb == "t" else n, batch)
        c_d = to_interleaved(c_m, offset, ldc, n, batch)

    def call(a_gpu, b_gpu, c_gpu, deps):
        if batch == 1:
            return sb._gemm(h, transa, transb, m, n, k, alpha, a_gpu + offset, lda, b_gpu + offset, ldb, beta,
                            c_gpu + offset, ldc, deps)
        return sb._gemm_batched(h, transa, transb, m, n, k, alpha, a_gpu + offset, lda, b_gpu + offset, ldb,
                                beta, c_gpu + offset, ldc, batch, batch_type, deps)

    c_gpu = run_device(h, call, a_d, b_d, c_d)
    if inter:
        c_gpu = from_interleaved(c_gpu, offset, ldc, n, batch)

    # ground truth 1: the oracle (bitwise).  A split-K launch is reproduced with
    # the cut the library reports; everything else is the plain k-ordered chain.
    depth, k_chunk = h.last_splitk()
    form = oracle.FORM_SPLITK if depth > 1 else oracle.FORM_REF
    c_orc = c_m.copy()
    # blas::_gemm passes zero strides (gemm_interface.hpp:194-197); _gemm_batched whole matrices (:213-219)
    o_sa, o_sb, o_sc = (0, 0, 0) if batch == 1 else (size_a, size_b, size_c)
    st = oracle.gemm_backend(transa, transb, m, n, k, alpha, a_m, lda, o_sa, b_m, ldb, o_sb, beta, c_orc,
                             ldc, o_sc, batch=batch, batch_type=0, form=form, depth=depth, k_chunk=k_chunk,
                             a_off=offset, b_off=offset, c_off=offset)
    assert st == 0
    # ground truth 2: OpenBLAS under the reference's predicate
    c_blas = c_m.copy()
    for i in range(batch):
        openblas_gemm(transa, transb, m, n, k, alpha, a_m, i * size_a + offset, lda, b_m, i * size_b + offset,
                      ldb, beta, c_blas, i * size_c + offset, ldc)
    what = f"gemm {np.dtype(dtype).name} off={offset} batch={batch} {m}x{n}x{k} {transa}{transb} a={alpha} b={beta}"
    assert_reference_predicate(c_gpu, c_blas, what)
    if exact:
        assert_same_bits(c_gpu, c_orc, what)
    else:
        assert_reference_predicate(c_gpu, c_orc, what + " (vs oracle)")
    return c_gpu


def verify_gemm_strided(h, dtype, offset, batch, m, n, k, transa, transb, alpha, beta, lda_mul, ldb_mul,
                        ldc_mul, sa_mul, sb_mul, sc_mul):
    """blas3_gemm_common.hpp:288-409."""
    rng = seeded("strided", np.dtype(dtype).name, offset, batch, m, n, k, transa, transb, alpha, beta, lda_mul,
                 ldb_mul, ldc_mul, sa_mul, sb_mul, sc_mul)
    lda = (k if transa != "n" else m) * lda_mul
    ldb = (n if transb != "n" else k) * ldb_mul
    ldc = m * ldc_mul
    size_a, size_b, size_c = m * k * lda_mul, k * n * ldb_mul, m * n * ldc_mul
    stride_a, stride_b, stride_c = sa_mul * size_a, sb_mul * size_b, sc_mul * size_c
    a_m = fill_random(rng, size_a + (batch - 1) * stride_a + offset, dtype)
    b_m = fill_random(rng, size_b + (batch - 1) * stride_b + offset, dtype)
    c_m = fill_random(rng, size_c + (batch - 1) * stride_c + offset, dtype)

    c_blas = c_m.copy()
    for i in range(batch):
        openblas_gemm(transa, transb, m, n, k, alpha, a_m, i * stride_a + offset, lda, b_m,
                      i * stride_b + offset, ldb, beta, c_blas, i * stride_c + offset, ldc)

    def call(a_gpu, b_gpu, c_gpu, deps):
        return sb._gemm_strided_batched(h, transa, transb, m, n, k, alpha, a_gpu + offset, lda, stride_a,
                                        b_gpu + offset, ldb, stride_b, beta, c_gpu + offset, ldc, stride_c,
                                        batch, deps)

    c_gpu = run_device(h, call, a_m, b_m, c_m)
    # the oracle replays the K cut the library reports (only a single item is ever cut)
    depth, k_chunk = h.last_splitk()
    form = oracle.FORM_SPLITK if depth > 1 else oracle.FORM_REF
    c_orc = c_m.copy()
    st = oracle.gemm_backend(transa, transb, m, n, k, alpha, a_m, lda, stride_a, b_m, ldb, stride_b, beta, c_orc,
                             ldc, stride_c, batch=batch, form=form, depth=depth, k_chunk=k_chunk,
                             a_off=offset, b_off=offset, c_off=offset)
    assert st == 0
    what = (f"strided {np.dtype(dtype).name} off={offset} batch={batch} {m}x{n}x{k} {transa}{transb} "
            f"a={alpha} b={beta} strides x{sa_mul},{sb_mul},{sc_mul}")
    assert_reference_predicate(c_gpu, c_blas, what)
    assert_same_bits(c_gpu, c_orc, what)
    return c_gpu
