"""GPU parity tests for blas::_gemm: the reference's parameter grids
(test/unittest/blas3/blas3_gemm_test.cpp:29-141) for float and double."""
import itertools

import numpy as np
import pytest

import oracle  # noqa: F401
import sycl_blas_amd as sb
from sycl_blas_amd import _lib

from common import strided, verify_gemm

pytestmark = pytest.mark.gpu

DTYPES = [np.float32, np.float64]


def grid(*axes):
    return list(itertools.product(*axes))


# (offset, batch, m, n, k, transa, transb, alpha, beta, lda_mul, ldb_mul, ldc_mul)
SMALL_BETA_NONZERO_LD_MATCH = grid([0], [1], [11, 16, 32], [11, 16, 32], [16, 17], "nt", "nt", [1.5], [1.5], [1], [1], [1])
SMALL_BETA_ZERO_LD_MATCH = grid([0], [1], [11, 32], [11, 32], [17], "nt", "nt", [1.5], [0.0], [1], [1], [1])
SMALL_BETA_ZERO_LD_MULTIPLIED = grid([0], [1], [11, 32], [11, 32], [17], "nt", "nt", [1.5], [0.0], [2], [3], [4])
ALPHA_ZERO = grid([0, 10], [1], [16], [16], [17], "n", "n", [0.0], [0.0, 1.0], [1, 2], [1, 2], [1, 2])
OFFSET_NONZERO = grid([1, 10], [1], [16, 63], [16, 63], [17, 63], "n", "n", [1.0], [1.0], [1, 2], [1, 2], [1, 2])
LARGE_BETA_NONZERO_LD_MATCH = grid([0], [1], [253, 511, 1024, 2048, 2200], [257, 511], [253, 511], "nt", "nt", [1.0], [1.0], [1], [1], [1])

GRIDS = {
    "SmallBetaNonZeroLDMatch": SMALL_BETA_NONZERO_LD_MATCH,
    "SmallBetaZeroLDMatch": SMALL_BETA_ZERO_LD_MATCH,
    "SmallBetaZeroLDMultiplied": SMALL_BETA_ZERO_LD_MULTIPLIED,
    "AlphaZero": ALPHA_ZERO,
    "OffsetNonZero": OFFSET_NONZERO,
    "LargeBetaNonZeroLDMatch": LARGE_BETA_NONZERO_LD_MATCH,
}


@pytest.mark.parametrize("dtype", DTYPES, ids=lambda d: np.dtype(d).name)
@pytest.mark.parametrize("name", list(GRIDS))
def test_gemm_grid(sb_handle, name, dtype):
    sb_handle.set_kernel(_lib.KERNEL_AUTO)
    for case in GRIDS[name]:
        verify_gemm(sb_handle, dtype, *case, batch_type=strided)


# Every kernel family on shapes that exercise edge tiles, K tails and all four
# transpose combinations, whatever the heuristic would have picked.
FAMILY_SHAPES = [(1, 1, 1), (7, 9, 5), (33, 65, 31), (64, 64, 64), (129, 127, 70), (200, 260, 133), (256, 128, 96)]


@pytest.mark.parametrize("dtype", DTYPES, ids=lambda d: np.dtype(d).name)
@pytest.mark.parametrize("family", ["LARGE", "SMALL", "HUGE", "NAIVE", "TINY", "MEDIUM"])
def test_every_kernel_family(sb_handle, family, dtype):
    sb_handle.set_kernel(getattr(_lib, "KERNEL_" + family))
    try:
        if family == "MEDIUM":  # its second tile (192 x 192): parameter bit 0
            sb_handle.set_kernel(_lib.KERNEL_MEDIUM, 1)
            for (m, n, k), ta, tb in itertools.product([(200, 260, 133), (385, 193, 70), (192, 384, 64)], "nt", "nt"):
                verify_gemm(sb_handle, dtype, 3, 1, m, n, k, ta, tb, 1.5, 0.5, 2, 2, 2)
                assert sb_handle.last_kernel() == _lib.KERNEL_MEDIUM
            sb_handle.set_kernel(_lib.KERNEL_MEDIUM, 0)
        for (m, n, k), ta, tb, (off, ldm) in itertools.product(FAMILY_SHAPES, "nt", "nt", [(0, 1), (3, 2)]):
            for alpha, beta in ((1.5, 0.5), (-2.0, 0.0)):
                verify_gemm(sb_handle, dtype, off, 1, m, n, k, ta, tb, alpha, beta, ldm, ldm, ldm)
            assert sb_handle.last_kernel() == getattr(_lib, "KERNEL_" + family)
    finally:
        sb_handle.set_kernel(_lib.KERNEL_AUTO)


@pytest.mark.parametrize("dtype", DTYPES, ids=lambda d: np.dtype(d).name)
def test_k_zero_and_empty(sb_handle, dtype):
    # K == 0 leaves beta*C (the K loops of gemm_local.hpp:451/:481 do nothing,
    # the epilogue still runs); M == 0 or N == 0 touches nothing.
    verify_gemm(sb_handle, dtype, 0, 1, 8, 9, 0, "n", "n", 1.5, 0.5, 1, 1, 1)
    verify_gemm(sb_handle, dtype, 0, 1, 8, 9, 0, "t", "n", 1.5, 0.0, 1, 1, 1)
    h = sb_handle
    c = sb.helper.allocate(h, dtype, 16)
    a = sb.helper.allocate(h, dtype, 16)
    sb._gemm(h, "n", "n", 0, 4, 4, 1.0, a, 1, a, 4, 0.0, c, 1)
    sb._gemm(h, "n", "n", 4, 0, 4, 1.0, a, 4, a, 4, 0.0, c, 4)
    assert h.last_kernel() == -1
    h.wait()


@pytest.mark.parametrize("dtype", DTYPES, ids=lambda d: np.dtype(d).name)
def test_beta_zero_never_reads_c(sb_handle, dtype):
    # C full of NaN must not leak into the result when beta == 0
    # (gemm_interface.hpp:92-99, gemm_ref.hpp:245-249).
    h = sb_handle
    m, n, k = 70, 45, 33
    rng = np.random.default_rng(5)
    a = rng.uniform(-2, 5, m * k).astype(dtype)
    b = rng.uniform(-2, 5, k * n).astype(dtype)
    c = np.full(m * n, np.nan, dtype=dtype)
    for fam in (_lib.KERNEL_AUTO, _lib.KERNEL_LARGE, _lib.KERNEL_NAIVE, _lib.KERNEL_SPLITK):
        h.set_kernel(fam, 3 if fam == _lib.KERNEL_SPLITK else 0)
        a_g, b_g, c_g = (sb.helper.allocate(h, dtype, x.size) for x in (a, b, c))
        for host, dev in ((a, a_g), (b, b_g), (c, c_g)):
            sb.helper.copy_to_device(h, host, dev)
        sb._gemm(h, "n", "n", m, n, k, 2.0, a_g, m, b_g, k, 0.0, c_g, m)
        out = np.empty_like(c)
        h.wait(sb.helper.copy_to_host(h, c_g, out))
        assert np.isfinite(out).all()
    h.set_kernel(_lib.KERNEL_AUTO)


def test_alpha_zero_semantics(sb_handle):
    # alpha == 0, beta == 0 through _gemm multiplies (NaN * 0 stays NaN) because
    # stride_c == 0 != ldc*n routes to _scal_matrix (gemm_interface.hpp:129-137,
    # blas1_interface.hpp:498-506); through _gemm_batched with ldc == m the
    # vector _scal stores zeros (blas1_interface.hpp:444-448).
    h = sb_handle
    m = n = 8
    c = np.arange(m * n, dtype=np.float32)
    c[5] = np.nan
    a_g = sb.helper.allocate(h, np.float32, m * n)
    c_g = sb.helper.allocate(h, np.float32, 2 * m * n)
    host2 = np.concatenate([c, c])
    out = np.empty_like(host2)

    sb.helper.copy_to_device(h, host2, c_g)
    sb._gemm(h, "x", "y", m, n, 4, 0.0, a_g, m, a_g, m, 0.0, c_g, m)  # trans flags are not looked at
    h.wait(sb.helper.copy_to_host(h, c_g, out))
    assert np.isnan(out[5]) and (out[:m * n][np.arange(m * n) != 5] == 0).all()
    assert np.array_equal(out[m * n:], host2[m * n:], equal_nan=True)

    sb.helper.copy_to_device(h, host2, c_g)
    sb._gemm_batched(h, "n", "n", m, n, 4, 0.0, a_g, m, a_g, m, 0.0, c_g, m, 2)
    h.wait(sb.helper.copy_to_host(h, c_g, out))
    assert (out == 0).all()


def test_argument_errors(sb_handle):
    # std::invalid_argument cases of gemm_interface.hpp:144-165
    h = sb_handle
    buf = sb.helper.allocate(h, np.float32, 4096)
    with pytest.raises(ValueError, match="invalid _TransA"):
        sb._gemm(h, "x", "n", 4, 4, 4, 1.0, buf, 4, buf, 4, 0.0, buf, 4)
    with pytest.raises(ValueError, match="invalid _TransB"):
        sb._gemm(h, "N", "q", 4, 4, 4, 1.0, buf, 4, buf, 4, 0.0, buf, 4)
    with pytest.raises(ValueError, match="invalid _stridec"):
        sb._gemm_strided_batched(h, "n", "n", 4, 4, 4, 1.0, buf, 4, 16, buf, 4, 16, 0.0, buf, 4, 15, 2)
    with pytest.raises(ValueError, match="invalid _stridea"):
        sb._gemm_strided_batched(h, "n", "n", 4, 4, 4, 1.0, buf, 4, -1, buf, 4, 16, 0.0, buf, 4, 16, 2)
    with pytest.raises(ValueError, match="invalid _strideb"):
        sb._gemm_strided_batched(h, "n", "n", 4, 4, 4, 1.0, buf, 4, 16, buf, 4, -16, 0.0, buf, 4, 16, 2)
    # batch == 1 skips the stride checks (gemm_interface.hpp:158)
    sb._gemm_strided_batched(h, "C", "T", 4, 4, 4, 1.0, buf, 4, -1, buf, 4, -1, 0.0, buf, 4, 0, 1)
    h.wait()


def test_cpp_sample_through_compat_header():
    """samples/gemm.cpp (the reference's sample shape over include/portblas_compat.hpp)."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "samples", "gemm_sample")
    assert os.path.exists(exe), "build it with `make -C samples` (or __graft_entry__.build())"
    for args in ([], ["1024"]):
        out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "PASS" in out.stdout


def test_cpp_symm_trsm_sample_through_compat_header():
    """samples/symm_trsm.cpp: blas::_symm with the reference sample's arguments (samples/symm.cpp:18-26) and blas::_trsm."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "samples", "symm_trsm_sample")
    assert os.path.exists(exe), "build it with `make -C samples` (or __graft_entry__.build())"
    for args in ([], ["600"]):
        out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "PASS" in out.stdout


@pytest.mark.parametrize("offset", [0, 4])
def test_operand_tiles_straight_into_lds(sb_handle, offset):
    # fp32, A 'n' and B 't', 16-byte-loadable operands: the 64x64 tile stages its
    # operands global -> LDS without registers.  Same matrix instructions in the same order as the
    # register-staged kernel: whole buffers equal the oracle bit for bit (verify_gemm), on interior tiles,
    # on tiles that hang over both edges, with padded leading dimensions, with and without reading C, for
    # a batch, and for K ranges shorter and longer than the three stages.
    h = sb_handle
    h.set_kernel(_lib.KERNEL_SMALL)
    try:
        # the last three: K tails (one more step through registers after the streamed steps)
        for (m, n, k) in [(64, 64, 128), (128, 192, 160), (200, 132, 1024), (1024, 1024, 256), (68, 60, 96 + 32),
                          (200, 132, 1000), (128, 64, 129), (64, 128, 191)]:
            for alpha, beta, ldm in ((1.5, 0.5, 1), (2.0, 0.0, 3)):
                verify_gemm(h, np.float32, offset, 1, m, n, k, "n", "t", alpha, beta, ldm, ldm, ldm)
                assert h.last_kernel() == _lib.KERNEL_SMALL
        verify_gemm(h, np.float32, offset, 5, 128, 64, 256, "n", "t", 1.5, 0.5, 1, 1, 1)
        # K slabs of a split on this tile take the same kernel (the last slab ends in a tail)
        h.set_kernel(_lib.KERNEL_SPLITK, 3 | (1 << 16))
        verify_gemm(h, np.float32, offset, 1, 256, 128, 1000, "n", "t", 1.5, 0.5, 1, 1, 1)
        assert h.last_splitk()[0] == 3
    finally:
        h.set_kernel(_lib.KERNEL_AUTO)


@pytest.mark.parametrize("dtype", DTYPES, ids=lambda d: np.dtype(d).name)
def test_short_k_products_walk_their_tiles(sb_handle, dtype):
    # One product with K = 64 and more 64 x 64 tiles of C than the chip holds workgroups twice over: resident
    # workgroups walk the tiles with the next tile's operand panels in flight (the batch-walking kernel with
    # tile addressing).  Same k-ordered chain per element as the tile kernels: whole buffers (the padding of a
    # multiplied leading dimension included) must equal the oracle bit for bit, for every transpose, with C
    # read and not read, at an offset that keeps the operands 16-byte aligned.
    sb_handle.set_kernel(_lib.KERNEL_AUTO)
    v = 16 // np.dtype(dtype).itemsize
    m, n = (4096, 2048) if dtype == np.float32 else (2048, 2048)
    for ta, tb in itertools.product("nt", "nt"):
        verify_gemm(sb_handle, dtype, 0, 1, m, n, 64, ta, tb, 1.5, 0.0, 1, 1, 1)
        assert sb_handle.last_kernel() == _lib.KERNEL_STREAM
        verify_gemm(sb_handle, dtype, 2 * v, 1, m, n + 64, 64, ta, tb, -1.0, 0.75, 2, 1, 2)
        assert sb_handle.last_kernel() == _lib.KERNEL_STREAM
    # what the walk does not take keeps the tile kernels: a ragged edge, another K, unaligned operands, too few tiles
    for args in ((0, 1, m - 4, n, 64), (0, 1, m, n, 96), (1, 1, m, n, 64), (0, 1, 1024, 1024, 64)):
        verify_gemm(sb_handle, dtype, *args, "n", "n", 1.0, 0.5, 1, 1, 1)
        assert sb_handle.last_kernel() != _lib.KERNEL_STREAM


@pytest.mark.parametrize("dtype", DTYPES, ids=lambda d: np.dtype(d).name)
def test_short_products_with_few_tiles_take_the_one_wave_tile(sb_handle, dtype):
    # One product with K <= 64 and at most one 64 x 64 tile per compute unit runs on the one-wave 32 x 32 tile
    # (four times as many workgroups over the chip).  Same chain per element: whole buffers equal the oracle bit
    # for bit, for every transpose, ragged edges and multiplied leading dimensions included.
    sb_handle.set_kernel(_lib.KERNEL_AUTO)
    for (m, n, k) in ((512, 512, 16), (100, 700, 30), (2048, 512, 49), (256, 250, 64), (33, 65, 7)):
        for ta, tb in itertools.product("nt", "nt"):
            verify_gemm(sb_handle, dtype, 0, 1, m, n, k, ta, tb, 1.5, 0.5, 1, 2, 1)
            assert sb_handle.last_kernel() == _lib.KERNEL_TINY
            assert sb_handle.last_splitk()[0] == 1
    # a single tile, a longer K or a batch keep their own families
    for args in ((1, 60, 60, 40), (1, 512, 512, 96), (3, 200, 200, 32)):
        verify_gemm(sb_handle, dtype, 0, *args, "n", "t", 1.0, 0.0, 1, 1, 1)
        assert sb_handle.last_kernel() != _lib.KERNEL_TINY
