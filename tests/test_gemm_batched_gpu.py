"""GPU parity tests for blas::_gemm_batched and blas::_gemm_strided_batched: the
reference's grids (test/unittest/blas3/blas3_gemm_batched_test.cpp:29-147)."""
import itertools

import numpy as np
import pytest

from sycl_blas_amd import _lib

from common import interleaved, strided, verify_gemm, verify_gemm_strided

pytestmark = pytest.mark.gpu

DTYPES = [np.float32, np.float64]


def grid(*axes):
    return list(itertools.product(*axes))


# (offset, batch, m, n, k, transa, transb, alpha, beta, lda_mul, ldb_mul, ldc_mul, batch_type)
BETA_NONZERO_LD_MATCH = grid([0, 33], [5], [63, 128], [63, 128], [63, 128], "nt", "nt", [3.0], [7.0], [1], [1], [1], [strided])
BETA_NONZERO_LD_MULTIPLIED = grid([0, 33], [1, 5], [63, 128, 129], [63, 128, 129], [63, 128, 129], "nt", "nt", [3.0], [7.0],
                                  [2], [3], [4], [strided, interleaved])
BETA_NONZERO_LD_MATCH_ALPHA0 = grid([0], [5], [128], [128], [128], "nt", "nt", [0.0], [7.0], [1], [1], [1], [strided])
BETA_NONZERO_LD_MULTIPLIED_ALPHA0 = grid([0, 33], [5], [63], [63], [63], "nt", "nt", [0.0], [7.0], [2], [3], [4], [strided])

GRIDS = {
    "BetaNonZeroLDMatch": BETA_NONZERO_LD_MATCH,
    "BetaNonZeroLDMultiplied": BETA_NONZERO_LD_MULTIPLIED,
    "BetaNonZeroLDMatchAlpha0": BETA_NONZERO_LD_MATCH_ALPHA0,
    "BetaNonZeroLDMultipliedAlpha0": BETA_NONZERO_LD_MULTIPLIED_ALPHA0,
}

# (offset, batch, m, n, k, transa, transb, alpha, beta, lda_mul, ldb_mul, ldc_mul, sa_mul, sb_mul, sc_mul)
DEFAULT_GEMM_AND_GEMM_BATCHED = grid([0, 33], [1, 5], [63, 128], [63, 128], [63, 128], "nt", "nt", [3.0], [7.0], [1], [1], [1],
                                     [1], [1], [1])
ALL_STRIDED_BATCHED = grid([0, 33], [5], [63], [63], [128], "nt", "nt", [3.0, 0.0], [7.0, 1.0, 0.0], [2], [3], [4],
                           [0, 1, 2], [0, 1, 2], [1, 3])

STRIDED_GRIDS = {
    "DefaultGemmAndGemmBatched": DEFAULT_GEMM_AND_GEMM_BATCHED,
    "AllStridedBatched": ALL_STRIDED_BATCHED,
}


@pytest.mark.parametrize("dtype", DTYPES, ids=lambda d: np.dtype(d).name)
@pytest.mark.parametrize("name", list(GRIDS))
def test_gemm_batched_grid(sb_handle, name, dtype):
    sb_handle.set_kernel(_lib.KERNEL_AUTO)
    for case in GRIDS[name]:
        *args, batch_type = case
        verify_gemm(sb_handle, dtype, *args, batch_type=batch_type)


@pytest.mark.parametrize("dtype", DTYPES, ids=lambda d: np.dtype(d).name)
@pytest.mark.parametrize("name", list(STRIDED_GRIDS))
def test_gemm_strided_batched_grid(sb_handle, name, dtype):
    sb_handle.set_kernel(_lib.KERNEL_AUTO)
    for case in STRIDED_GRIDS[name]:
        verify_gemm_strided(sb_handle, dtype, *case)


@pytest.mark.parametrize("dtype", DTYPES, ids=lambda d: np.dtype(d).name)
@pytest.mark.parametrize("family", ["LARGE", "SMALL", "HUGE", "NAIVE", "TINY"])
def test_batched_every_kernel_family(sb_handle, family, dtype):
    sb_handle.set_kernel(getattr(_lib, "KERNEL_" + family))
    try:
        for ta, tb in itertools.product("nt", "nt"):
            verify_gemm_strided(sb_handle, dtype, 5, 3, 70, 33, 41, ta, tb, 1.5, 0.5, 2, 1, 3, 1, 0, 2)
            verify_gemm_strided(sb_handle, dtype, 0, 4, 64, 64, 64, ta, tb, 1.0, 0.0, 1, 1, 1, 1, 1, 1)
    finally:
        sb_handle.set_kernel(_lib.KERNEL_AUTO)


def test_config3_shape_sample(sb_handle):
    # BASELINE config #3 at reduced batch: strided-batched SGEMM of 64x64x64 items.
    verify_gemm_strided(sb_handle, np.float32, 0, 64, 64, 64, 64, "n", "n", 1.0, 0.0, 1, 1, 1, 1, 1, 1)
    assert sb_handle.last_kernel() == _lib.KERNEL_SMALL


@pytest.mark.parametrize("dtype", DTYPES, ids=lambda d: np.dtype(d).name)
def test_batches_of_tiny_matrices(sb_handle, dtype):
    # Shapes the 64x64 tile would mostly pad: up to 16x16 takes one lane per C element or, from 512 bytes of C
    # per item on, a 4x4 block of C per lane (both reported as the NAIVE family), up to 32x32 the one-wave 32x32
    # tile; all keep the k-ordered chain (bitwise against the oracle).
    sb_handle.set_kernel(_lib.KERNEL_AUTO)
    for (m, n, k, batch), want in [((4, 4, 4, 300), _lib.KERNEL_NAIVE), ((8, 8, 8, 64), _lib.KERNEL_NAIVE),
                                   ((16, 9, 40, 17), _lib.KERNEL_NAIVE), ((16, 16, 16, 40), _lib.KERNEL_NAIVE), ((13, 11, 7, 29), _lib.KERNEL_NAIVE),
                                   ((17, 16, 16, 33), _lib.KERNEL_TINY),
                                   ((24, 24, 24, 50), _lib.KERNEL_TINY), ((32, 32, 100, 9), _lib.KERNEL_TINY),
                                   ((33, 32, 32, 9), _lib.KERNEL_SMALL)]:
        for ta, tb in itertools.product("nt", "nt"):
            verify_gemm_strided(sb_handle, dtype, 3, batch, m, n, k, ta, tb, 1.5, 0.5, 2, 1, 1, 1, 1, 2)
            assert sb_handle.last_kernel() == want


@pytest.mark.parametrize("dtype", DTYPES, ids=lambda d: np.dtype(d).name)
def test_interleaved_items_big_enough_to_repack(sb_handle, dtype):
    # Interleaved items with enough arithmetic per byte (about 128^3 and more) are repacked into per-item matrices and run on the
    # strided kernels; the result lands back in the interleaved buffer, padding rows untouched
    # (verify_gemm compares the whole buffer with the oracle bit for bit).
    sb_handle.set_kernel(_lib.KERNEL_AUTO)
    for (m, n, k, batch), (ta, tb) in itertools.product([(160, 128, 192, 5), (130, 257, 200, 3), (192, 192, 192, 9)],
                                                        itertools.product("nt", "nt")):
        verify_gemm(sb_handle, dtype, 0, batch, m, n, k, ta, tb, 1.5, 0.5, 2, 3, 2, batch_type=interleaved)
        assert sb_handle.last_kernel() != _lib.KERNEL_INTERLEAVED
        verify_gemm(sb_handle, dtype, 7, batch, m, n, k, ta, tb, -1.0, 0.0, 1, 1, 1, batch_type=interleaved)
    # just below the threshold the lanes-over-batch kernel still runs
    verify_gemm(sb_handle, dtype, 0, 4, 31, 200, 200, "n", "n", 1.0, 1.0, 1, 1, 1, batch_type=interleaved)
    assert sb_handle.last_kernel() == _lib.KERNEL_INTERLEAVED


@pytest.mark.parametrize("dtype", DTYPES, ids=lambda d: np.dtype(d).name)
def test_long_batches_of_single_tile_items(sb_handle, dtype):
    # More 64x64x64-class items than a chip holds workgroups: resident workgroups walk the batch with the
    # next item's operands requested ahead.  Same chain per element as the tile kernel: whole buffers
    # (padding between items and rows included) must equal the oracle bit for bit.  Batch sizes that do
    # and do not divide over the workgroups, edges along M and N, padded leading dimensions, a
    # broadcast operand, C read and not read.
    sb_handle.set_kernel(_lib.KERNEL_AUTO)
    v = 16 // np.dtype(dtype).itemsize
    for ta, tb in itertools.product("nt", "nt"):
        verify_gemm_strided(sb_handle, dtype, 0, 1100, 64, 64, 64, ta, tb, 1.0, 0.0, 1, 1, 1, 1, 1, 1)
        assert sb_handle.last_kernel() == _lib.KERNEL_STREAM
        verify_gemm_strided(sb_handle, dtype, v, 2051, 64, 64, 64, ta, tb, 1.5, 0.5, 1, 1, 1, 1, 1, 1)
        assert sb_handle.last_kernel() == _lib.KERNEL_STREAM
        verify_gemm_strided(sb_handle, dtype, 0, 1300, 60, 44, 64, ta, tb, -1.0, 7.0, 2, 1, 3, 1, 0, 2)
        assert sb_handle.last_kernel() == _lib.KERNEL_STREAM
        verify_gemm_strided(sb_handle, dtype, 2 * v, 1025, 4, 64, 64, ta, tb, 3.0, 0.0, 1, 2, 1, 2, 1, 1)
        assert sb_handle.last_kernel() == _lib.KERNEL_STREAM
    # what the walking kernel does not take keeps the tile kernel: unaligned operands, a vector that
    # would straddle the edge, another K
    verify_gemm_strided(sb_handle, dtype, 1, 1100, 64, 64, 64, "n", "n", 1.0, 0.0, 1, 1, 1, 1, 1, 1)
    assert sb_handle.last_kernel() == _lib.KERNEL_SMALL
    verify_gemm_strided(sb_handle, dtype, 0, 1100, 63, 64, 64, "n", "n", 1.0, 0.0, 1, 1, 1, 1, 1, 1)
    assert sb_handle.last_kernel() == _lib.KERNEL_SMALL
    verify_gemm_strided(sb_handle, dtype, 0, 1100, 64, 64, 48, "n", "n", 1.0, 0.0, 1, 1, 1, 1, 1, 1)
    assert sb_handle.last_kernel() == _lib.KERNEL_SMALL


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
def test_interleaved_long_batches_take_the_big_register_block(sb_handle, dtype):
    # Batches long enough to fill the chip with a quarter of the lanes run an 8 x 8 block of C per lane (the
    # grids above, with their short batches, stay on 4 x 4): whole items, items with ragged edges in both
    # dimensions, every transpose, beta == 0 and != 0, padded leading dimensions.  Same fma chain per element.
    for (m, n, k, batch), (ta, tb) in itertools.product([(16, 16, 8, 33000), (17, 22, 5, 15000), (8, 16, 3, 66000)],
                                                        [("n", "n"), ("t", "n"), ("n", "t"), ("t", "t")]):
        verify_gemm(sb_handle, dtype, 0, batch, m, n, k, ta, tb, 1.5, 0.5, 1, 1, 1, batch_type=interleaved)
        assert sb_handle.last_kernel() == _lib.KERNEL_INTERLEAVED
    verify_gemm(sb_handle, dtype, 3, 22000, 16, 24, 7, "n", "t", -1.0, 0.0, 2, 3, 2, batch_type=interleaved)


@pytest.mark.parametrize("dtype", DTYPES, ids=lambda d: np.dtype(d).name)
def test_interleaved_alpha_zero_scales_every_item_once(sb_handle, dtype):
    # alpha == 0 on an interleaved batch: every item's m x n window is multiplied by beta once, the
    # padding rows and the other buffers are untouched (verify_gemm compares the whole buffer with the
    # oracle's per-item scaling).  A deliberate difference: the reference has no interleaved branch in
    # its alpha == 0 shortcut (src/interface/gemm_interface.hpp:112-139) -- _gemm_batched passes
    # _stridec = 0 for that layout (:209-220), so it scales the strided-addressed m x n window at C
    # `batch` times and leaves the other items as they were; its tests only cover alpha == 0 with the
    # strided layout (test/unittest/blas3/blas3_gemm_batched_test.cpp:68-104).
    sb_handle.set_kernel(_lib.KERNEL_AUTO)
    for beta, off, ldm in ((7.0, 0, 1), (0.0, 33, 4), (1.0, 0, 2)):
        verify_gemm(sb_handle, dtype, off, 5, 63, 17, 9, "n", "t", 0.0, beta, 2, 3, ldm, batch_type=interleaved)
        assert sb_handle.last_kernel() == -1  # no GEMM kernel ran
