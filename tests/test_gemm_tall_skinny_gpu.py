"""GPU parity tests for the tall-skinny (split-K) path: the reference's grids
(test/unittest/blas3/blas3_gemm_tall_skinny_test.cpp:29-103)."""
import itertools

import numpy as np
import pytest

from sycl_blas_amd import _lib

from common import verify_gemm

pytestmark = pytest.mark.gpu

DTYPES = [np.float32, np.float64]


def grid(*axes):
    return list(itertools.product(*axes))


# (offset, batch, m, n, k, transa, transb, alpha, beta, lda_mul, ldb_mul, ldc_mul)
BETA_NONZERO_LD_MATCH = grid([0], [1], [7, 65], [9, 126], [2049], "nt", "nt", [1.5], [0.5], [1], [1], [1])
BETA_NONZERO_LD_MULTIPLIED = grid([0], [1], [7, 65], [9, 126], [2049], "nt", "nt", [1.5], [0.5], [2], [3], [4])
BETA_ZERO = grid([0], [1], [7], [9], [1026], "nt", "nt", [1.5], [0.0], [1], [1], [1])
OFFSET_NONZERO = grid([10], [1], [7], [9], [1026], "nt", "nt", [1.5], [0.5], [1], [1], [1])

GRIDS = {
    "BetaNonZeroLDMatch": BETA_NONZERO_LD_MATCH,
    "BetaNonZeroLDMultiplied": BETA_NONZERO_LD_MULTIPLIED,
    "BetaZero": BETA_ZERO,
    "OffsetNonZero": OFFSET_NONZERO,
}


@pytest.mark.parametrize("dtype", DTYPES, ids=lambda d: np.dtype(d).name)
@pytest.mark.parametrize("name", list(GRIDS))
def test_tall_skinny_grid(sb_handle, name, dtype):
    sb_handle.set_kernel(_lib.KERNEL_AUTO)
    saw_split = False
    for case in GRIDS[name]:
        verify_gemm(sb_handle, dtype, *case)
        saw_split |= sb_handle.last_kernel() == _lib.KERNEL_SPLITK
    # These shapes (a handful of tiles, K in the thousands) must take the split
    # path.  The reference keeps k <= 2048 in one final partial kernel
    # (portblas_handle.hpp:323-331); here the planner cuts K whenever M x N alone
    # cannot fill the chip, and verify_gemm replays the reported cut in the oracle.
    assert saw_split


@pytest.mark.parametrize("dtype", DTYPES, ids=lambda d: np.dtype(d).name)
@pytest.mark.parametrize("depth", [2, 3, 7, 64])
def test_forced_split_depths(sb_handle, depth, dtype):
    # Any cube depth must reproduce the oracle's split-K form with the same cut,
    # on both partial-kernel tiles (64x64 and 128x128) and on ragged edges.
    sb_handle.set_kernel(_lib.KERNEL_SPLITK, depth)
    try:
        for (m, n, k), ta, tb in itertools.product([(7, 9, 300), (65, 126, 1000), (200, 130, 2049), (256, 256, 4096)],
                                                  "nt", "nt"):
            for alpha, beta, ldm in ((1.5, 0.5, 1), (2.0, 0.0, 3)):
                verify_gemm(sb_handle, dtype, 0, 1, m, n, k, ta, tb, alpha, beta, ldm, ldm, ldm)
                assert sb_handle.last_kernel() == _lib.KERNEL_SPLITK
    finally:
        sb_handle.set_kernel(_lib.KERNEL_AUTO)


def test_config4_shape_sample(sb_handle):
    # BASELINE config #4 at reduced K: SGEMM 256 x 256 x 65536 goes through split-K.
    verify_gemm(sb_handle, np.float32, 0, 1, 256, 256, 65536, "n", "n", 1.5, 0.5, 1, 1, 1)
    assert sb_handle.last_kernel() == _lib.KERNEL_SPLITK
    depth, k_chunk = sb_handle.last_splitk()
    assert depth > 1 and depth * k_chunk >= 65536


@pytest.mark.parametrize("dtype", DTYPES, ids=lambda d: np.dtype(d).name)
@pytest.mark.parametrize("tile", [1, 2, 6], ids=["64x64", "128x128", "96x96"])
def test_two_halves_in_one_workgroup(sb_handle, tile, dtype):
    # Depth 2 with both K halves inside one workgroup (parameter bit 24) must give the bits of the cube
    # route with two slabs -- (alpha*S0 + alpha*S1) + beta*C -- on interior tiles, ragged edges, odd
    # leading dimensions (element loads) and every transpose; a K that does not split into whole K
    # steps falls back to the cube route.
    sb_handle.set_kernel(_lib.KERNEL_SPLITK, 2 | (tile << 16) | (1 << 24))
    try:
        # the last two: halves that are not whole K steps (both sets then run the same K tail)
        for (m, n, k), ta, tb in itertools.product([(64, 64, 128), (128, 256, 512), (200, 130, 1024), (7, 9, 192),
                                                    (257, 63, 640), (200, 132, 1000), (64, 64, 136)], "nt", "nt"):
            for alpha, beta, ldm in ((1.5, 0.5, 1), (2.0, 0.0, 3)):
                verify_gemm(sb_handle, dtype, 0, 1, m, n, k, ta, tb, alpha, beta, ldm, ldm, ldm)
                assert sb_handle.last_kernel() == _lib.KERNEL_SPLITK
                assert sb_handle.last_splitk() == (2, k // 2)
                assert sb_handle.last_tile() == ({1: 64, 2: 128, 6: 96}[tile],) * 2 + (True,)
        # a K whose halves would not start 16-byte aligned takes the cube route
        verify_gemm(sb_handle, dtype, 0, 1, 65, 126, 1002, "n", "t", 1.5, 0.5, 1, 1, 1)
        assert sb_handle.last_splitk()[0] == 2 and sb_handle.last_splitk()[1] != 501
    finally:
        sb_handle.set_kernel(_lib.KERNEL_AUTO)


@pytest.mark.parametrize("dtype", DTYPES, ids=lambda d: np.dtype(d).name)
def test_four_quarters_in_one_workgroup(sb_handle, dtype):
    # Depth 4 with the four K quarters inside one workgroup of four one-wave sets on the 32 x 32 tile must give the
    # bits of the cube route with four slabs -- ((alpha*S0 + alpha*S1) + alpha*S2) + alpha*S3, then + beta*C.
    sb_handle.set_kernel(_lib.KERNEL_SPLITK, 4 | (4 << 16) | (1 << 24))
    try:
        for (m, n, k), ta, tb in itertools.product([(32, 32, 128), (128, 256, 512), (200, 130, 1024), (7, 9, 192),
                                                    (257, 63, 640), (100, 132, 1008), (64, 64, 144)], "nt", "nt"):
            for alpha, beta, ldm in ((1.5, 0.5, 1), (2.0, 0.0, 3)):
                verify_gemm(sb_handle, dtype, 0, 1, m, n, k, ta, tb, alpha, beta, ldm, ldm, ldm)
                assert sb_handle.last_kernel() == _lib.KERNEL_SPLITK
                assert sb_handle.last_splitk() == (4, k // 4)
                assert sb_handle.last_tile() == (32, 32, True)
        # quarters that would not start 16-byte aligned take the cube route
        verify_gemm(sb_handle, dtype, 0, 1, 65, 126, 1004, "n", "t", 1.5, 0.5, 1, 1, 1)
        assert sb_handle.last_splitk()[0] == 4 and sb_handle.last_tile() == (64, 64, False)
    finally:
        sb_handle.set_kernel(_lib.KERNEL_AUTO)


def test_planner_uses_the_halves_for_a_half_filled_chip(sb_handle):
    # 1024^3 in fp64: 256 tiles of 64x64 leave three of four workgroup slots per compute unit empty.
    verify_gemm(sb_handle, np.float64, 0, 1, 1024, 1024, 1024, "n", "n", 1.5, 0.5, 1, 1, 1)
    assert sb_handle.last_kernel() == _lib.KERNEL_SPLITK and sb_handle.last_splitk() == (2, 512)


@pytest.mark.parametrize("dtype", DTYPES, ids=lambda d: np.dtype(d).name)
@pytest.mark.parametrize("tile", [6, 7], ids=["96x96", "192x192"])
def test_forced_split_on_the_medium_tiles(sb_handle, tile, dtype):
    # K slabs through the cube on the 96 x 96 and 192 x 192 tiles (the 16x16x4 matrix instruction in fp32): every
    # transpose, K cuts that are and are not whole K steps, ragged edges, odd leading dimensions.
    try:
        for depth in (2, 3, 5):
            sb_handle.set_kernel(_lib.KERNEL_SPLITK, depth | (tile << 16))
            for (m, n, k), ta, tb in itertools.product([(192, 192, 2048), (300, 200, 2064), (97, 388, 1000)], "nt", "nt"):
                for alpha, beta, ldm in ((1.5, 0.5, 1), (2.0, 0.0, 3)):
                    verify_gemm(sb_handle, dtype, 0, 1, m, n, k, ta, tb, alpha, beta, ldm, ldm, ldm)
                    assert sb_handle.last_kernel() == _lib.KERNEL_SPLITK and sb_handle.last_splitk()[0] == depth
                    assert sb_handle.last_tile() == ((96, 96, False) if tile == 6 else (192, 192, False))
    finally:
        sb_handle.set_kernel(_lib.KERNEL_AUTO)


@pytest.mark.parametrize("dtype", DTYPES, ids=lambda d: np.dtype(d).name)
def test_forced_split_on_the_biggest_tile(sb_handle, dtype):
    # K slabs on the biggest tile family, every transpose (fp32 with A read along K takes a narrower tile
    # with a deeper K step there), K cuts that are and are not whole K steps of either tile, ragged edges.
    try:
        for depth in (3, 4):
            sb_handle.set_kernel(_lib.KERNEL_SPLITK, depth | (3 << 16))
            for (m, n, k), ta, tb in itertools.product([(256, 256, 4096), (300, 200, 2064), (512, 260, 1000)], "nt", "nt"):
                for alpha, beta, ldm in ((1.5, 0.5, 1), (2.0, 0.0, 3)):
                    verify_gemm(sb_handle, dtype, 0, 1, m, n, k, ta, tb, alpha, beta, ldm, ldm, ldm)
                    assert sb_handle.last_kernel() == _lib.KERNEL_SPLITK and sb_handle.last_splitk()[0] == depth
    finally:
        sb_handle.set_kernel(_lib.KERNEL_AUTO)
