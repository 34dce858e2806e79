"""blas::_gemm_batched for the 16-bit pairs (half -> half, half -> float, bfloat16 -> float), both batch
layouts: the reference's grid test/unittest/blas3/blas3_gemm_batched_test.cpp:48-66 (instantiated for
every type pair, src/interface/blas3/gemm.cpp.in:56-79).

Checks per case (inputs exactly representable in the storage type, the reduced-precision recipe of
test/unittest/joint_matrix/joint_matrix_common.hpp:93-111):
  * the reference's predicate for the OUTPUT type against the oracle's fp32 chain on the same values;
  * the fp32 accumulation bound (K + 2) * 2^-24 * (|alpha| sum|a||b| + |beta||c|) around the float64
    result (+ one binary16 rounding for half output);
  * nothing outside the m x n windows changes;
  * when the lanes-over-batch kernel ran (small interleaved items) the result IS the oracle's fp32 chain,
    rounded once: bit for bit.
"""
import itertools

import numpy as np
import pytest

import oracle
import sycl_blas_amd as sb
from sycl_blas_amd import _lib, helper

from common import from_interleaved, interleaved, seeded, strided, to_interleaved
from test_gemm_half_gpu import PAIRS, make_inputs

pytestmark = pytest.mark.gpu


def verify_batched16(h, pair, offset, batch, m, n, k, ta, tb, alpha, beta, lda_mul, ldb_mul, ldc_mul, batch_type):
    storage, out_name = pair
    rng = seeded("half-batched", pair, offset, batch, m, n, k, ta, tb, alpha, beta, lda_mul, ldb_mul, ldc_mul,
                 int(batch_type))
    lda = (k if ta != "n" else m) * lda_mul
    ldb = (n if tb != "n" else k) * ldb_mul
    ldc = m * ldc_mul
    size_a, size_b, size_c = m * k * lda_mul, k * n * ldb_mul, m * n * ldc_mul
    a_bits, a_f = make_inputs(rng, batch * size_a + offset, storage)
    b_bits, b_f = make_inputs(rng, batch * size_b + offset, storage)
    if out_name == "float16":
        c_h = rng.uniform(-2.0, 5.0, batch * size_c + offset).astype(np.float16)
        c_dev, c_f = c_h.view(np.uint16).copy(), c_h.astype(np.float32)
    else:
        c_f = rng.uniform(-2.0, 5.0, batch * size_c + offset).astype(np.float32)
        c_dev = c_f.copy()

    inter = batch > 1 and batch_type == interleaved
    a_d, b_d, c_d = a_bits, b_bits, c_dev
    if inter:  # blas3_gemm_common.hpp:172-180
        a_d = to_interleaved(a_bits, offset, lda, m if ta == "t" else k, batch)
        b_d = to_interleaved(b_bits, offset, ldb, k if tb == "t" else n, batch)
        c_d = to_interleaved(c_dev, offset, ldc, n, batch)

    a_g = helper.allocate(h, storage, a_d.size)
    b_g = helper.allocate(h, storage, b_d.size)
    c_g = helper.allocate(h, out_name, c_d.size)
    try:
        deps = [helper.copy_to_device(h, a_d, a_g), helper.copy_to_device(h, b_d, b_g), helper.copy_to_device(h, c_d, c_g)]
        ev = sb._gemm_batched(h, ta, tb, m, n, k, alpha, a_g + offset, lda, b_g + offset, ldb, beta, c_g + offset, ldc,
                              batch, batch_type, deps)
        h.wait(ev)
        kernel = h.last_kernel()
        got_dev = np.empty_like(c_d)
        h.wait(helper.copy_to_host(h, c_g, got_dev))
        h.wait()
    finally:
        for buf in (a_g, b_g, c_g):
            helper.deallocate(buf)
    if inter:
        got_dev = from_interleaved(got_dev, offset, ldc, n, batch)

    what = (f"{storage}->{out_name} off={offset} batch={batch} {m}x{n}x{k} {ta}{tb} a={alpha} b={beta} "
            f"{'interleaved' if inter else 'strided'}")
    # oracle: the fp32 chain on the same values (what the reference's test compares with, in float)
    want = c_f.copy()
    assert oracle.gemm_backend(ta, tb, m, n, k, alpha, a_f, lda, size_a, b_f, ldb, size_b, beta, want, ldc, size_c,
                               batch=batch, a_off=offset, b_off=offset, c_off=offset) == 0
    if out_name == "float16":
        got_f = got_dev.view(np.float16).astype(np.float32)
        want_out = want.astype(np.float16).astype(np.float32)
    else:
        got_f, want_out = got_dev, want
    assert oracle.compare_vectors(got_f, want_out, out_name) == -1, what
    if kernel in (_lib.KERNEL_INTERLEAVED, _lib.KERNEL_NAIVE):
        assert np.array_equal(got_f.view(np.uint32), want_out.view(np.uint32)), what + ": not the fp32 chain"

    # accumulation bound around the float64 result, and untouched padding
    a64, b64, c64 = a_f.astype(np.float64), b_f.astype(np.float64), c_f.astype(np.float64)
    isz = 8
    mask = np.ones(got_f.size, bool)
    every = (offset + np.arange(batch)[:, None, None] * size_c + np.arange(m)[None, :, None] + np.arange(n)[None, None, :] * ldc)
    mask[every.ravel()] = False
    # (long batches: the bound on a spread sample of the items -- their results were already compared with the chain)
    for z in (range(batch) if batch <= 64 else np.unique(np.linspace(0, batch - 1, 64).astype(np.int64))):
        ar, ac = (k, m) if ta != "n" else (m, k)
        br, bc = (n, k) if tb != "n" else (k, n)
        A = np.lib.stride_tricks.as_strided(a64[offset + z * size_a:], (ar, ac), (isz, lda * isz))
        B = np.lib.stride_tricks.as_strided(b64[offset + z * size_b:], (br, bc), (isz, ldb * isz))
        Cv = np.lib.stride_tricks.as_strided(c64[offset + z * size_c:], (m, n), (isz, ldc * isz))
        opa = A.T if ta != "n" else A
        opb = B.T if tb != "n" else B
        truth = alpha * (opa @ opb) + beta * Cv
        mag = abs(alpha) * (np.abs(opa) @ np.abs(opb)) + abs(beta) * np.abs(Cv)
        idx = offset + z * size_c + np.arange(m)[:, None] + np.arange(n)[None, :] * ldc
        bound = (k + 2) * 2.0 ** -24 * mag + 1e-30
        if out_name == "float16":
            bound = bound + 2.0 ** -11 * np.abs(truth) + 1e-7
        err = np.abs(got_f.astype(np.float64)[idx] - truth)
        assert (err <= bound).all(), f"{what}: error {err.max():.3e} beyond the fp32 accumulation bound"
    assert np.array_equal(got_dev[mask], c_dev[mask]), what + ": wrote outside the m x n windows"
    return kernel


# blas3_gemm_batched_test.cpp:48-66 (BetaNonZeroLDMultiplied): offsets {0, 33}, batch {1, 5}, m, n, k in
# {63, 128, 129}, every transpose, alpha 3, beta 7, leading-dimension multipliers 2 / 3 / 4, both layouts
REFERENCE_GRID = list(itertools.product([0, 33], [1, 5], [63, 128, 129], [63, 128, 129], [63, 128, 129], "nt", "nt",
                                        [3.0], [7.0], [2], [3], [4], [strided, interleaved]))


@pytest.mark.parametrize("pair", PAIRS, ids=lambda p: f"{p[0]}-{p[1]}")
def test_reference_batched_grid(sb_handle, pair):
    sb_handle.set_kernel(_lib.KERNEL_AUTO)
    seen = set()
    for case in REFERENCE_GRID:
        seen.add(verify_batched16(sb_handle, pair, *case))
    assert _lib.KERNEL_INTERLEAVED in seen


@pytest.mark.parametrize("pair", PAIRS, ids=lambda p: f"{p[0]}-{p[1]}")
def test_interleaved_items_big_enough_to_repack(sb_handle, pair):
    # Items with enough arithmetic per repacked byte leave the lanes-over-batch kernel for the
    # matrix-instruction kernels (repacked into padded per-item matrices and back); sizes that are not
    # multiples of a 16-byte vector, C read and not read, an element offset.
    sb_handle.set_kernel(_lib.KERNEL_AUTO)
    for (m, n, k, batch), (ta, tb) in itertools.product([(160, 128, 192, 5), (130, 257, 200, 3), (257, 129, 333, 2)],
                                                        itertools.product("nt", "nt")):
        kern = verify_batched16(sb_handle, pair, 0, batch, m, n, k, ta, tb, 1.5, 0.5, 2, 3, 2, interleaved)
        assert kern != _lib.KERNEL_INTERLEAVED
        verify_batched16(sb_handle, pair, 7, batch, m, n, k, ta, tb, -1.0, 0.0, 1, 1, 1, interleaved)
    assert verify_batched16(sb_handle, pair, 0, 4, 31, 200, 200, "n", "n", 1.0, 1.0, 1, 1, 1,
                            interleaved) == _lib.KERNEL_INTERLEAVED


@pytest.mark.parametrize("pair", PAIRS, ids=lambda p: f"{p[0]}-{p[1]}")
def test_interleaved_long_batches_take_the_big_register_block(sb_handle, pair):
    # long batches run an 8 x 8 block of C per lane (see the fp32 / fp64 twin of this test)
    for (m, n, k, batch), (ta, tb) in itertools.product([(16, 16, 8, 33000), (17, 22, 5, 15000)],
                                                        [("n", "n"), ("t", "n"), ("n", "t"), ("t", "t")]):
        assert verify_batched16(sb_handle, pair, 0, batch, m, n, k, ta, tb, 1.5, 0.5, 1, 1, 1, interleaved) == _lib.KERNEL_INTERLEAVED
    verify_batched16(sb_handle, pair, 3, 22000, 16, 24, 7, "n", "t", -1.0, 0.0, 2, 3, 2, interleaved)


@pytest.mark.parametrize("pair", PAIRS, ids=lambda p: f"{p[0]}-{p[1]}")
def test_strided_batches_of_tiny_items(sb_handle, pair):
    # items of at most 16 x 16 in a strided batch run on lanes (one element or a 4 x 4 block of C per lane), not on a
    # 64 x 64 matrix-instruction tile per item: the result is the fp32 chain of the widened inputs, rounded once
    for (m, n, k, batch), (ta, tb) in itertools.product([(4, 4, 4, 300), (8, 8, 8, 64), (16, 9, 40, 17), (16, 16, 16, 40), (13, 11, 7, 29)],
                                                        [("n", "n"), ("t", "n"), ("n", "t"), ("t", "t")]):
        assert verify_batched16(sb_handle, pair, 3, batch, m, n, k, ta, tb, 1.5, 0.5, 2, 1, 2, strided) == _lib.KERNEL_NAIVE
        verify_batched16(sb_handle, pair, 0, batch, m, n, k, ta, tb, -1.0, 0.0, 1, 1, 1, strided)


@pytest.mark.parametrize("pair", PAIRS, ids=lambda p: f"{p[0]}-{p[1]}")
def test_batched_alpha_zero_and_errors(sb_handle, pair):
    # blas3_gemm_batched_test.cpp:68-104: alpha == 0 scales every item's window by beta
    for bt in (strided, interleaved):
        verify_batched16(sb_handle, pair, 0, 5, 63, 63, 63, "n", "t", 0.0, 7.0, 2, 3, 4, bt)
    storage, out_name = pair
    buf = helper.allocate(sb_handle, storage, 64)
    out = helper.allocate(sb_handle, out_name, 64)
    with pytest.raises(ValueError, match="invalid _TransA"):
        sb._gemm_batched(sb_handle, "x", "n", 4, 4, 4, 1.0, buf, 4, buf, 4, 0.0, out, 4, 2, strided)
    with pytest.raises(ValueError, match="invalid _TransB"):
        sb._gemm_batched(sb_handle, "n", "x", 4, 4, 4, 1.0, buf, 4, buf, 4, 0.0, out, 4, 2, interleaved)
