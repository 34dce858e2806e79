"""GPU parity tests for the 16-bit storage pairs: half -> half, half -> float
(the reference's instantiations, cmake/CmakeFunctionHelper.cmake:148-159) and
bfloat16 -> float.

Method: the reference's reduced-precision test recipe
(test/unittest/joint_matrix/joint_matrix_common.hpp:93-111, test/blas_test.hpp:225-244):
make every input exactly representable in the storage type, then compare with a
float GEMM of the same values.  Products of two 16-bit values are exact in fp32
and the kernels accumulate in fp32, so:
  * float outputs must pass the reference's FLOAT predicate against the oracle's
    fp32 chain (float_comparison.hpp:100-143) and stay inside the fp32
    accumulation bound  |got - exact| <= (K + 2) * 2^-24 * (|alpha| sum|a||b| + |beta||c|);
  * half outputs add one rounding to binary16 (2^-11 relative) and must pass the
    reference's HALF predicate (float_comparison.hpp:123-158).
"""
import itertools

import numpy as np
import pytest

import oracle
import sycl_blas_amd as sb
from sycl_blas_amd import DeviceBuffer, _lib, helper

from common import seeded

pytestmark = pytest.mark.gpu


def to_bf16_bits(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)


def from_bf16_bits(b):
    return (b.astype(np.uint32) << np.uint32(16)).view(np.float32)


def make_inputs(rng, count, storage):
    """(bit patterns as stored on the device, the same values as float32)."""
    x = rng.uniform(-2.0, 5.0, count).astype(np.float32)
    if storage == "float16":
        h = x.astype(np.float16)
        return h.view(np.uint16), h.astype(np.float32)
    bits = to_bf16_bits(x)
    return bits, from_bf16_bits(bits)


def run_case(h, pair, m, n, k, ta, tb, alpha, beta, ldm=1, offset=0, batch=1, family=_lib.KERNEL_AUTO, param=0):
    storage, out_name = pair
    rng = seeded("half", pair, m, n, k, ta, tb, alpha, beta, ldm, offset, batch)
    lda = (k if ta != "n" else m) * ldm
    ldb = (n if tb != "n" else k) * ldm
    ldc = m * ldm
    sa, sb_, sc = m * k * ldm, k * n * ldm, m * n * ldm
    a_bits, a_f = make_inputs(rng, batch * sa + offset, storage)
    b_bits, b_f = make_inputs(rng, batch * sb_ + offset, storage)
    if out_name == "float16":
        c_h = rng.uniform(-2.0, 5.0, batch * sc + offset).astype(np.float16)
        c_dev, c_f = c_h.view(np.uint16).copy(), c_h.astype(np.float32)
    else:
        c_f = rng.uniform(-2.0, 5.0, batch * sc + offset).astype(np.float32)
        c_dev = c_f.copy()

    # oracle: fp32 chain on the same values (what the reference's test compares with)
    want = c_f.copy()
    st = oracle.gemm_backend(ta, tb, m, n, k, alpha, a_f, lda, sa, b_f, ldb, sb_, beta, want, ldc, sc, batch=batch,
                             a_off=offset, b_off=offset, c_off=offset)
    assert st == 0

    h.set_kernel(family, param)
    a_g = helper.allocate(h, storage, a_bits.size)
    b_g = helper.allocate(h, storage, b_bits.size)
    c_g = helper.allocate(h, out_name, c_dev.size)
    deps = [helper.copy_to_device(h, a_bits, a_g), helper.copy_to_device(h, b_bits, b_g),
            helper.copy_to_device(h, c_dev, c_g)]
    if batch == 1:
        ev = sb._gemm(h, ta, tb, m, n, k, alpha, a_g + offset, lda, b_g + offset, ldb, beta, c_g + offset, ldc, deps)
    else:
        ev = sb._gemm_batched(h, ta, tb, m, n, k, alpha, a_g + offset, lda, b_g + offset, ldb, beta, c_g + offset,
                              ldc, batch, sb.gemm_batch_type_t.strided, deps)
    h.wait(ev)
    got_dev = np.empty_like(c_dev)
    h.wait(helper.copy_to_host(h, c_g, got_dev))
    for buf in (a_g, b_g, c_g):
        helper.deallocate(buf)
    h.set_kernel(_lib.KERNEL_AUTO)

    what = f"{storage}->{out_name} {m}x{n}x{k} {ta}{tb} a={alpha} b={beta} ld*{ldm} off={offset} batch={batch}"
    # fp32 accumulation bound against the long-double result, item by item
    for z in range(batch):
        truth, mag = oracle.gemm_truth(ta != "n", tb != "n", m, n, k, alpha,
                                       a_f[offset + z * sa:].astype(np.float64), lda,
                                       b_f[offset + z * sb_:].astype(np.float64), ldb, beta,
                                       c_f[offset + z * sc:].astype(np.float64), ldc)
        idx = (offset + z * sc + np.arange(m)[:, None] + np.arange(n)[None, :] * ldc).T.ravel()  # column-major order
        if out_name == "float16":
            got = got_dev.view(np.float16).astype(np.float64)[idx]
            bound = (k + 2) * 2.0 ** -24 * mag + 2.0 ** -11 * np.abs(truth) + 1e-7
        else:
            got = got_dev.astype(np.float64)[idx]
            bound = (k + 2) * 2.0 ** -24 * mag + 1e-30
        err = np.abs(got - truth)
        assert (err <= bound).all(), f"{what}: error {err.max():.3e} beyond the fp32 accumulation bound"
    if out_name == "float16":
        got_all = got_dev.view(np.float16).astype(np.float32)
        assert oracle.compare_vectors(got_all, want.astype(np.float16).astype(np.float32), "float16") == -1, what
        # nothing outside the m x n windows may change
        mask = np.ones(got_all.size, bool)
        for z in range(batch):
            mask[(offset + z * sc + np.arange(m)[:, None] + np.arange(n)[None, :] * ldc).ravel()] = False
        assert np.array_equal(got_dev[mask], c_dev[mask]), what
    else:
        assert oracle.compare_vectors(got_dev, want, "float32") == -1, what
        mask = np.ones(got_dev.size, bool)
        for z in range(batch):
            mask[(offset + z * sc + np.arange(m)[:, None] + np.arange(n)[None, :] * ldc).ravel()] = False
        assert np.array_equal(got_dev[mask], c_dev[mask]), what


PAIRS = [("float16", "float16"), ("float16", "float32"), ("bfloat16", "float32")]


@pytest.mark.parametrize("pair", PAIRS, ids=lambda p: f"{p[0]}-{p[1]}")
def test_small_grids(sb_handle, pair):
    # blas3_gemm_test.cpp:29-84 shapes
    for m, n, k, ta, tb in itertools.product([11, 16, 32], [11, 32], [16, 17], "nt", "nt"):
        run_case(sb_handle, pair, m, n, k, ta, tb, 1.5, 1.5)
    for m, n, ta, tb in itertools.product([11, 32], [11, 32], "nt", "nt"):
        run_case(sb_handle, pair, m, n, 17, ta, tb, 1.5, 0.0, ldm=2)


@pytest.mark.parametrize("pair", PAIRS, ids=lambda p: f"{p[0]}-{p[1]}")
def test_large_offsets_and_batches(sb_handle, pair):
    # blas3_gemm_test.cpp:105-141 and blas3_gemm_batched_test.cpp:29-46 shapes (a slice)
    for m, n, k, ta, tb in itertools.product([253, 1024], [257], [253, 511], "nt", "nt"):
        run_case(sb_handle, pair, m, n, k, ta, tb, 1.0, 1.0)
    for off, m, k in itertools.product([1, 10], [16, 63], [17, 63]):
        run_case(sb_handle, pair, m, 63, k, "n", "n", 1.0, 1.0, ldm=2, offset=off)
    for off, ta, tb in itertools.product([0, 33], "nt", "nt"):
        run_case(sb_handle, pair, 63, 128, 128, ta, tb, 3.0, 7.0, offset=off, batch=5)


@pytest.mark.parametrize("pair", PAIRS, ids=lambda p: f"{p[0]}-{p[1]}")
@pytest.mark.parametrize("family", ["LARGE", "SMALL"])
def test_both_tile_families(sb_handle, pair, family):
    fam = getattr(_lib, "KERNEL_" + family)
    for (m, n, k), ta, tb in itertools.product([(1, 1, 1), (129, 127, 70), (256, 128, 192), (200, 260, 133)], "nt", "nt"):
        run_case(sb_handle, pair, m, n, k, ta, tb, 1.5, 0.5, family=fam)
        run_case(sb_handle, pair, m, n, k, ta, tb, -2.0, 0.0, ldm=2, offset=8, family=fam)
        # odd leading dimensions behind an odd offset: every 16-byte vector of the operands is misaligned, and
        # the vectors that end past the matrix edge are gathered element by element
        run_case(sb_handle, pair, m, n, k, ta, tb, 1.0, 1.0, offset=3, family=fam)


@pytest.mark.parametrize("pair", PAIRS, ids=lambda p: f"{p[0]}-{p[1]}")
def test_direct_to_lds_family(sb_handle, pair):
    # The 256x256 ring kernel (K steps of 32, four tiles in flight): whole tiles, tiles hanging
    # over both edges, fewer K steps than the ring is deep and many more, padded leading
    # dimensions and a 16-byte-aligned offset, every transpose combination.
    fam = _lib.KERNEL_HUGE
    # K tails (k % 32 != 0: one zero-padded tile behind the ring): shorter and longer than half a K step, behind
    # one ring step and behind many.
    for (m, n, k), ta, tb in itertools.product([(256, 256, 32), (304, 520, 96), (512, 264, 160), (1000, 776, 256), (264, 256, 1024),
                                                (512, 256, 2048), (256, 512, 2104),  # four wavefronts of 128 x 128
                                                (256, 256, 33), (304, 520, 100), (512, 264, 1007), (264, 520, 49), (256, 264, 40), (520, 256, 1016)],
                                               "nt", "nt"):
        # a K tail makes the leading dimension of a K-contiguous operand odd: those fall back (16-byte rows needed)
        def takes(ldm):
            lda, ldb = (k if ta != "n" else m) * ldm, (n if tb != "n" else k) * ldm
            return lda % 8 == 0 and ldb % 8 == 0

        run_case(sb_handle, pair, m, n, k, ta, tb, 1.5, 0.5, family=fam)
        assert (sb_handle.last_kernel() == fam) == takes(1)
        run_case(sb_handle, pair, m, n, k, ta, tb, -2.0, 0.0, ldm=2, offset=8, family=fam)
        assert (sb_handle.last_kernel() == fam) == takes(2)
    # shapes it cannot take (K shorter than a K step, odd M with M contiguous, unaligned offset) fall back to the other tiles
    run_case(sb_handle, pair, 256, 256, 31, "n", "n", 1.0, 1.0, family=fam)
    assert sb_handle.last_kernel() != fam
    run_case(sb_handle, pair, 257, 256, 128, "n", "n", 1.0, 1.0, family=fam)
    assert sb_handle.last_kernel() != fam
    run_case(sb_handle, pair, 256, 256, 128, "t", "n", 1.0, 1.0, offset=3, family=fam)
    assert sb_handle.last_kernel() != fam
    # ... but an odd M is fine when K is A's contiguous index
    run_case(sb_handle, pair, 257, 256, 128, "t", "n", 1.0, 1.0, family=fam)
    assert sb_handle.last_kernel() == fam


@pytest.mark.parametrize("pair", PAIRS, ids=lambda p: f"{p[0]}-{p[1]}")
def test_k_cut_for_thin_shapes(sb_handle, pair):
    # Few tiles and a long K: the K range is cut over extra workgroups, partial sums in fp32.
    for (m, n, k), ta, tb in itertools.product([(48, 130, 2048), (2, 64, 4100), (100, 7, 1537), (300, 500, 2048)], "nt", "nt"):
        run_case(sb_handle, pair, m, n, k, ta, tb, 1.5, 0.5)
        assert sb_handle.last_kernel() == _lib.KERNEL_SPLITK and sb_handle.last_splitk()[0] > 1
        run_case(sb_handle, pair, m, n, k, ta, tb, -2.0, 0.0, ldm=2, offset=8)
    # a pinned depth on a shape that would not be cut by itself
    h = sb_handle
    h.set_kernel(_lib.KERNEL_SPLITK, 3)
    try:
        run_case(h, pair, 200, 150, 333, "n", "t", 1.0, 1.0, family=_lib.KERNEL_SPLITK, param=3)
    finally:
        h.set_kernel(_lib.KERNEL_AUTO)


def test_alpha_zero_and_beta_zero_on_half_output(sb_handle):
    h = sb_handle
    m = n = 16
    c = np.arange(m * n, dtype=np.float16)
    c[7] = np.nan
    a_g = helper.allocate(h, "float16", m * n)
    c_g = helper.allocate(h, "float16", m * n)
    helper.copy_to_device(h, np.ones(m * n, np.float16).view(np.uint16), a_g)
    out = np.empty(m * n, np.uint16)
    # alpha == 0 multiplies C by beta in place; the trans flags are not looked at
    helper.copy_to_device(h, c.view(np.uint16), c_g)
    sb._gemm(h, "x", "y", m, n, 4, 0.0, a_g, m, a_g, m, 2.0, c_g, m)
    h.wait(helper.copy_to_host(h, c_g, out))
    want = (c.astype(np.float32) * 2).astype(np.float16)
    assert np.array_equal(out.view(np.float16), want, equal_nan=True)
    # beta == 0 never reads C: the NaN does not survive
    helper.copy_to_device(h, c.view(np.uint16), c_g)
    sb._gemm(h, "n", "n", m, n, 16, 1.0, a_g, m, a_g, m, 0.0, c_g, m)
    h.wait(helper.copy_to_host(h, c_g, out))
    assert (out.view(np.float16) == 16).all()
    with pytest.raises(ValueError, match="invalid _TransB"):
        sb._gemm(h, "n", "q", m, n, 16, 1.0, a_g, m, a_g, m, 0.0, c_g, m)


@pytest.mark.parametrize("pair", PAIRS, ids=lambda p: f"{p[0]}-{p[1]}")
def test_every_kernel_gives_the_same_bits(sb_handle, pair):
    """One problem through every 16-bit kernel that can take it: the register-staged tiles, both tiles of the
    direct-to-LDS kernel, and -- for one K cut -- slabs on either kernel.  All of them feed the same matrix
    instruction the same 16 k at a time in the same order, so the outputs have to be identical, K tail included."""
    h = sb_handle
    storage, out_name = pair
    # (the last shape -- whole 256 x 256 tiles, K of 2048 and a tail -- takes the four-wavefront grid of the big tile
    # unless bit 1 of its parameter asks for eight)
    for (m, n, k), ta, tb in itertools.product([(264, 392, 616), (136, 520, 1000), (256, 512, 2072)], "nt", "nt"):
        rng = seeded("same-bits", pair, m, n, k, ta, tb)
        lda, ldb = (k if ta != "n" else m), (n if tb != "n" else k)
        a_bits, _ = make_inputs(rng, m * k, storage)
        b_bits, _ = make_inputs(rng, k * n, storage)
        c0 = rng.uniform(-2.0, 5.0, m * n).astype(np.float16 if out_name == "float16" else np.float32)
        c_dev = c0.view(np.uint16).copy() if out_name == "float16" else c0
        a_g, b_g, c_g = (helper.allocate(h, storage, a_bits.size), helper.allocate(h, storage, b_bits.size),
                         helper.allocate(h, out_name, c_dev.size))
        helper.copy_to_device(h, a_bits, a_g)
        helper.copy_to_device(h, b_bits, b_g)

        def run(family, param):
            h.set_kernel(family, param)
            helper.copy_to_device(h, c_dev, c_g)
            sb._gemm(h, ta, tb, m, n, k, 1.5, a_g, lda, b_g, ldb, 0.5, c_g, m)
            out = np.empty_like(c_dev)
            h.wait(helper.copy_to_host(h, c_g, out))
            return out, h.last_kernel()

        try:
            base, ran = run(_lib.KERNEL_SMALL, 0)
            assert ran == _lib.KERNEL_SMALL
            for family, param in [(_lib.KERNEL_LARGE, 0), (_lib.KERNEL_LARGE, 1), (_lib.KERNEL_HUGE, 0), (_lib.KERNEL_HUGE, 2)]:
                got, ran = run(family, param)
                assert ran == family and np.array_equal(got, base), (pair, m, n, k, ta, tb, family, param)
            ring, ran = run(_lib.KERNEL_SPLITK, 3)
            assert ran == _lib.KERNEL_SPLITK and h.last_splitk()[0] == 3
            staged, _ = run(_lib.KERNEL_SPLITK, 3 | (1 << 16))
            assert h.last_splitk()[0] == 3 and np.array_equal(ring, staged), (pair, m, n, k, ta, tb, "K cut")
        finally:
            h.set_kernel(_lib.KERNEL_AUTO)
            for buf in (a_g, b_g, c_g):
                helper.deallocate(buf)


@pytest.mark.parametrize("pair", PAIRS, ids=lambda p: f"{p[0]}-{p[1]}")
def test_unaligned_operands_of_big_products(sb_handle, pair):
    """Big products whose operands the direct-to-LDS kernel cannot take as they lie (odd leading dimensions, an odd
    offset, M / N not a multiple of 8): the library copies them into padded, aligned buffers first (here forced, at
    sizes the CPU check finishes quickly: bit 0 of the automatic choice's parameter).  Checked like every other case."""
    h = sb_handle
    for (m, n, k), ta, tb in [((1100, 1300, 1028), "n", "n"), ((1036, 1540, 999), "t", "n"), ((1101, 1203, 1024), "n", "t"),
                              ((1500, 1100, 1032), "t", "t"), ((301, 203, 2048), "n", "t"), ((7, 5, 64), "n", "t")]:
        run_case(h, pair, m, n, k, ta, tb, 1.5, 0.5, offset=3, family=_lib.KERNEL_AUTO, param=1)
        if m >= 1000:  # enough 64 x 64 tiles for the 128 x 128 direct-to-LDS tile to be the cheaper choice
            assert h.last_kernel() in (_lib.KERNEL_HUGE, _lib.KERNEL_LARGE)
        run_case(h, pair, m, n, k, ta, tb, -1.0, 0.0, ldm=2, offset=0, family=_lib.KERNEL_AUTO, param=1)


def test_hgemm_against_the_reference_half_arithmetic(sb_handle):
    """half -> half: the reference accumulates in binary16 (sycl::mad(half, half, half),
    src/operations/blas3/gemm_common.hpp:45-51) where this library accumulates in float and rounds once -- a
    deliberate difference (better numerics, different bits).  Both must satisfy what the reference's own test asks
    of that pair: the HALF predicate (common/include/common/float_comparison.hpp:123-158).  Checked on the
    reference's grids (blas3_gemm_test.cpp:29-84, :124-141 a slice) against the reference's binary16 chain in both of
    its forms AND against the long-double result."""
    h = sb_handle
    cases = [(m, n, k) for m, n, k in itertools.product([11, 16, 32], [11, 32], [16, 17])] + [(253, 257, 511), (1024, 257, 253)]
    for (m, n, k), ta, tb, (alpha, beta) in itertools.product(cases, "nt", "nt", [(1.5, 1.5), (1.5, 0.0)]):
        rng = seeded("half-arith", m, n, k, ta, tb, alpha, beta)
        lda, ldb = (k if ta == "t" else m), (n if tb == "t" else k)
        a_h = rng.uniform(-2.0, 5.0, m * k).astype(np.float16)
        b_h = rng.uniform(-2.0, 5.0, k * n).astype(np.float16)
        c_h = rng.uniform(-2.0, 5.0, m * n).astype(np.float16)
        a_g, b_g, c_g = (helper.allocate(h, "float16", x.size) for x in (a_h, b_h, c_h))
        deps = [helper.copy_to_device(h, x.view(np.uint16), g) for x, g in ((a_h, a_g), (b_h, b_g), (c_h, c_g))]
        h.wait(sb._gemm(h, ta, tb, m, n, k, alpha, a_g, lda, b_g, ldb, beta, c_g, m, deps))
        got = np.empty(m * n, np.uint16)
        h.wait(helper.copy_to_host(h, c_g, got))
        for buf in (a_g, b_g, c_g):
            helper.deallocate(buf)
        got = got.view(np.float16).astype(np.float64)
        what = f"half->half {m}x{n}x{k} {ta}{tb} a={alpha} b={beta}"
        truth, _ = oracle.gemm_truth(ta == "t", tb == "t", m, n, k, alpha, a_h.astype(np.float64), lda,
                                     b_h.astype(np.float64), ldb, beta, c_h.astype(np.float64), m)
        assert oracle.compare_vectors(got, truth.astype(np.float16).astype(np.float64), "float16") == -1, what + " vs truth"
        for form in (oracle.FORM_LOCAL, oracle.FORM_REF):
            chain = c_h.view(np.uint16).copy()
            assert oracle.gemm_half_chain(ta, tb, m, n, k, alpha, a_h.view(np.uint16), lda, b_h.view(np.uint16), ldb, beta,
                                          chain, m, form=form) == 0
            assert oracle.compare_vectors(got, oracle.half_to_double(chain), "float16") == -1, what + f" vs binary16 chain (form {form})"
