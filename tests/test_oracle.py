"""CPU tests of the oracle itself (no GPU).

The reference has no golden vectors for GEMM: its tests compare with a system
CBLAS on random inputs under the predicate of
common/include/common/float_comparison.hpp:163-217.  The oracle is pinned the
same way -- against OpenBLAS (scipy) on the reference's parameter grids -- and
against the committed fixtures in tests/golden/.
"""
import itertools
import os

import numpy as np
import pytest

import oracle

from common import fill_random, openblas_gemm, seeded

DTYPES = [np.float32, np.float64]
HERE = os.path.dirname(os.path.abspath(__file__))


def one_case(dtype, offset, batch, m, n, k, ta, tb, alpha, beta, lda_mul, ldb_mul, ldc_mul, form=oracle.FORM_REF,
             depth=1, k_chunk=0):
    rng = seeded("oracle", np.dtype(dtype).name, offset, batch, m, n, k, ta, tb, alpha, beta)
    lda = (k if ta != "n" else m) * lda_mul
    ldb = (n if tb != "n" else k) * ldb_mul
    ldc = m * ldc_mul
    size_a, size_b, size_c = m * k * lda_mul, k * n * ldb_mul, m * n * ldc_mul
    a = fill_random(rng, batch * size_a + offset, dtype)
    b = fill_random(rng, batch * size_b + offset, dtype)
    c = fill_random(rng, batch * size_c + offset, dtype)
    got = c.copy()
    st = oracle.gemm_backend(ta, tb, m, n, k, alpha, a, lda, size_a, b, ldb, size_b, beta, got, ldc, size_c,
                             batch=batch, form=form, depth=depth, k_chunk=k_chunk, a_off=offset, b_off=offset,
                             c_off=offset)
    assert st == 0
    ref = c.copy()
    for i in range(batch):
        openblas_gemm(ta, tb, m, n, k, alpha, a, i * size_a + offset, lda, b, i * size_b + offset, ldb, beta, ref,
                      i * size_c + offset, ldc)
    assert oracle.compare_vectors(got, ref) == -1
    return a, b, c, got


@pytest.mark.parametrize("dtype", DTYPES, ids=lambda d: np.dtype(d).name)
def test_against_openblas_small_grids(dtype):
    # blas3_gemm_test.cpp:29-122
    for m, n, k, ta, tb in itertools.product([11, 16, 32], [11, 16, 32], [16, 17], "nt", "nt"):
        one_case(dtype, 0, 1, m, n, k, ta, tb, 1.5, 1.5, 1, 1, 1)
    for m, n, ta, tb in itertools.product([11, 32], [11, 32], "nt", "nt"):
        one_case(dtype, 0, 1, m, n, 17, ta, tb, 1.5, 0.0, 2, 3, 4)
    for off, m, n, k, la, lb, lc in itertools.product([1, 10], [16, 63], [16, 63], [17, 63], [1, 2], [1, 2], [1, 2]):
        one_case(dtype, off, 1, m, n, k, "n", "n", 1.0, 1.0, la, lb, lc)


@pytest.mark.parametrize("dtype", DTYPES, ids=lambda d: np.dtype(d).name)
def test_against_openblas_large_and_batched(dtype):
    # blas3_gemm_test.cpp:124-141 (a slice) and blas3_gemm_batched_test.cpp:29-46
    for m, n, k, ta, tb in itertools.product([253, 1024], [257], [253, 511], "nt", "nt"):
        one_case(dtype, 0, 1, m, n, k, ta, tb, 1.0, 1.0, 1, 1, 1)
    for off, m, ta, tb in itertools.product([0, 33], [63, 128], "nt", "nt"):
        one_case(dtype, off, 5, m, 63, 128, ta, tb, 3.0, 7.0, 1, 1, 1)


@pytest.mark.parametrize("dtype", DTYPES, ids=lambda d: np.dtype(d).name)
def test_all_arithmetic_forms_agree(dtype):
    # naive form, (beta/alpha) form and both split-K forms differ only by rounding
    # (blas3_gemm_tall_skinny_test.cpp:29-65 shapes)
    for m, n, ta, tb in itertools.product([7, 65], [9, 126], "nt", "nt"):
        results = []
        for form in (oracle.FORM_REF, oracle.FORM_LOCAL, oracle.FORM_SPLITK, oracle.FORM_SPLITK_TREE):
            *_, got = one_case(dtype, 0, 1, m, n, 2049, ta, tb, 1.5, 0.5, 2, 3, 4, form=form, depth=9, k_chunk=240)
            results.append(got)
        for other in results[1:]:
            assert oracle.compare_vectors(other, results[0]) == -1


def test_split_of_one_slab_equals_reference_form():
    rng = np.random.default_rng(3)
    m, n, k = 19, 23, 77
    a, b, c = (fill_random(rng, x, np.float32) for x in (m * k, k * n, m * n))
    r0, r1 = c.copy(), c.copy()
    oracle.gemm_backend("n", "t", m, n, k, 1.5, a, m, 0, b, n, 0, 0.0, r0, m, 0, form=oracle.FORM_REF)
    oracle.gemm_backend("n", "t", m, n, k, 1.5, a, m, 0, b, n, 0, 0.0, r1, m, 0, form=oracle.FORM_SPLITK, depth=1,
                        k_chunk=k)
    assert np.array_equal(r0, r1)


def test_chain_is_sequential_fma():
    # One element, by hand: the accumulator is fma(a_k, b_k, acc) for k = 0, 1, ...
    # (gemm_ref.hpp:239-244), then alpha*acc + beta*c with separate roundings.
    rng = np.random.default_rng(11)
    k = 300
    a = fill_random(rng, k, np.float32)
    b = fill_random(rng, k, np.float32)
    c = np.array([0.75], dtype=np.float32)
    acc = np.float32(0)
    for i in range(k):
        acc = np.float32(np.float64(a[i]) * np.float64(b[i]) + np.float64(acc))  # exact product, one rounding
    want = np.float32(np.float32(np.float32(1.5) * acc) + np.float32(np.float32(0.5) * c[0]))
    got = c.copy()
    oracle.gemm_backend("t", "n", 1, 1, k, 1.5, a, k, 0, b, k, 0, 0.5, got, 1, 0)
    assert got[0] == want


@pytest.mark.parametrize("dtype", DTYPES, ids=lambda d: np.dtype(d).name)
def test_interleaved_equals_strided(dtype):
    from common import from_interleaved, to_interleaved
    for ta, tb in itertools.product("nt", "nt"):
        m, n, k, batch, off = 13, 9, 21, 5, 3
        lda, ldb, ldc = (k if ta != "n" else m) * 2, (n if tb != "n" else k) * 3, m * 4
        sa, sb, sc = m * k * 2, k * n * 3, m * n * 4
        rng = seeded("inter", np.dtype(dtype).name, ta, tb)
        a, b, c = (fill_random(rng, batch * s + off, dtype) for s in (sa, sb, sc))
        want = c.copy()
        assert oracle.gemm_backend(ta, tb, m, n, k, 3.0, a, lda, sa, b, ldb, sb, 7.0, want, ldc, sc, batch=batch,
                                   a_off=off, b_off=off, c_off=off) == 0
        ai = to_interleaved(a, off, lda, m if ta == "t" else k, batch)
        bi = to_interleaved(b, off, ldb, k if tb == "t" else n, batch)
        ci = to_interleaved(c, off, ldc, n, batch)
        assert oracle.gemm_backend(ta, tb, m, n, k, 3.0, ai, lda, 0, bi, ldb, 0, 7.0, ci, ldc, 0, batch=batch,
                                   batch_type=1, a_off=off, b_off=off, c_off=off) == 0
        assert np.array_equal(from_interleaved(ci, off, ldc, n, batch), want)


def test_front_end_rules():
    # src/interface/gemm_interface.hpp:112-166
    f = np.float32
    a = np.ones(64, f)
    b = np.ones(64, f)
    c = np.arange(64, dtype=f)
    assert oracle.gemm_backend("x", "n", 4, 4, 4, 1.0, a, 4, 0, b, 4, 0, 0.0, c.copy(), 4, 0) == 1
    assert oracle.gemm_backend("n", "x", 4, 4, 4, 1.0, a, 4, 0, b, 4, 0, 0.0, c.copy(), 4, 0) == 2
    assert oracle.gemm_backend("n", "n", 4, 4, 4, 1.0, a, 4, 16, b, 4, 16, 0.0, c.copy(), 4, 15, batch=2) == 3
    assert oracle.gemm_backend("n", "n", 4, 4, 4, 1.0, a, 4, -1, b, 4, 16, 0.0, c.copy(), 4, 16, batch=2) == 4
    assert oracle.gemm_backend("n", "n", 4, 4, 4, 1.0, a, 4, 16, b, 4, -1, 0.0, c.copy(), 4, 16, batch=2) == 5
    # alpha == 0 is decided before the trans flags are looked at
    out = c.copy()
    assert oracle.gemm_backend("x", "y", 4, 4, 4, 0.0, a, 4, 0, b, 4, 0, 2.0, out, 4, 0) == 0
    assert np.array_equal(out[:16], 2 * c[:16]) and np.array_equal(out[16:], c[16:])
    # 'c' behaves as 't' (:150-151), upper case accepted (:141-142)
    r1, r2 = c.copy(), c.copy()
    a2 = np.arange(64, dtype=f)
    oracle.gemm_backend("C", "N", 4, 4, 4, 1.0, a2, 4, 0, b, 4, 0, 0.0, r1, 4, 0)
    oracle.gemm_backend("t", "n", 4, 4, 4, 1.0, a2, 4, 0, b, 4, 0, 0.0, r2, 4, 0)
    assert np.array_equal(r1, r2)


def test_alpha_zero_scal_versus_scal_matrix():
    # whole-vector _scal with beta == 0 stores zeros (blas1_interface.hpp:444-448);
    # _scal_matrix multiplies, so NaN survives (:498-506) and padding rows are untouched
    f = np.float32
    a = np.ones(64, f)
    c = np.arange(64, dtype=f)
    c[3] = np.nan
    out = c.copy()
    oracle.gemm_backend("n", "n", 4, 4, 4, 0.0, a, 4, 16, a, 4, 16, 0.0, out, 4, 16, batch=2)  # stride == ldc*n, ldc == m
    assert (out[:32] == 0).all() and np.array_equal(out[32:], c[32:])
    out = c.copy()
    oracle.gemm_backend("n", "n", 4, 4, 4, 0.0, a, 4, 0, a, 4, 0, 0.0, out, 4, 0)  # plain _gemm: stride_c == 0
    assert np.isnan(out[3]) and (np.delete(out[:16], 3) == 0).all()
    out = c.copy()
    oracle.gemm_backend("n", "n", 3, 4, 4, 0.0, a, 4, 0, a, 4, 0, 5.0, out, 4, 0)  # ldc > m: row 3 of each column stays
    assert np.array_equal(out[7:16:4], c[7:16:4]) and out[4] == 5 * c[4]


def test_golden_fixtures():
    """Committed seeded cases (tests/golden/make_golden.py) must reproduce bit for bit."""
    path = os.path.join(HERE, "golden", "gemm_golden.npz")
    data = np.load(path, allow_pickle=False)
    names = sorted({key.rsplit(".", 1)[0] for key in data.files})
    assert names, "no fixtures found"
    for name in names:
        meta = data[name + ".meta"]
        form, ta, tb, m, n, k, lda, ldb, ldc, batch, sa, sb, sc, depth, k_chunk = (int(x) for x in meta)
        alpha, beta = data[name + ".scal"]
        a, b, c, want = data[name + ".a"], data[name + ".b"], data[name + ".c"], data[name + ".out"]
        got = c.copy()
        st = oracle.gemm_backend("nt"[ta], "nt"[tb], m, n, k, float(alpha), a, lda, sa, b, ldb, sb, float(beta), got,
                                 ldc, sc, batch=batch, form=form, depth=depth, k_chunk=k_chunk)
        assert st == 0
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), name


def test_default_backend_tile_kernel_equals_the_local_form():
    """The restatement of the kernel the reference's CPU back end runs for config #1 (no_local partial,
    Tile<4,4,4,4>: src/interface/blas3/backend/default.hpp:98-112,
    src/operations/blas3/gemm_no_local_partial_vec.hpp:189-416) computes, element for element, the
    alpha*((beta/alpha)*C + sum) form of every tiled kernel: whole buffers, ragged edges in both
    dimensions, padded leading dimensions, every transpose, beta zero and non-zero."""
    rng = np.random.default_rng(2024)
    for dtype in (np.float32, np.float64):
        for (m, n, k), ta, tb in itertools.product([(37, 29, 19), (64, 48, 33), (17, 100, 5), (130, 131, 40), (1, 1, 1)],
                                                   "nt", "nt"):
            lda = (k if ta != "n" else m) + 3
            ldb = (n if tb != "n" else k) + 2
            ldc = m + 1
            a = rng.uniform(-2, 5, lda * (m if ta != "n" else k)).astype(dtype)
            b = rng.uniform(-2, 5, ldb * (k if tb != "n" else n)).astype(dtype)
            c0 = rng.uniform(-2, 5, ldc * n).astype(dtype)
            for alpha, beta in ((1.5, 0.5), (2.0, 0.0)):
                got, want = c0.copy(), c0.copy()
                assert oracle.gemm_default_tile(ta, tb, m, n, k, alpha, a, lda, b, ldb, beta, got, ldc) == 0
                assert oracle.gemm_backend(ta, tb, m, n, k, alpha, a, lda, 0, b, ldb, 0, beta, want, ldc, 0,
                                           form=oracle.FORM_LOCAL) == 0
                assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (dtype, m, n, k, ta, tb, alpha, beta)
    # beta == 0 never reads C (NaN in C must not survive)
    c = np.full(16 * 16, np.nan, np.float32)
    ones = np.ones(16 * 16, np.float32)
    oracle.gemm_default_tile("n", "n", 16, 16, 16, 1.0, ones, 16, ones, 16, 0.0, c, 16)
    assert (c == 16).all()


# ---- half -> half: the reference accumulates in binary16 (src/operations/blas3/gemm_common.hpp:45-51)


def test_half_conversion_is_numpy_float16_rounding():
    # numpy's float64 -> float16 conversion is correctly rounded; the oracle's own conversion must agree on
    # normals, subnormals, ties (and near-ties), overflow and the special values
    rng = np.random.default_rng(11)
    h = rng.integers(0, 0x7BFF, 50000).astype(np.uint16)
    lo = h.view(np.float16).astype(np.float64)
    hi = (h + 1).astype(np.uint16).view(np.float16).astype(np.float64)
    mid = (lo + hi) / 2
    x = np.concatenate([rng.uniform(-70000, 70000, 100000), rng.uniform(-1e-4, 1e-4, 100000), rng.uniform(-1e-7, 1e-7, 50000),
                        mid, mid * (1 + 1e-15), mid * (1 - 1e-15), -mid,
                        [0.0, -0.0, np.inf, -np.inf, 65504, 65519.99, 65520, 65536, 2.0 ** -24, 2.0 ** -25, 3 * 2.0 ** -25, 1e-320]])
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).view(np.uint16)
    assert np.array_equal(oracle.half_from_double(x), want)
    every = np.arange(65536, dtype=np.uint16)
    assert np.array_equal(oracle.half_to_double(every), every.view(np.float16).astype(np.float64), equal_nan=True)
    assert np.isnan(oracle.half_to_double(oracle.half_from_double(np.array([np.nan])))).all()


def _half_inputs(rng, count):
    h = rng.uniform(-2.0, 5.0, count).astype(np.float16)
    return h.view(np.uint16).copy(), h.astype(np.float64)


def test_half_chain_equals_a_step_by_step_float16_loop():
    # the same chain written with numpy float16 scalars: every product-and-add formed in float64 (exact product,
    # sum rounded once) and rounded to float16 by numpy
    rng = np.random.default_rng(5)
    m, n, k = 5, 4, 37
    for form, ta, tb in itertools.product([oracle.FORM_REF, oracle.FORM_LOCAL], "nt", "nt"):
        lda, ldb, ldc = (k if ta == "t" else m) + 1, (n if tb == "t" else k) + 2, m + 3
        a_bits, a = _half_inputs(rng, lda * (m if ta == "t" else k))
        b_bits, b = _half_inputs(rng, ldb * (k if tb == "t" else n))
        c_bits, c = _half_inputs(rng, ldc * n)
        alpha, beta = np.float16(1.5), np.float16(-0.75)
        got = c_bits.copy()
        assert oracle.gemm_half_chain(ta, tb, m, n, k, alpha, a_bits, lda, b_bits, ldb, beta, got, ldc, form=form) == 0
        want = c_bits.view(np.float16).copy()
        f16, f64 = np.float16, np.float64
        for j, i in itertools.product(range(n), range(m)):
            acc = f16(0)
            if form == oracle.FORM_LOCAL:
                acc = f16(f64(f16(f64(beta) / f64(alpha))) * c[i + j * ldc])
            for l in range(k):
                av = a[l + i * lda] if ta == "t" else a[i + l * lda]
                bv = b[j + l * ldb] if tb == "t" else b[l + j * ldb]
                acc = f16(av * bv + f64(acc))
            out = f16(f64(alpha) * f64(acc))
            if form == oracle.FORM_REF:
                out = f16(f64(out) + f64(f16(f64(beta) * c[i + j * ldc])))
            want[i + j * ldc] = out
        assert np.array_equal(got, want.view(np.uint16)), (form, ta, tb)


def test_half_chain_passes_the_reference_half_predicate():
    # What the reference's own half -> half test checks (test/unittest/blas3/blas3_gemm_common.hpp:147-152, :217-225
    # with the half margins of float_comparison.hpp:123-158): its binary16 chain against a float GEMM of the same
    # values.  Reference grids blas3_gemm_test.cpp:29-84 and :124-141 (a slice).
    for (m, n, k), ta, tb in itertools.product([(11, 16, 17), (32, 11, 16), (253, 257, 511), (64, 33, 253)], "nt", "nt"):
        rng = seeded("half-chain", m, n, k, ta, tb)
        lda, ldb = (k if ta == "t" else m), (n if tb == "t" else k)
        a_bits, a = _half_inputs(rng, m * k)
        b_bits, b = _half_inputs(rng, k * n)
        c_bits, c = _half_inputs(rng, m * n)
        truth, _ = oracle.gemm_truth(ta == "t", tb == "t", m, n, k, 1.5, a, lda, b, ldb, 1.5, c, m)
        for form, fused in itertools.product([oracle.FORM_REF, oracle.FORM_LOCAL], [True, False]):
            got = c_bits.copy()
            assert oracle.gemm_half_chain(ta, tb, m, n, k, 1.5, a_bits, lda, b_bits, ldb, 1.5, got, m, form=form, fused=fused) == 0
            assert oracle.compare_vectors(oracle.half_to_double(got), truth.astype(np.float16).astype(np.float64), "float16") == -1
