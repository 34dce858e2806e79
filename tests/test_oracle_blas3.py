"""CPU checks of the oracle's _symm / _trsm restatements against the system BLAS.

The reference validates these routines against a system BLAS with
utils::compare_vectors (test/unittest/blas3/blas3_symm_test.cpp:73-99,
blas3_trsm_test.cpp:73-102); scipy's OpenBLAS plays that role here, on the
reference's parameter grids.
"""
import numpy as np
import pytest

import oracle
from blas3_cases import scipy_symm, scipy_trsm, symm_cases, symm_inputs, trsm_cases, trsm_inputs
from common import assert_reference_predicate


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_symm_matches_system_blas(dtype):
    count = 0
    for case in symm_cases():
        name, m, n, side, uplo, alpha, beta = case[:7]
        k, lda, ldb, ldc, a, b, c = symm_inputs(dtype, *case)
        want = scipy_symm(side, uplo, m, n, alpha, a, lda, b, ldb, beta, c, ldc)
        got = c.copy()
        assert oracle.symm(side, uplo, m, n, alpha, a, lda, b, ldb, beta, got, ldc) == 0
        assert_reference_predicate(got, want, f"symm {case}")
        count += 1
    assert count == 36 + 64 + 128 + 16


def test_symm_reads_only_the_stored_triangle():
    m, n = 37, 29
    for side in "lr":
        for uplo in "lu":
            k, lda, ldb, ldc, a, b, c = symm_inputs(np.float64, "tri", m, n, side, uplo, 1.25, -0.5, 2, 1, 1)
            poisoned = a.copy()
            A = np.lib.stride_tricks.as_strided(poisoned, shape=(k, k), strides=(8, lda * 8))
            idx = np.triu_indices(k, 1) if uplo == "l" else np.tril_indices(k, -1)
            A[idx] = np.nan
            c0, c1 = c.copy(), c.copy()
            assert oracle.symm(side, uplo, m, n, 1.25, a, lda, b, ldb, -0.5, c0, ldc) == 0
            assert oracle.symm(side, uplo, m, n, 1.25, poisoned, lda, b, ldb, -0.5, c1, ldc) == 0
            assert np.array_equal(c0, c1) and np.isfinite(c1).all()


def test_symm_argument_errors():
    z = np.zeros(16, dtype=np.float32)
    assert oracle.symm("l", "x", 4, 4, 1.0, z, 4, z, 4, 0.0, z.copy(), 4) == 11  # "invalid _uplo"
    assert oracle.symm("x", "u", 4, 4, 1.0, z, 4, z, 4, 0.0, z.copy(), 4) == 12  # "invalid _side"
    assert oracle.symm("x", "x", 4, 4, 1.0, z, 4, z, 4, 0.0, z.copy(), 4) == 11  # uplo is checked first


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_trsm_matches_system_blas(dtype):
    for case in trsm_cases(sizes=(7, 513)):
        m, n, trans, side, diag, uplo, alpha, la, lb, unused = case
        k, lda, ldb, a, b = trsm_inputs(dtype, *case)
        want = scipy_trsm(side, uplo, trans, diag, m, n, alpha, a, lda, b, ldb)
        got = b.copy()
        assert oracle.trsm(side, uplo, trans, diag, m, n, alpha, a, lda, got, ldb) == 0
        assert not np.isnan(got[: m]).any(), f"trsm {case} read the unused triangle"
        assert_reference_predicate(got, want, f"trsm {case}")


def test_trsm_largest_reference_size():
    for case in [(1027, 1027, "n", "l", "n", "l", 2.0, 2, 2, float("nan")),
                 (1027, 513, "t", "r", "u", "u", 2.0, 2, 2, 0.0)]:
        m, n, trans, side, diag, uplo, alpha, la, lb, unused = case
        k, lda, ldb, a, b = trsm_inputs(np.float64, *case)
        want = scipy_trsm(side, uplo, trans, diag, m, n, alpha, a, lda, b, ldb)
        got = b.copy()
        assert oracle.trsm(side, uplo, trans, diag, m, n, alpha, a, lda, got, ldb) == 0
        assert_reference_predicate(got, want, f"trsm {case}")


def test_trsm_solves_the_system():
    """Property check independent of any BLAS: op(A) * X == alpha * B."""
    m, n = 45, 23
    for side, uplo, trans, diag in [("l", "l", "n", "n"), ("l", "u", "t", "u"), ("r", "u", "n", "n"),
                                    ("r", "l", "t", "u")]:
        k, lda, ldb, a, b = trsm_inputs(np.float64, m, n, trans, side, diag, uplo, 2.0, 2, 2, 0.0)
        x = b.copy()
        assert oracle.trsm(side, uplo, trans, diag, m, n, 2.0, a, lda, x, ldb) == 0
        A = np.lib.stride_tricks.as_strided(a, shape=(k, k), strides=(8, lda * 8)).copy()
        if diag == "u":
            np.fill_diagonal(A, 1.0)
        opA = A.T if trans == "t" else A
        X = np.lib.stride_tricks.as_strided(x, shape=(m, n), strides=(8, ldb * 8))
        B = np.lib.stride_tricks.as_strided(b, shape=(m, n), strides=(8, ldb * 8))
        lhs = opA @ X if side == "l" else X @ opA
        assert np.allclose(lhs, 2.0 * B, rtol=1e-9, atol=1e-9)
        # rows of the ldb-strided buffer outside the m x n window are untouched
        mask = np.ones(x.size, dtype=bool)
        for j in range(n):
            mask[j * ldb: j * ldb + m] = False
        assert np.array_equal(x[mask], b[mask])


def test_trsm_argument_errors():
    z = np.ones(16, dtype=np.float64)
    assert oracle.trsm("l", "l", "n", "n", 0, 4, 1.0, z, 4, z.copy(), 4) == 13
    assert oracle.trsm("l", "l", "n", "n", 4, 4, 1.0, z, 0, z.copy(), 4) == 13
    assert oracle.trsm("x", "l", "n", "n", 4, 4, 1.0, z, 4, z.copy(), 4) == 14
    assert oracle.trsm("l", "x", "n", "n", 4, 4, 1.0, z, 4, z.copy(), 4) == 15
    assert oracle.trsm("l", "l", "c", "n", 4, 4, 1.0, z, 4, z.copy(), 4) == 16
    assert oracle.trsm("l", "l", "n", "x", 4, 4, 1.0, z, 4, z.copy(), 4) == 17
