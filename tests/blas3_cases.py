"""Inputs and parameter grids of the reference's _symm / _trsm tests, seeded.

Grids: test/unittest/blas3/blas3_symm_test.cpp:153-211 and
test/unittest/blas3/blas3_trsm_test.cpp:155-168.  The triangular matrix follows
fill_trsm_matrix (test/blas_test.hpp:185-214): a diagonally dominant triangle
whose unused half holds a marker (0 or NaN) that the solver must never read.
"""
import itertools

import numpy as np

from common import fill_random, seeded

SYMM_GRIDS = {
    # name: (m values, n values, alpha, betas, ld multipliers)
    "SmallBetaNonZeroLDMatch": ((11, 16, 32), (11, 16, 32), 1.5, (0.5,), (1,)),
    "AlphaZero": ((16,), (16,), 0.0, (0.0, 1.0), (1, 2)),
    "OffsetNonZero": ((16, 63), (16, 63), 1.0, (1.0,), (1, 2)),
    "LargeBetaNonZeroLDMatch": ((253, 511), (257, 511), 1.0, (1.0,), (1,)),
}


def symm_cases():
    for name, (ms, ns, alpha, betas, muls) in SYMM_GRIDS.items():
        for m, n, side, uplo, beta, la, lb, lc in itertools.product(ms, ns, "lr", "lu", betas, muls, muls, muls):
            yield (name, m, n, side, uplo, alpha, beta, la, lb, lc)


def symm_inputs(dtype, name, m, n, side, uplo, alpha, beta, lda_mul, ldb_mul, ldc_mul):
    """verify_symm's buffers (blas3_symm_test.cpp:54-70)."""
    rng = seeded("symm", np.dtype(dtype).name, name, m, n, side, uplo, alpha, beta, lda_mul, ldb_mul, ldc_mul)
    k = m if side == "l" else n
    lda, ldb, ldc = k * lda_mul, m * ldb_mul, m * ldc_mul
    a = fill_random(rng, k * k * lda_mul, dtype)
    b = fill_random(rng, m * n * ldb_mul, dtype)
    c = fill_random(rng, m * n * ldc_mul, dtype)
    return k, lda, ldb, ldc, a, b, c


def trsm_cases(sizes=(7, 513, 1027)):
    for m, n, trans, side, diag, uplo, unused in itertools.product(sizes, sizes, "nt", "lr", "un", "lu",
                                                                   (0.0, float("nan"))):
        yield (m, n, trans, side, diag, uplo, 2.0, 2, 2, unused)


def fill_trsm_matrix(rng, k, lda, uplo, unit_diag, diag, unused, dtype):
    """test/blas_test.hpp:185-214, same recurrence in float64 then rounded to dtype."""
    A = np.zeros(k * lda, dtype=np.float64)
    draws = rng.uniform(-1.0, 1.0, size=(k, k))
    for i in range(k):
        total = 1.0 if unit_diag == "u" else abs(diag)
        row = np.empty(k)
        for j in range(k):
            if i == j:
                value = diag
            elif i > j:
                value = 0.0
                if total >= 1.0:
                    value = draws[i, j] * (total / np.sqrt(float(k) - float(j)))
                    total -= abs(value)
            else:
                value = unused
            row[j] = value
        cols = np.arange(k)
        if uplo == "l":
            A[i + cols * lda] = row
        else:
            A[cols + i * lda] = row
    return A.astype(dtype)


def trsm_inputs(dtype, m, n, trans, side, diag, uplo, alpha, lda_mul, ldb_mul, unused):
    """run_test's buffers (blas3_trsm_test.cpp:46-63)."""
    rng = seeded("trsm", np.dtype(dtype).name, m, n, trans, side, diag, uplo, alpha, lda_mul, ldb_mul, repr(unused))
    k = m if side == "l" else n
    lda, ldb = k * lda_mul, m * ldb_mul
    diag_value = float(rng.uniform(1.0, 10.0))
    a = fill_trsm_matrix(rng, k, lda, uplo, diag, diag_value, unused, dtype)
    b = fill_random(rng, n * ldb, dtype)
    return k, lda, ldb, a, b


def scipy_symm(side, uplo, m, n, alpha, a, lda, b, ldb, beta, c, ldc):
    """reference_blas::symm (system BLAS) on flat buffers; returns the updated flat C."""
    from scipy.linalg import blas

    fn = blas.ssymm if c.dtype == np.float32 else blas.dsymm
    k = m if side == "l" else n
    isz = c.itemsize
    A = np.lib.stride_tricks.as_strided(a, shape=(k, k), strides=(isz, lda * isz))
    B = np.lib.stride_tricks.as_strided(b, shape=(m, n), strides=(isz, ldb * isz))
    out = c.copy()
    Cv = np.lib.stride_tricks.as_strided(out, shape=(m, n), strides=(isz, ldc * isz))
    res = fn(alpha, np.asfortranarray(A), np.asfortranarray(B), beta=beta, c=np.asfortranarray(Cv),
             side=1 if side == "r" else 0, lower=1 if uplo == "l" else 0)
    Cv[...] = res
    return out


def scipy_trsm(side, uplo, trans, diag, m, n, alpha, a, lda, b, ldb):
    """reference_blas::trsm (system BLAS) on flat buffers; returns the updated flat B."""
    from scipy.linalg import blas

    fn = blas.strsm if b.dtype == np.float32 else blas.dtrsm
    k = m if side == "l" else n
    isz = b.itemsize
    A = np.lib.stride_tricks.as_strided(a, shape=(k, k), strides=(isz, lda * isz))
    # the unused triangle may hold NaN: BLAS never reads it, but keep scipy's finite check quiet
    A = np.nan_to_num(np.asfortranarray(A), nan=0.0)
    out = b.copy()
    Bv = np.lib.stride_tricks.as_strided(out, shape=(m, n), strides=(isz, ldb * isz))
    res = fn(alpha, A, np.asfortranarray(Bv), side=1 if side == "r" else 0, lower=1 if uplo == "l" else 0,
             trans_a=1 if trans == "t" else 0, diag=1 if diag == "u" else 0)
    Bv[...] = res
    return out
