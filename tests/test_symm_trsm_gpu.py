"""GPU parity of the GEMM callers _symm and _trsm, through the C ABI.

Same parameter grids as test/unittest/blas3/blas3_symm_test.cpp:153-211 and
blas3_trsm_test.cpp:155-168.  Every case is checked twice: bit for bit against
the CPU oracle (same expansion / same inverted blocks / same GEMM sequence, each
GEMM a k-ordered fma chain), and with the reference's predicate against the
system BLAS as the reference's own tests do.
"""
import numpy as np
import pytest

import oracle
import sycl_blas_amd as sb
from blas3_cases import scipy_symm, scipy_trsm, symm_cases, symm_inputs, trsm_cases, trsm_inputs
from common import assert_reference_predicate, assert_same_bits, run_device
from sycl_blas_amd import helper

pytestmark = pytest.mark.gpu


def _oracle_symm(h, side, uplo, m, n, alpha, a, lda, b, ldb, beta, c, ldc):
    """The oracle with the K cut the library reports for the GEMM it just ran (1 = a single kernel)."""
    depth, k_chunk = h.last_splitk() if alpha != 0 else (1, 0)
    form = oracle.FORM_SPLITK if depth > 1 else oracle.FORM_REF
    return oracle.symm(side, uplo, m, n, alpha, a, lda, b, ldb, beta, c, ldc, form=form, depth=depth, k_chunk=k_chunk)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("grid", ["SmallBetaNonZeroLDMatch", "AlphaZero", "OffsetNonZero", "LargeBetaNonZeroLDMatch"])
def test_symm(sb_handle, dtype, grid):
    ran = 0
    for case in symm_cases():
        if case[0] != grid:
            continue
        name, m, n, side, uplo, alpha, beta = case[:7]
        k, lda, ldb, ldc, a, b, c = symm_inputs(dtype, *case)

        def call(a_gpu, b_gpu, c_gpu, deps):
            return sb._symm(sb_handle, side, uplo, m, n, alpha, a_gpu, lda, b_gpu, ldb, beta, c_gpu, ldc, deps)

        got = run_device(sb_handle, call, a, b, c)
        want = c.copy()
        assert _oracle_symm(sb_handle, side, uplo, m, n, alpha, a, lda, b, ldb, beta, want, ldc) == 0
        assert_same_bits(got, want, f"symm {case}")
        assert_reference_predicate(got, scipy_symm(side, uplo, m, n, alpha, a, lda, b, ldb, beta, c, ldc),
                                   f"symm {case} vs system BLAS")
        ran += 1
    assert ran > 0


def test_symm_never_reads_the_other_triangle(sb_handle):
    m, n = 130, 70
    for side in "lr":
        for uplo in "lu":
            k, lda, ldb, ldc, a, b, c = symm_inputs(np.float32, "tri", m, n, side, uplo, 0.75, 2.0, 2, 1, 2)
            A = np.lib.stride_tricks.as_strided(a, shape=(k, k), strides=(4, lda * 4))
            A[np.triu_indices(k, 1) if uplo == "l" else np.tril_indices(k, -1)] = np.nan

            def call(a_gpu, b_gpu, c_gpu, deps):
                return sb._symm(sb_handle, side, uplo, m, n, 0.75, a_gpu, lda, b_gpu, ldb, 2.0, c_gpu, ldc, deps)

            got = run_device(sb_handle, call, a, b, c)
            want = c.copy()
            assert _oracle_symm(sb_handle, side, uplo, m, n, 0.75, a, lda, b, ldb, 2.0, want, ldc) == 0
            assert np.isfinite(got).all()
            assert_same_bits(got, want, f"symm poisoned {side}{uplo}")


def test_symm_argument_errors(sb_handle):
    buf = helper.allocate(sb_handle, np.float32, 64)
    try:
        with pytest.raises(ValueError, match="invalid _uplo"):
            sb._symm(sb_handle, "l", "x", 4, 4, 1.0, buf, 4, buf, 4, 0.0, buf, 4)
        with pytest.raises(ValueError, match="invalid _side"):
            sb._symm(sb_handle, "x", "u", 4, 4, 1.0, buf, 4, buf, 4, 0.0, buf, 4)
        with pytest.raises(ValueError, match="invalid _uplo"):  # uplo is looked at first
            sb._symm(sb_handle, "x", "x", 4, 4, 1.0, buf, 4, buf, 4, 0.0, buf, 4)
    finally:
        helper.deallocate(buf)


def _run_trsm(h, case, dtype):
    m, n, trans, side, diag, uplo, alpha, la, lb, unused = case
    k, lda, ldb, a, b = trsm_inputs(dtype, *case)
    a_gpu = helper.allocate(h, a.dtype, a.size)
    b_gpu = helper.allocate(h, b.dtype, b.size)
    try:
        deps = [helper.copy_to_device(h, a, a_gpu, a.size), helper.copy_to_device(h, b, b_gpu, b.size)]
        h.wait(sb._trsm(h, side, uplo, trans, diag, m, n, alpha, a_gpu, lda, b_gpu, ldb, deps))
        got = np.empty_like(b)
        h.wait(helper.copy_to_host(h, b_gpu, got, got.size))
        h.wait()
    finally:
        helper.deallocate(a_gpu)
        helper.deallocate(b_gpu)
    want = b.copy()
    assert oracle.trsm(side, uplo, trans, diag, m, n, alpha, a, lda, want, ldb) == 0
    assert_same_bits(got, want, f"trsm {np.dtype(dtype).name} {case}")
    assert_reference_predicate(got, scipy_trsm(side, uplo, trans, diag, m, n, alpha, a, lda, b, ldb),
                               f"trsm {case} vs system BLAS")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("side", ["l", "r"])
@pytest.mark.parametrize("uplo", ["l", "u"])
def test_trsm(sb_handle, dtype, side, uplo):
    ran = 0
    for case in trsm_cases():
        if case[3] != side or case[5] != uplo:
            continue
        _run_trsm(sb_handle, case, dtype)
        ran += 1
    assert ran == 72


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("side", ["l", "r"])
def test_trsm_routes(sb_handle, dtype, side):
    """Aligned, ragged and odd shapes (odd ones make the solver read A through an aligned copy), every sweep
    direction, and alpha == 0 (whose first block goes through the GEMM entry point's scaling path).  Every
    route must give the oracle's bits: the oracle is the reference's block-by-block GEMM sequence."""
    seen = set()
    for (m, n), trans, uplo, diag, alpha in [
            ((528, 260), "n", "l", "n", 2.0), ((528, 260), "t", "u", "u", 2.0), ((528, 260), "n", "u", "n", 0.5),
            ((528, 260), "t", "l", "n", 0.5),
            ((1036, 132), "n", "l", "u", 2.0), ((1036, 132), "t", "l", "n", 2.0), ((1036, 132), "n", "u", "n", 2.0),
            ((132, 1036), "n", "l", "n", 2.0), ((132, 1036), "t", "u", "n", 2.0), ((132, 1036), "n", "u", "u", 2.0),
            ((132, 1036), "t", "l", "n", 2.0),
            ((301, 515), "n", "l", "n", 2.0), ((301, 515), "t", "u", "n", 2.0), ((515, 301), "n", "u", "n", 2.0),
            ((515, 301), "t", "l", "u", 2.0),
            ((260, 264), "n", "l", "n", 0.0), ((260, 264), "t", "u", "n", 0.0)]:
        for lda_mul in (1, 2):
            case = (m, n, trans, side, diag, uplo, alpha, lda_mul, 1, float("nan"))
            _run_trsm(sb_handle, case, dtype)
            diag_launches, panel_launches = sb_handle.last_trsm_launches()
            k = m if side == "l" else n
            if alpha == 0.0:
                assert diag_launches == 0 and panel_launches >= 1, case
                seen.add("blocks")
            else:
                assert diag_launches >= 1 and (panel_launches >= 1 or k < 512), case
                seen.add("panels" if panel_launches else "diag")
    assert {"blocks", "panels", "diag"} <= seen


def test_trsm_argument_errors(sb_handle):
    buf = helper.allocate(sb_handle, np.float64, 64)
    try:
        for args, text in [(("l", "l", "n", "n", 0, 4, 1.0, buf, 4, buf, 4), "invalid matrix size argument"),
                           (("l", "l", "n", "n", 4, 4, 1.0, buf, 4, buf, 0), "invalid matrix size argument"),
                           (("x", "l", "n", "n", 4, 4, 1.0, buf, 4, buf, 4), "invalid Side argument"),
                           (("l", "x", "n", "n", 4, 4, 1.0, buf, 4, buf, 4), "invalid Triangle argument"),
                           (("l", "l", "c", "n", 4, 4, 1.0, buf, 4, buf, 4), "invalid Transpose argument"),
                           (("l", "l", "n", "x", 4, 4, 1.0, buf, 4, buf, 4), "invalid Diagonal argument")]:
            with pytest.raises(ValueError, match=text):
                sb._trsm(sb_handle, *args)
    finally:
        helper.deallocate(buf)
