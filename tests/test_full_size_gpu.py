"""Parity at BASELINE.json's full sizes.  The oracle cannot redo 1.1 TFLOP per case in
seconds, so each case checks a slab of C bit for bit (every C column is an independent
fma chain over all of K, so a column slab exercises the full depth of the problem and
all the tiles that own those columns) plus cheap whole-matrix properties."""
import numpy as np
import pytest

import oracle
import sycl_blas_amd as sb
from sycl_blas_amd import _lib, helper

pytestmark = pytest.mark.gpu


def device_operand(h, dtype, count, seed):
    buf = helper.allocate(h, dtype, count)
    helper.fill_uniform(h, buf, count, seed)
    return buf


def to_host(h, buf, dtype, count, offset=0):
    out = np.empty(count, dtype=dtype)
    h.wait(helper.copy_to_host(h, buf + offset, out, count))
    return out


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["sgemm", "dgemm"])
@pytest.mark.parametrize("trans", ["nn", "nt"])
def test_config2_8192_cubed(sb_handle, dtype, trans):
    # BASELINE config #2: S/DGEMM 8192^3 NN and NT.  Slab = 64 columns at the start, 64 in the
    # middle (straddling a tile boundary) and the last 32.
    h = sb_handle
    h.set_kernel(_lib.KERNEL_AUTO)
    n = 8192
    ta, tb = trans
    a_g = device_operand(h, dtype, n * n, 101)
    b_g = device_operand(h, dtype, n * n, 102)
    c_g = helper.allocate(h, dtype, n * n)
    sb._gemm(h, ta, tb, n, n, n, 1.0, a_g, n, b_g, n, 0.0, c_g, n)
    h.wait()
    assert h.last_kernel() == _lib.KERNEL_HUGE
    a = to_host(h, a_g, dtype, n * n)
    b = to_host(h, b_g, dtype, n * n)
    c = to_host(h, c_g, dtype, n * n)
    assert np.isfinite(c).all()
    for c0, cols in ((0, 64), (4064, 64), (8160, 32)):
        want = np.zeros(n * cols, dtype=dtype)
        if tb == "n":   # B is k x n: columns c0.. are contiguous
            st = oracle.gemm_backend(ta, tb, n, cols, n, 1.0, a, n, 0, b, n, 0, 0.0, want, n, 0, b_off=c0 * n)
        else:           # B is stored n x k: rows c0.. of every column
            st = oracle.gemm_backend(ta, tb, n, cols, n, 1.0, a, n, 0, b, n, 0, 0.0, want, n, 0, b_off=c0)
        assert st == 0
        got = c[c0 * n:(c0 + cols) * n]
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), f"columns {c0}..{c0 + cols}"
    # linearity property on the whole matrix: C(2A, B) == 2*C(A, B) exactly (power-of-two scaling)
    c2_g = helper.allocate(h, dtype, n * n)
    sb._gemm(h, ta, tb, n, n, n, 2.0, a_g, n, b_g, n, 0.0, c2_g, n)
    c2 = to_host(h, c2_g, dtype, n * n)
    assert np.array_equal(c2, 2 * c)
    for buf in (a_g, b_g, c_g, c2_g):
        helper.deallocate(buf)


def test_config1_1024_cubed(sb_handle):
    # BASELINE config #1 (the sample's call at 1024^3): whole matrix, bit for bit.
    h = sb_handle
    n = 1024
    a_g = device_operand(h, np.float32, n * n, 11)
    b_g = device_operand(h, np.float32, n * n, 12)
    c_g = device_operand(h, np.float32, n * n, 13)
    a, b, c0 = (to_host(h, x, np.float32, n * n) for x in (a_g, b_g, c_g))
    sb._gemm(h, "n", "n", n, n, n, 1.5, a_g, n, b_g, n, 0.5, c_g, n)
    got = to_host(h, c_g, np.float32, n * n)
    want = c0.copy()
    depth, k_chunk = h.last_splitk()  # replay the K cut if the planner chose one
    form = oracle.FORM_SPLITK if depth > 1 else oracle.FORM_REF
    assert oracle.gemm_backend("n", "n", n, n, n, 1.5, a, n, 0, b, n, 0, 0.5, want, n, 0, form=form, depth=depth,
                               k_chunk=k_chunk) == 0
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
    # the device generator matches its host replay
    assert np.array_equal(a, sb.uniform_reference(n * n, 11).astype(np.float32))


def test_config3_full_batch(sb_handle):
    # BASELINE config #3: strided-batched SGEMM, batch 4096 of 64x64x64, strides 4096 elements.
    h = sb_handle
    m, batch = 64, 4096
    per = m * m
    a_g = device_operand(h, np.float32, per * batch, 21)
    b_g = device_operand(h, np.float32, per * batch, 22)
    c_g = helper.allocate(h, np.float32, per * batch)
    sb._gemm_strided_batched(h, "n", "n", m, m, m, 1.0, a_g, m, per, b_g, m, per, 0.0, c_g, m, per, batch)
    h.wait()
    assert h.last_kernel() == _lib.KERNEL_SMALL
    a, b, got = (to_host(h, x, np.float32, per * batch) for x in (a_g, b_g, c_g))
    want = np.zeros(per * batch, np.float32)
    assert oracle.gemm_backend("n", "n", m, m, m, 1.0, a, m, per, b, m, per, 0.0, want, m, per, batch=batch) == 0
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))


def test_config4_full_k(sb_handle):
    # BASELINE config #4: SGEMM 256 x 256 x 4194304 through split-K; 16 columns checked bit for bit
    # with the oracle's split-K form and the cut the library reports.
    h = sb_handle
    h.set_kernel(_lib.KERNEL_AUTO)
    m = n = 256
    k = 1 << 22
    a_g = device_operand(h, np.float32, m * k, 41)
    b_g = device_operand(h, np.float32, k * n, 42)
    c_g = device_operand(h, np.float32, m * n, 43)
    c0 = to_host(h, c_g, np.float32, m * n)
    sb._gemm(h, "n", "n", m, n, k, 1.5, a_g, m, b_g, k, 0.5, c_g, m)
    h.wait()
    assert h.last_kernel() == _lib.KERNEL_SPLITK
    depth, k_chunk = h.last_splitk()
    assert depth > 1 and depth * k_chunk >= k
    got = to_host(h, c_g, np.float32, m * n)
    assert np.isfinite(got).all()
    cols = 16
    a = to_host(h, a_g, np.float32, m * k)
    b = to_host(h, b_g, np.float32, k * cols)
    want = c0[:m * cols].copy()
    st = oracle.gemm_backend("n", "n", m, cols, k, 1.5, a, m, 0, b, k, 0, 0.5, want, m, 0, form=oracle.FORM_SPLITK,
                             depth=depth, k_chunk=k_chunk)
    assert st == 0
    assert np.array_equal(got[:m * cols].view(np.uint8), want.view(np.uint8))
    # and within the reference's float predicate of the plain k-ordered chain
    plain = c0[:m * cols].copy()
    oracle.gemm_backend("n", "n", m, cols, k, 1.5, a, m, 0, b, k, 0, 0.5, plain, m, 0)
    assert oracle.compare_vectors(got[:m * cols], plain) == -1
    for buf in (a_g, b_g, c_g):
        helper.deallocate(buf)
