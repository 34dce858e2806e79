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
    assert h.last_kernel() == _lib.KERNEL_STREAM
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


@pytest.mark.parametrize("ta", ["n", "t"])
def test_operand_beyond_2_to_31_elements(sb_handle, ta):
    # A holds 49152 x 49152 = 2.4e9 floats (9.7 GB): element offsets inside the kernels exceed 32 bits
    # (the reference's default index type overflows here).  Sampled C elements are recomputed
    # from single rows of A and columns of B replayed on the host, bit for bit.
    h = sb_handle
    h.set_kernel(_lib.KERNEL_AUTO)
    m = k = 49152
    n = 512
    a_g = device_operand(h, np.float32, m * k, 301)
    b_g = device_operand(h, np.float32, k * n, 302)
    c_g = helper.allocate(h, np.float32, m * n)
    try:
        sb._gemm(h, ta, "n", m, n, k, 1.0, a_g, m if ta == "n" else k, b_g, k, 0.0, c_g, m)
        h.wait()
        depth, k_chunk = h.last_splitk()
        c = to_host(h, c_g, np.float32, m * n)
        assert np.isfinite(c).all()
        rng = np.random.default_rng(9)
        rows = np.concatenate([[0, m - 1, 43691, 43690], rng.integers(0, m, 20)])
        cols = np.concatenate([[0, n - 1, 255, 256], rng.integers(0, n, 20)])
        kk = np.arange(k, dtype=np.uint64)
        for i, j in zip(rows, cols):
            # element (i, l) of op(A): 'n' -> i + l*m, 't' -> l + i*k
            idx = (np.uint64(i) + kk * np.uint64(m)) if ta == "n" else (kk + np.uint64(i) * np.uint64(k))
            a_row = _replay(idx, 301).astype(np.float32)
            b_col = sb.uniform_reference(k, 302, start=int(j) * k).astype(np.float32)
            want = np.zeros(1, dtype=np.float32)
            form = oracle.FORM_SPLITK if depth > 1 else oracle.FORM_REF
            # a 1 x 1 product: op(A) as a row vector (lda = 1), B as a column
            assert oracle.gemm_backend("n", "n", 1, 1, k, 1.0, a_row, 1, 0, b_col, k, 0, 0.0, want, 1, 0, form=form,
                                       depth=depth, k_chunk=k_chunk) == 0
            got = c[int(i) + int(j) * m]
            assert got.tobytes() == want[0].tobytes(), f"C[{i},{j}] = {got!r}, expected {want[0]!r}"
    finally:
        for buf in (a_g, b_g, c_g):
            helper.deallocate(buf)


def _replay(indices, seed, lo=-2.0, hi=5.0):
    """sbamd_fill_uniform's value at arbitrary element indices (uint64 array), as float64."""
    with np.errstate(over="ignore"):
        x = np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + indices
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    u = (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return lo + (hi - lo) * u
