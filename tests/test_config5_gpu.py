"""Parity for BASELINE config #5 as far as one GPU can run it: DGEMM 65536^3 cut into 2 x 4 C blocks
over 8 GPUs, plus its bf16-input variant.

One rank's share is a 32768 x 16384 x 65536 product on its A row-panel and B column-panel.  The test
keeps the WHOLE 65536^2 A and B resident (32 GiB each, generated on the device; element i of a buffer
depends only on the seed and i, so any element can be replayed on the host) and runs the share of rank
(1, 3) exactly as sycl_blas_amd.distributed.block_gemm addresses it: panels at element offsets inside
the full operands (the B panel starts 3.2e9 elements in), leading dimension 65536.  The oracle cannot
redo 70 TFLOP, so sampled C elements are recomputed from a replayed row of A and column of B, bit for
bit (each C element is one fma chain over all of K)."""
import numpy as np
import pytest

import oracle
import sycl_blas_amd as sb
from sycl_blas_amd import _lib, helper
from sycl_blas_amd.distributed import block_gemm, block_of, grid_for, panel_offsets

from test_full_size_gpu import _replay
from test_gemm_half_gpu import from_bf16_bits, to_bf16_bits

pytestmark = pytest.mark.gpu

G = 65536  # global M = N = K of config #5


def test_config5_one_rank_share_dgemm(sb_handle):
    h = sb_handle
    h.set_kernel(_lib.KERNEL_AUTO)
    pr, pc = grid_for(8)
    assert (pr, pc) == (2, 4)
    rank = 1 * pc + 3  # grid position (1, 3): the last block row, the last block column
    (r0, r1), (c0, c1) = block_of(rank, pr, pc, G, G)
    assert (r0, r1, c0, c1) == (32768, 65536, 49152, 65536)
    a_off, b_off = panel_offsets(rank, pr, pc, "n", "n", G, G, G, G)
    assert a_off == 32768 and b_off == 49152 * G and b_off > 2 ** 31
    bm, bn = r1 - r0, c1 - c0
    alpha, beta = 1.5, 0.5
    a_g = helper.allocate(h, np.float64, G * G)
    b_g = helper.allocate(h, np.float64, G * G)
    c_g = helper.allocate(h, np.float64, bm * bn)  # the rank's block, packed (ldc = 32768)
    try:
        helper.fill_uniform(h, a_g, G * G, 501)
        helper.fill_uniform(h, b_g, G * G, 502)
        helper.fill_uniform(h, c_g, bm * bn, 503)
        seen = {}

        def local_gemm(ta, tb, m, n, k, al, a, lda, b, ldb, be, c, ldc):
            seen["shape"] = (m, n, k, lda, ldb, ldc)
            sb._gemm(h, ta, tb, m, n, k, al, a, lda, b, ldb, be, c, ldc)

        # full A and B, the rank's own packed C block: hand block_gemm the panels it would compute
        block_gemm(local_gemm, rank, pr, pc, "n", "n", G, G, G, alpha, a_g + a_off, G, b_g + b_off, G, beta, c_g, bm,
                   full_operands=False)
        h.wait()
        assert seen["shape"] == (32768, 16384, 65536, G, G, 32768)
        assert h.last_kernel() == _lib.KERNEL_HUGE and h.last_splitk()[0] == 1

        rng = np.random.default_rng(55)
        rows = np.concatenate([[0, bm - 1, 255, 256, 21845], rng.integers(0, bm, 19)])
        cols = np.concatenate([[0, bn - 1, 127, 128, 10923], rng.integers(0, bn, 19)])
        kk = np.arange(G, dtype=np.uint64)
        one = np.empty(1, np.float64)
        for i, j in zip(rows, cols):
            i, j = int(i), int(j)
            a_row = _replay(np.uint64(r0 + i) + kk * np.uint64(G), 501)   # A[r0 + i, :]
            b_col = _replay(kk + np.uint64(c0 + j) * np.uint64(G), 502)   # B[:, c0 + j]
            want = _replay(np.array([i + j * bm], dtype=np.uint64), 503)
            assert oracle.gemm_backend("n", "n", 1, 1, G, alpha, a_row, 1, 0, b_col, G, 0, beta, want, 1, 0) == 0
            h.wait(helper.copy_to_host(h, c_g + (i + j * bm), one, 1))
            assert one.tobytes() == want.tobytes(), f"C[{i},{j}] = {one[0]!r}, expected {want[0]!r}"
        # the generator itself at the far end of a 32 GiB buffer
        h.wait(helper.copy_to_host(h, b_g + (G * G - 1), one, 1))
        assert one[0] == _replay(np.array([G * G - 1], dtype=np.uint64), 502)[0]
    finally:
        h.wait()
        for buf in (a_g, b_g, c_g):
            helper.deallocate(buf)


def _bf16_replay(indices, seed):
    """Host value of sbamd_fill_uniform(dtype bf16) at the given element indices, as float32."""
    return from_bf16_bits(to_bf16_bits(_replay(indices, seed).astype(np.float32)))


def test_config5_bf16_variant_8192_cubed(sb_handle):
    # The bf16-input / fp32-output kernel of config #5's variant at 8192^3 (direct-to-LDS 256 x 256 tiles),
    # both layouts bench.py times.  Column slabs at the start, across a tile boundary and at the end are
    # checked against the fp32 accumulation bound (K + 2) * 2^-24 * sum|a||b| around the float64 product of
    # the same bf16 values, and with the reference's float predicate against the oracle's fp32 chain.
    h = sb_handle
    h.set_kernel(_lib.KERNEL_AUTO)
    n = 8192
    a_g = helper.allocate(h, "bfloat16", n * n)
    b_g = helper.allocate(h, "bfloat16", n * n)
    c_g = helper.allocate(h, "float32", n * n)
    try:
        helper.fill_uniform(h, a_g, n * n, 511)
        helper.fill_uniform(h, b_g, n * n, 512)
        a_bits = np.empty(n * n, np.uint16)
        b_bits = np.empty(n * n, np.uint16)
        h.wait(helper.copy_to_host(h, a_g, a_bits))
        h.wait(helper.copy_to_host(h, b_g, b_bits))
        assert np.array_equal(from_bf16_bits(a_bits[:4096]), _bf16_replay(np.arange(4096, dtype=np.uint64), 511))
        a_f = from_bf16_bits(a_bits)
        b_f = from_bf16_bits(b_bits)
        a64 = a_f.astype(np.float64).reshape(n, n)  # [col, row] of the stored matrix
        c = np.empty(n * n, np.float32)
        for ta, tb in (("t", "n"), ("n", "n")):
            sb._gemm(h, ta, tb, n, n, n, 1.0, a_g, n, b_g, n, 0.0, c_g, n)
            h.wait()
            assert h.last_kernel() == _lib.KERNEL_HUGE
            h.wait(helper.copy_to_host(h, c_g, c))
            assert np.isfinite(c).all()
            op_a = a64 if ta == "t" else a64.T          # op(A)[i, l]
            for j0, cols in ((0, 16), (248, 16), (n - 16, 16)):
                b_cols = b_f[j0 * n:(j0 + cols) * n].astype(np.float64).reshape(cols, n).T  # B[:, j0..]
                truth = op_a @ b_cols
                mag = np.abs(op_a) @ np.abs(b_cols)
                got = c[j0 * n:(j0 + cols) * n].astype(np.float64).reshape(cols, n).T
                err = np.abs(got - truth)
                assert (err <= (n + 2) * 2.0 ** -24 * mag).all(), f"{ta}{tb} columns {j0}..: error {err.max():.3e}"
                want = np.zeros(n * cols, np.float32)
                assert oracle.gemm_backend(ta, tb, n, cols, n, 1.0, a_f, n, 0, b_f, n, 0, 0.0, want, n, 0,
                                           b_off=j0 * n) == 0
                assert oracle.compare_vectors(c[j0 * n:(j0 + cols) * n], want, "float32") == -1
    finally:
        h.wait()
        for buf in (a_g, b_g, c_g):
            helper.deallocate(buf)


def test_config5_bf16_one_rank_share(sb_handle):
    # The bf16 variant at the rank share's size: 32768 x 16384 x 65536 on packed panels (A 4 GiB, B 2 GiB,
    # C 2 GiB).  Sampled elements against the float64 dot product of the replayed bf16 values, inside the
    # fp32 accumulation bound.
    h = sb_handle
    h.set_kernel(_lib.KERNEL_AUTO)
    m, n, k = 32768, 16384, G
    a_g = helper.allocate(h, "bfloat16", m * k)
    b_g = helper.allocate(h, "bfloat16", k * n)
    c_g = helper.allocate(h, "float32", m * n)
    try:
        helper.fill_uniform(h, a_g, m * k, 521)
        helper.fill_uniform(h, b_g, k * n, 522)
        sb._gemm(h, "t", "n", m, n, k, 1.0, a_g, k, b_g, k, 0.0, c_g, m)  # both operands K-contiguous
        h.wait()
        assert h.last_kernel() == _lib.KERNEL_HUGE
        rng = np.random.default_rng(56)
        rows = np.concatenate([[0, m - 1, 255, 256], rng.integers(0, m, 20)])
        cols = np.concatenate([[0, n - 1, 255, 256], rng.integers(0, n, 20)])
        kk = np.arange(k, dtype=np.uint64)
        one = np.empty(1, np.float32)
        for i, j in zip(rows, cols):
            i, j = int(i), int(j)
            a_row = _bf16_replay(kk + np.uint64(i) * np.uint64(k), 521).astype(np.float64)
            b_col = _bf16_replay(kk + np.uint64(j) * np.uint64(k), 522).astype(np.float64)
            truth = float(a_row @ b_col)
            mag = float(np.abs(a_row) @ np.abs(b_col))
            h.wait(helper.copy_to_host(h, c_g + (i + j * m), one, 1))
            assert abs(float(one[0]) - truth) <= (k + 2) * 2.0 ** -24 * mag, f"C[{i},{j}] = {one[0]!r} vs {truth!r}"
            # and well inside the reference's float predicate (relative 5e-3)
            assert abs(float(one[0]) - truth) <= 5e-3 * (abs(truth) + abs(float(one[0])))
    finally:
        h.wait()
        for buf in (a_g, b_g, c_g):
            helper.deallocate(buf)
