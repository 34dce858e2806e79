"""sbamd_sgemm_noalloc: the SGEMM entry point that allocates no device memory.

It has no counterpart in the reference; it must give, bit for bit, what sbamd_sgemm gives
(which the other tests pin against the oracle).  A split-K call keeps its partial products in
a fixed per-GPU array instead of the handle's on-demand buffer, so the cases here are split-K
shapes: two calls in a row, the second smaller (the array keeps the first call's contents),
then a shape that does not split, then one whose cube exceeds the array.
"""
import ctypes as C

import numpy as np
import pytest

from sycl_blas_amd import _lib, helper

from common import assert_same_bits, seeded

pytestmark = pytest.mark.gpu


def run_both(h, transa, transb, m, n, k, alpha, beta, key):
    """The same problem through sbamd_sgemm and sbamd_sgemm_noalloc; returns both C buffers and
    the K cut each call reported."""
    lib = _lib.load()
    rng = seeded("noalloc", key, m, n, k, transa, transb)
    lda = k if transa != "n" else m
    ldb = n if transb != "n" else k
    a = rng.uniform(-2.0, 5.0, m * k).astype(np.float32)
    b = rng.uniform(-2.0, 5.0, k * n).astype(np.float32)
    c = rng.uniform(-2.0, 5.0, m * n).astype(np.float32)
    a_g, b_g = helper.allocate(h, np.float32, a.size), helper.allocate(h, np.float32, b.size)
    c_g = [helper.allocate(h, np.float32, c.size) for _ in range(2)]
    try:
        helper.copy_to_device(h, a, a_g)
        helper.copy_to_device(h, b, b_g)
        for buf in c_g:
            helper.copy_to_device(h, c, buf)
        out, cuts = [], []
        for fn, buf in ((lib.sbamd_sgemm, c_g[0]), (lib.sbamd_sgemm_noalloc, c_g[1])):
            st = fn(h._h, transa.encode(), transb.encode(), m, n, k, C.c_float(alpha), a_g.ptr, lda, b_g.ptr, ldb,
                    C.c_float(beta), buf.ptr, m)
            assert st == _lib.OK, _lib.status_string(st)
            cuts.append(h.last_splitk())
            got = np.empty_like(c)
            h.wait(helper.copy_to_host(h, buf, got))
            out.append(got)
        h.wait()
    finally:
        for buf in (a_g, b_g, *c_g):
            helper.deallocate(buf)
    return out, cuts


def test_noalloc_matches_sgemm_on_two_calls_in_a_row(sb_handle):
    h = sb_handle
    h.set_kernel(_lib.KERNEL_AUTO)
    # first call: a long-K shape the planner cuts into many slabs; second: a smaller one, which
    # finds the first call's partial products still lying in the array
    for key, (m, n, k, ta, tb, alpha, beta) in enumerate([(256, 256, 65536, "n", "n", 1.5, 0.5),
                                                          (65, 126, 20000, "t", "n", 2.0, 0.0)]):
        (want, got), (cut_a, cut_b) = run_both(h, ta, tb, m, n, k, alpha, beta, key)
        assert cut_a == cut_b and cut_a[0] > 1, "both entry points must cut K the same way, and must cut it"
        assert cut_a[0] * m * n * 4 <= _lib.NOALLOC_SCRATCH_BYTES
        assert_same_bits(got, want, f"noalloc {m}x{n}x{k} {ta}{tb}")
    # the handle's own cube is back in place afterwards: the ordinary entry point still splits and agrees
    (want, got), (cut_a, cut_b) = run_both(h, "n", "t", 200, 130, 30000, 1.5, 0.5, 7)
    assert cut_a == cut_b and cut_a[0] > 1
    assert_same_bits(got, want, "noalloc after restoring the handle's buffers")


def test_noalloc_without_a_k_cut(sb_handle):
    h = sb_handle
    h.set_kernel(_lib.KERNEL_AUTO)
    (want, got), (cut_a, cut_b) = run_both(h, "n", "n", 1000, 900, 300, 1.5, 0.5, 11)
    assert cut_a == cut_b
    assert_same_bits(got, want, "noalloc 1000x900x300")


def test_noalloc_rejects_a_cube_beyond_the_fixed_array(sb_handle):
    h = sb_handle
    lib = _lib.load()
    m = n = 2048
    k = 4096
    depth = 8  # 2048 * 2048 * 8 floats = 128 MiB > 64 MiB
    assert m * n * depth * 4 > _lib.NOALLOC_SCRATCH_BYTES
    a_g, b_g, c_g = (helper.allocate(h, np.float32, m * k) for _ in range(3))
    h.set_kernel(_lib.KERNEL_SPLITK, depth | (2 << 16))
    try:
        sentinel = np.full(m * n, 3.0, np.float32)
        h.wait(helper.copy_to_device(h, sentinel, c_g, m * n))
        lib.sbamd_memset(h._h, a_g.ptr, 0, m * k * 4)
        lib.sbamd_memset(h._h, b_g.ptr, 0, m * k * 4)
        st = lib.sbamd_sgemm_noalloc(h._h, b"n", b"n", m, n, k, C.c_float(1.0), a_g.ptr, m, b_g.ptr, k,
                                     C.c_float(0.0), c_g.ptr, m)
        assert st == _lib.SCRATCH_TOO_SMALL, _lib.status_string(st)
        assert "scratch" in _lib.status_string(st)
        got = np.empty_like(sentinel)
        h.wait(helper.copy_to_host(h, c_g, got, m * n))
        assert (got == 3.0).all(), "a refused call must not touch C"
        # the ordinary entry point serves the same call from the handle's own buffer
        st = lib.sbamd_sgemm(h._h, b"n", b"n", m, n, k, C.c_float(1.0), a_g.ptr, m, b_g.ptr, k, C.c_float(0.0),
                             c_g.ptr, m)
        assert st == _lib.OK and h.last_splitk()[0] == depth
        h.wait(helper.copy_to_host(h, c_g, got, m * n))
        assert (got == 0.0).all()
    finally:
        h.set_kernel(_lib.KERNEL_AUTO)
        h.wait()
        for buf in (a_g, b_g, c_g):
            helper.deallocate(buf)
