"""sbamd_sgemm_noalloc: the SGEMM entry point that allocates no device memory.

It has no counterpart in the reference; it must give, bit for bit, what sbamd_sgemm gives
(which the other tests pin against the oracle).  A split-K call keeps its partial products in
a fixed per-GPU array instead of the handle's on-demand buffer, so the cases here are split-K
shapes: two calls in a row, the second smaller (the array keeps the first call's contents),
then a shape that does not split, then a pinned cut whose cube exceeds the array.  The property in the
name is checked directly: on a handle that has never run anything else (so it owns no cube), free device
memory is identical before and after the calls, and the call is legal under HIP stream capture, where a
hipMalloc is not.
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


def _hip():
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    hip.hipStreamBeginCapture.argtypes = [C.c_void_p, C.c_int]
    hip.hipStreamEndCapture.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    hip.hipGraphInstantiate.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    hip.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipGraphExecDestroy.argtypes = [C.c_void_p]
    hip.hipGraphDestroy.argtypes = [C.c_void_p]
    return hip


def _free_bytes(hip):
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_noalloc_allocates_nothing_on_a_fresh_handle(sb_handle):
    """A handle that has never run a K cut owns no cube: were the entry point to fall back on the handle's
    buffer (or on hipMalloc in any form), free device memory would drop by the cube's 16 MiB."""
    import sycl_blas_amd as sb

    lib, hip = _lib.load(), _hip()
    m, n, k = 256, 256, 65536
    rng = seeded("noalloc-fresh", m, n, k)
    a = rng.uniform(-2.0, 5.0, m * k).astype(np.float32)
    b = rng.uniform(-2.0, 5.0, k * n).astype(np.float32)
    c = rng.uniform(-2.0, 5.0, m * n).astype(np.float32)
    h0 = sb_handle
    a_g, b_g = helper.allocate(h0, np.float32, a.size), helper.allocate(h0, np.float32, b.size)
    c_g = [helper.allocate(h0, np.float32, c.size) for _ in range(3)]
    fresh = sb.SB_Handle(0)
    try:
        helper.copy_to_device(h0, a, a_g)
        helper.copy_to_device(h0, b, b_g)
        for buf in c_g:
            helper.copy_to_device(h0, c, buf)
        h0.wait()

        def call(fn, h, buf):
            st = fn(h._h, b"n", b"n", m, n, k, C.c_float(1.5), a_g.ptr, m, b_g.ptr, k, C.c_float(0.5), buf.ptr, m)
            assert st == _lib.OK, _lib.status_string(st)

        # the kernels' code is loaded (and their attributes set) by a first call through the SAME entry point on
        # the session's handle; that one-time cost belongs to the library, not to the call
        call(lib.sbamd_sgemm_noalloc, h0, c_g[2])
        h0.wait()
        before = _free_bytes(hip)
        call(lib.sbamd_sgemm_noalloc, fresh, c_g[1])  # the fresh handle's very first call
        cut = fresh.last_splitk()
        fresh.wait()
        after = _free_bytes(hip)
        assert cut[0] > 1 and cut[0] * m * n * 4 >= (2 << 20), "the shape must need a cube that an allocation would show"
        assert after == before, f"free device memory changed by {before - after} bytes across a no-allocation call"
        call(lib.sbamd_sgemm, h0, c_g[0])
        assert h0.last_splitk() == cut
        got, want = np.empty_like(c), np.empty_like(c)
        h0.wait(helper.copy_to_host(h0, c_g[1], got))
        h0.wait(helper.copy_to_host(h0, c_g[0], want))
        assert_same_bits(got, want, "noalloc on a fresh handle")
    finally:
        fresh.close()
        h0.wait()
        for buf in (a_g, b_g, *c_g):
            helper.deallocate(buf)


def test_noalloc_runs_under_stream_capture(sb_handle):
    """Callers under HIP graph capture are who the entry point is for: a K-cut call is captured into a graph
    (an allocation would make the capture fail), the graph is launched, and C equals sbamd_sgemm's."""
    import sycl_blas_amd as sb

    lib, hip = _lib.load(), _hip()
    m, n, k = 128, 192, 32768
    rng = seeded("noalloc-capture", m, n, k)
    a = rng.uniform(-2.0, 5.0, m * k).astype(np.float32)
    b = rng.uniform(-2.0, 5.0, k * n).astype(np.float32)
    c = rng.uniform(-2.0, 5.0, m * n).astype(np.float32)
    h = sb.SB_Handle(0)  # its own stream: nothing else is enqueued on it while it captures
    a_g, b_g = helper.allocate(h, np.float32, a.size), helper.allocate(h, np.float32, b.size)
    c_g = [helper.allocate(h, np.float32, c.size) for _ in range(2)]
    graph, graph_exec = C.c_void_p(), C.c_void_p()
    try:
        helper.copy_to_device(h, a, a_g)
        helper.copy_to_device(h, b, b_g)
        for buf in c_g:
            helper.copy_to_device(h, c, buf)

        def call(fn, buf):
            return fn(h._h, b"t", b"n", m, n, k, C.c_float(1.5), a_g.ptr, k, b_g.ptr, k, C.c_float(0.5), buf.ptr, m)

        assert call(lib.sbamd_sgemm, c_g[0]) == _lib.OK  # also loads the kernels before the capture starts
        cut = h.last_splitk()
        assert cut[0] > 1
        h.wait()
        stream = C.c_void_p()
        assert lib.sbamd_get_stream(h._h, C.byref(stream)) == _lib.OK
        assert hip.hipStreamBeginCapture(stream, 0) == 0  # hipStreamCaptureModeGlobal
        try:
            st = call(lib.sbamd_sgemm_noalloc, c_g[1])
        finally:
            end = hip.hipStreamEndCapture(stream, C.byref(graph))
        assert st == _lib.OK, _lib.status_string(st)
        assert end == 0 and graph.value, "the capture must survive the call"
        assert h.last_splitk() == cut
        assert hip.hipGraphInstantiate(C.byref(graph_exec), graph, None, None, 0) == 0
        assert hip.hipGraphLaunch(graph_exec, stream) == 0
        h.wait()
        got, want = np.empty_like(c), np.empty_like(c)
        h.wait(helper.copy_to_host(h, c_g[1], got))
        h.wait(helper.copy_to_host(h, c_g[0], want))
        assert_same_bits(got, want, "noalloc replayed from a captured graph")
    finally:
        if graph_exec.value:
            hip.hipGraphExecDestroy(graph_exec)
        if graph.value:
            hip.hipGraphDestroy(graph)
        h.wait()
        for buf in (a_g, b_g, *c_g):
            helper.deallocate(buf)
        h.close()


def test_noalloc_fits_the_cut_to_the_fixed_array(sb_handle):
    """Left to the planner, a shape whose preferred cube would exceed the array gets the best plan that fits (a
    shallower cut or a single launch) instead of a refusal; the result is then the oracle's for THAT cut."""
    import oracle

    h = sb_handle
    lib = _lib.load()
    h.set_kernel(_lib.KERNEL_AUTO)
    m, n, k = 1280, 1280, 32768  # the planner's own cut for this shape (12 slabs of 6.25 MiB) exceeds 64 MiB
    rng = seeded("noalloc-fit", m, n, k)
    a = rng.uniform(-2.0, 5.0, m * k).astype(np.float32)
    b = rng.uniform(-2.0, 5.0, k * n).astype(np.float32)
    c = rng.uniform(-2.0, 5.0, m * n).astype(np.float32)
    a_g, b_g, c_g = (helper.allocate(h, np.float32, x.size) for x in (a, b, c))
    try:
        for x, g in ((a, a_g), (b, b_g), (c, c_g)):
            helper.copy_to_device(h, x, g)
        st = lib.sbamd_sgemm_noalloc(h._h, b"n", b"n", m, n, k, C.c_float(1.0), a_g.ptr, m, b_g.ptr, k, C.c_float(1.0),
                                     c_g.ptr, m)
        assert st == _lib.OK, _lib.status_string(st)
        depth, k_chunk = h.last_splitk()
        assert 1 < depth and depth * m * n * 4 <= _lib.NOALLOC_SCRATCH_BYTES
        got = np.empty_like(c)
        h.wait(helper.copy_to_host(h, c_g, got))
        want = c.copy()
        form = oracle.FORM_SPLITK if depth > 1 else oracle.FORM_REF
        assert oracle.gemm_backend("n", "n", m, n, k, 1.0, a, m, 0, b, k, 0, 1.0, want, m, 0, form=form, depth=depth,
                                   k_chunk=k_chunk) == 0
        assert_same_bits(got, want, f"noalloc with the cut fitted to the array (depth {depth})")
        # the ordinary entry point is free to cut deeper than the array allows
        st = lib.sbamd_sgemm(h._h, b"n", b"n", m, n, k, C.c_float(1.0), a_g.ptr, m, b_g.ptr, k, C.c_float(0.0), c_g.ptr, m)
        assert st == _lib.OK and h.last_splitk()[0] * m * n * 4 > _lib.NOALLOC_SCRATCH_BYTES
    finally:
        h.wait()
        for buf in (a_g, b_g, c_g):
            helper.deallocate(buf)


def test_noalloc_without_a_k_cut(sb_handle):
    h = sb_handle
    h.set_kernel(_lib.KERNEL_AUTO)
    (want, got), (cut_a, cut_b) = run_both(h, "n", "n", 1000, 900, 300, 1.5, 0.5, 11)
    assert cut_a == cut_b
    assert_same_bits(got, want, "noalloc 1000x900x300")


def test_noalloc_rejects_a_pinned_cut_beyond_the_fixed_array(sb_handle):
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
