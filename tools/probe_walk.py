"""Developer probe for the tile walk of short-K products (K = 64, M and N multiples of 64): the automatic route
against the pinned 256-wide tile family -- same bits, times of both.
usage: probe_walk.py [M=8192] [N=8192]"""
import os
import sys

import ctypes as C

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sycl_blas_amd as sb  # noqa: E402
from sycl_blas_amd import _lib, helper  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
k = 64
HUGE = _lib.KERNEL_HUGE if hasattr(_lib, "KERNEL_HUGE") else 5


def timed(h, call, reps=50):
    for _ in range(5):
        call()
    h.wait()
    e0 = h.record_event()
    for _ in range(reps):
        call()
    e1 = h.record_event()
    e1.wait()
    return e0.elapsed_ms(e1) / reps * 1e3


with sb.SB_Handle(0) as h:
    for dtype in (np.float32, np.float64):
        a_g = helper.allocate(h, dtype, m * k)
        b_g = helper.allocate(h, dtype, k * n)
        c0_g = helper.allocate(h, dtype, m * n)
        c1_g = helper.allocate(h, dtype, m * n)
        helper.fill_uniform(h, a_g, m * k, 1)
        helper.fill_uniform(h, b_g, k * n, 2)
        for ta, tb in ("nn", "nt", "tn", "tt"):
            lda = k if ta != "n" else m
            ldb = n if tb != "n" else k
            for beta in (0.0, 0.75):
                helper.fill_uniform(h, c0_g, m * n, 3)
                helper.fill_uniform(h, c1_g, m * n, 3)
                h.set_kernel(0, 0)
                h.wait(sb._gemm(h, ta, tb, m, n, k, 1.5, a_g, lda, b_g, ldb, beta, c0_g, m))
                kern = h.last_kernel()
                h.set_kernel(HUGE, 0)
                h.wait(sb._gemm(h, ta, tb, m, n, k, 1.5, a_g, lda, b_g, ldb, beta, c1_g, m))
                got, want = np.empty(m * n, dtype), np.empty(m * n, dtype)
                h.wait(helper.copy_to_host(h, c0_g, got))
                h.wait(helper.copy_to_host(h, c1_g, want))
                same = np.array_equal(got.view(np.uint8), want.view(np.uint8))
                # timing straight through the C ABI (the Python mirror's per-call work would be the bottleneck)
                lib = _lib.load()
                fn = lib.sbamd_sgemm if dtype == np.float32 else lib.sbamd_dgemm
                sc = C.c_float if dtype == np.float32 else C.c_double
                raw = lambda c_buf: fn(h._h, ta.encode(), tb.encode(), m, n, k, sc(1.5), a_g.ptr, lda, b_g.ptr, ldb, sc(beta),
                                       c_buf.ptr, m)
                h.set_kernel(0, 0)
                us_auto = timed(h, lambda: raw(c0_g))
                h.set_kernel(HUGE, 0)
                us_huge = timed(h, lambda: raw(c1_g))
                h.set_kernel(0, 0)
                bytes_c = m * n * np.dtype(dtype).itemsize * (2 if beta else 1)
                print(f"{np.dtype(dtype).name} {ta}{tb} {m}x{n}x{k} beta={beta}: auto kernel={kern} {us_auto:7.1f} us "
                      f"({bytes_c / us_auto / 1e6:.2f} TB/s of C) | pinned 256-wide tile {us_huge:7.1f} us | "
                      f"{'same bits' if same else 'BITS DIFFER'}", flush=True)
                if not same:
                    sys.exit(1)
