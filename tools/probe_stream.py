"""Developer probe: batches of 64x64x64 items, planner's choice (the walking-workgroup kernel)
against the pinned one-workgroup-per-item kernel -- timing and bitwise agreement."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sycl_blas_amd as sb  # noqa: E402
from sycl_blas_amd import _lib, helper  # noqa: E402


def run(h, dtype, m, n, k, batch, ta, tb, beta, fam, reps):
    sa, sb_, sc = m * k, k * n, m * n
    a_g = helper.allocate(h, dtype, sa * batch)
    b_g = helper.allocate(h, dtype, sb_ * batch)
    c_g = helper.allocate(h, dtype, sc * batch)
    helper.fill_uniform(h, a_g, sa * batch, 1)
    helper.fill_uniform(h, b_g, sb_ * batch, 2)
    helper.fill_uniform(h, c_g, sc * batch, 3)
    lda = k if ta != "n" else m
    ldb = n if tb != "n" else k
    h.set_kernel(fam)
    call = lambda: sb._gemm_strided_batched(h, ta, tb, m, n, k, 1.5, a_g, lda, sa, b_g, ldb, sb_, beta, c_g, m, sc,
                                            batch)
    call()
    got = np.empty(sc * batch, dtype)
    h.wait(helper.copy_to_host(h, c_g, got))
    for _ in range(3):
        call()
    h.wait()
    e0 = h.record_event()
    for _ in range(reps):
        call()
    e1 = h.record_event()
    e1.wait()
    us = e0.elapsed_ms(e1) / reps * 1e3
    for buf in (a_g, b_g, c_g):
        helper.deallocate(buf)
    return us, got


def main():
    with sb.SB_Handle(0) as h:
        for dtype in (np.float32, np.float64):
            for (m, n, k, batch) in ((64, 64, 64, 4096), (64, 64, 64, 5000), (60, 48, 64, 4096), (64, 64, 64, 16384)):
                for ta, tb in (("n", "n"), ("n", "t"), ("t", "n"), ("t", "t")):
                    for beta in (0.0, 0.5):
                        us_a, got_a = run(h, dtype, m, n, k, batch, ta, tb, beta, _lib.KERNEL_AUTO, 20)
                        us_s, got_s = run(h, dtype, m, n, k, batch, ta, tb, beta, _lib.KERNEL_SMALL, 20)
                        same = np.array_equal(got_a.view(np.uint8), got_s.view(np.uint8))
                        bytes_ = (m * k + k * n + m * n * (2 if beta else 1)) * batch * np.dtype(dtype).itemsize
                        print(f"{np.dtype(dtype).name} {m}x{n}x{k} x{batch} {ta}{tb} beta={beta}: stream {us_a:7.1f} us "
                              f"({bytes_ / us_a / 1e6:5.2f} TB/s)  per-item {us_s:7.1f} us ({bytes_ / us_s / 1e6:5.2f} TB/s)  "
                              f"same_bits={same}", flush=True)
                        if not same:
                            sys.exit(1)


if __name__ == "__main__":
    main()
