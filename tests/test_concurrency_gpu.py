"""Two host threads, each with its own handle (its own stream, scratch areas and planner state), issue GEMMs of
every kind at the same time: one handle per thread is the contract (the reference's SB_Handle wraps one queue,
include/sb_handle/portblas_handle.h:46-60), and the library's process-wide state -- the once-per-device kernel
attributes, the per-device address of the fixed scratch array -- has to hold up under it.  ctypes releases the
interpreter lock during a call, so the calls really overlap."""
import threading

import numpy as np
import pytest

import sycl_blas_amd as sb

from common import verify_gemm, verify_gemm_strided

pytestmark = pytest.mark.gpu

CASES = [
    (np.float32, 0, 1, 300, 200, 129, "n", "t"), (np.float64, 3, 1, 129, 257, 300, "t", "n"),
    (np.float32, 0, 1, 64, 48, 4100, "n", "n"),    # K cut: cube in the handle's own scratch
    (np.float64, 0, 1, 96, 96, 2500, "t", "t"),
    (np.float32, 1, 5, 63, 65, 70, "n", "n"),      # strided batch
    (np.float32, 0, 1, 513, 255, 64, "t", "t"), (np.float64, 0, 1, 255, 513, 31, "n", "n"),
]


def worker(seed, errors):
    try:
        with sb.SB_Handle(0) as h:
            order = np.random.default_rng(seed).permutation(len(CASES))
            for rep in range(3):
                for i in order:
                    dtype, off, batch, m, n, k, ta, tb = CASES[i]
                    verify_gemm(h, dtype, off, batch, m, n, k, ta, tb, 1.5, 0.5, 1, 1, 1)
                verify_gemm_strided(h, np.float32, 0, 3, 40, 33, 65, "n", "t", 1.0, 1.0, 1, 1, 1, 2, 2, 2)
    except BaseException as exc:  # noqa: BLE001 -- reported by the main thread
        errors.append((seed, repr(exc)))


def test_two_threads_two_handles():
    errors = []
    threads = [threading.Thread(target=worker, args=(s, errors)) for s in (1, 2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=280)
    assert not any(t.is_alive() for t in threads), "a worker did not finish"
    assert not errors, errors


def test_scratch_grows_behind_queued_work_without_a_wait():
    """A small K cut, a much larger one and an interleaved batch that is repacked are queued back to back on a
    FRESH handle with no wait in between: the cube and the repack area grow twice while kernels that use the
    smaller buffers are still queued (the old buffers are parked behind an event and released later).  Every
    result must still be the oracle's, bit for bit."""
    import oracle
    from sycl_blas_amd import _lib, helper
    from common import assert_same_bits, fill_random, seeded, to_interleaved, from_interleaved

    with sb.SB_Handle(0) as h:
        rng = seeded("grow-without-wait")
        jobs = []
        # (m, n, k, pinned depth): 0.25 MiB of cube, then 48 MiB, then 6 MiB again
        for m, n, k, depth in ((128, 128, 1024, 4), (1024, 1024, 6144, 12), (512, 256, 2048, 12)):
            a, b, c = (fill_random(rng, cnt, np.float32) for cnt in (m * k, k * n, m * n))
            bufs = [helper.allocate(h, np.float32, x.size) for x in (a, b, c)]
            for host, dev in zip((a, b, c), bufs):
                helper.copy_to_device(h, host, dev, host.size)
            jobs.append((m, n, k, depth, a, b, c, bufs))
        # an interleaved batch big enough to be repacked (grows the second scratch area), queued in between
        bm, bs = 256, 4
        ia, ib, ic = (fill_random(rng, bm * bm * bs, np.float32) for _ in range(3))
        ibufs = [helper.allocate(h, np.float32, bm * bm * bs) for _ in range(3)]
        for host, dev in zip((ia, ib, ic), ibufs):
            helper.copy_to_device(h, to_interleaved(host, 0, bm, bm, bs), dev, host.size)
        cuts = []
        try:
            for i, (m, n, k, depth, a, b, c, bufs) in enumerate(jobs):
                h.set_kernel(_lib.KERNEL_SPLITK, depth | (1 << 16))
                sb._gemm(h, "n", "t", m, n, k, 1.5, bufs[0], m, bufs[1], n, 0.5, bufs[2], m, [])
                cuts.append(h.last_splitk())
                if i == 0:
                    h.set_kernel(_lib.KERNEL_AUTO)
                    sb._gemm_batched(h, "n", "n", bm, bm, bm, 1.0, ibufs[0], bm, ibufs[1], bm, 0.5, ibufs[2], bm, bs,
                                     sb.gemm_batch_type_t.interleaved, [])
                    assert h.last_kernel() != _lib.KERNEL_INTERLEAVED, "the batch must take the repacking route"
            h.set_kernel(_lib.KERNEL_AUTO)
            h.wait()
            for (m, n, k, depth, a, b, c, bufs), (d, chunk) in zip(jobs, cuts):
                assert d > 1
                got = np.empty_like(c)
                h.wait(helper.copy_to_host(h, bufs[2], got, got.size))
                want = c.copy()
                assert oracle.gemm_backend("n", "t", m, n, k, 1.5, a, m, 0, b, n, 0, 0.5, want, m, 0, batch=1, batch_type=0,
                                           form=oracle.FORM_SPLITK, depth=d, k_chunk=chunk) == 0
                assert_same_bits(got, want, f"queued K cut {m}x{n}x{k} depth {d}")
            got = np.empty_like(ic)
            h.wait(helper.copy_to_host(h, ibufs[2], got, got.size))
            want = ic.copy()
            assert oracle.gemm_backend("n", "n", bm, bm, bm, 1.0, ia, bm, bm * bm, ib, bm, bm * bm, 0.5, want, bm, bm * bm,
                                       batch=bs, batch_type=0, form=oracle.FORM_REF, depth=1, k_chunk=bm) == 0
            assert_same_bits(from_interleaved(got, 0, bm, bm, bs), want, "queued interleaved batch")
        finally:
            h.set_kernel(_lib.KERNEL_AUTO)
            h.wait()
            for job in jobs:
                for buf in job[7]:
                    helper.deallocate(buf)
            for buf in ibufs:
                helper.deallocate(buf)
