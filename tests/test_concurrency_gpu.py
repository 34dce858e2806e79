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
