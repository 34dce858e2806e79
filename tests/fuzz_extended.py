"""Longer randomised walk than test_gemm_fuzz_gpu.py, biased towards the shapes the newer kernel paths take
(vector-aligned M / N / K, every transpose, K with and without tails, forced K cuts).  The walks are functions:
tests/test_fuzz_slices_gpu.py runs a fixed-seed slice of each under pytest; by hand on a GPU box the whole walk
is  python tests/fuzz_extended.py [seed] [cases]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sycl_blas_amd as sb  # noqa: E402
from sycl_blas_amd import _lib  # noqa: E402

from common import verify_gemm, verify_gemm_strided  # noqa: E402


def walk_general(h, rng, cases, log=None):
    """Pinned and automatic families, K cuts of every kind, small batches.  Returns {family: count}."""
    seen = {}
    for i in range(cases):
        dtype = np.float32 if rng.integers(0, 3) else np.float64
        v = 16 // np.dtype(dtype).itemsize
        m = int(rng.integers(1, 160)) * v if rng.integers(0, 4) else int(rng.integers(1, 600))
        n = int(rng.integers(1, 160)) * v if rng.integers(0, 4) else int(rng.integers(1, 600))
        k = int(rng.choice([int(rng.integers(1, 40)) * 32, int(rng.integers(1, 300)) * 2 * v, int(rng.integers(1, 3000))]))
        ta, tb = str(rng.choice(["n", "t"])), str(rng.choice(["n", "t"]))
        alpha = float(rng.choice([1.0, 1.5, -0.75]))
        beta = float(rng.choice([0.0, 1.0, 0.5]))
        ldm = int(rng.choice([1, 1, 2]))
        offset = int(rng.choice([0, 0, v, 7]))
        mode = int(rng.integers(0, 7))
        if mode == 0:
            h.set_kernel(_lib.KERNEL_SMALL)
        elif mode == 1:
            h.set_kernel(_lib.KERNEL_SPLITK, int(rng.choice([2, 3, 4, 5])) | (int(rng.choice([1, 2, 3, 6, 7])) << 16) | (int(rng.integers(0, 2)) << 24))
        elif mode == 6:
            h.set_kernel(_lib.KERNEL_MEDIUM, int(rng.integers(0, 2)))
        else:
            h.set_kernel(_lib.KERNEL_AUTO)
        try:
            if mode == 5:
                batch = int(rng.integers(2, 6))
                verify_gemm(h, dtype, offset, batch, min(m, 200), min(n, 200), min(k, 400), ta, tb, alpha, beta, ldm, 1, 1)
            else:
                verify_gemm(h, dtype, offset, 1, m, n, k, ta, tb, alpha, beta, ldm, ldm, ldm)
        finally:
            h.set_kernel(_lib.KERNEL_AUTO)
        ran = h.last_kernel()
        seen[ran] = seen.get(ran, 0) + 1
        if mode == 0:
            assert ran in (_lib.KERNEL_SMALL, _lib.KERNEL_STREAM), (ran, m, n, k)
        if mode == 6:
            assert ran == _lib.KERNEL_MEDIUM, (ran, m, n, k)
        if log and i % 25 == 0:
            log(f"case {i}: {np.dtype(dtype).name} {m}x{n}x{k} {ta}{tb} mode {mode} kernel {ran} cut {h.last_splitk()}")
    return seen


def walk_mid(h, rng, cases, log=None):
    """Mid-size products under the planner's own choice.  Every second case has at most one 96 x 96 tile per
    compute unit of a 256-unit part (15 or 16 tiles along each edge), where the planner takes that tile."""
    seen = {}
    h.set_kernel(_lib.KERNEL_AUTO)
    for i in range(cases):
        dtype = np.float32 if rng.integers(0, 2) else np.float64
        m, n = int(rng.integers(600, 1800)), int(rng.integers(600, 1800))
        if rng.integers(0, 2):
            m, n = (m + 95) // 96 * 96, (n + 95) // 96 * 96
        k = int(rng.integers(40, 260))
        if i % 2:
            m, n, k = int(rng.choice([1440, 1536])), int(rng.choice([1440, 1536])), int(rng.integers(64, 600))
        ta, tb = str(rng.choice(["n", "t"])), str(rng.choice(["n", "t"]))
        verify_gemm(h, dtype, 0, 1, m, n, k, ta, tb, 1.5, float(rng.choice([0.0, 0.5])), 1, 1, 1)
        ran = h.last_kernel()
        seen[ran] = seen.get(ran, 0) + 1
        if log:
            log(f"mid case {i}: {np.dtype(dtype).name} {m}x{n}x{k} {ta}{tb} kernel {ran} cut {h.last_splitk()}")
    return seen


def walk_long_batches(h, rng, cases, log=None):
    """Long batches of single-tile items (the batch-walking kernel wherever the operands allow it)."""
    seen = {}
    h.set_kernel(_lib.KERNEL_AUTO)
    for i in range(cases):
        dtype = np.float32 if rng.integers(0, 2) else np.float64
        v = 16 // np.dtype(dtype).itemsize
        m, n = int(rng.integers(1, 17)) * v, int(rng.integers(1, 17)) * v
        m, n = min(m, 64), min(n, 64)
        ta, tb = str(rng.choice(["n", "t"])), str(rng.choice(["n", "t"]))
        batch = int(rng.integers(1030, 2400))
        verify_gemm_strided(h, dtype, int(rng.choice([0, v])), batch, m, n, 64, ta, tb, 1.5, float(rng.choice([0.0, 0.5])),
                            int(rng.choice([1, 2])), 1, int(rng.choice([1, 2])), int(rng.choice([0, 1, 2])), 1, int(rng.choice([1, 2])))
        ran = h.last_kernel()
        seen[ran] = seen.get(ran, 0) + 1
        if log:
            log(f"batch case {i}: {np.dtype(dtype).name} {m}x{n}x64 x{batch} {ta}{tb} kernel {ran}")
    return seen


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    rng = np.random.default_rng(seed)

    def log(line):
        print(line, flush=True)

    with sb.SB_Handle(0) as h:
        print(walk_general(h, rng, cases, log))
        print(walk_mid(h, rng, 16, log))
        print(walk_long_batches(h, rng, 12, log))
    print("fuzz ok")


if __name__ == "__main__":
    main()
