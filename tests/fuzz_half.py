"""Randomised walk over the 16-bit GEMM paths (every storage pair, tile family pinned or automatic, ragged and
vector-aligned shapes, K with and without tails, odd leading dimensions and offsets), each case checked like the
16-bit parity tests.  tests/test_fuzz_slices_gpu.py runs a fixed-seed slice under pytest; by hand on a GPU box the
whole walk is  python tests/fuzz_half.py [seed] [cases]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sycl_blas_amd as sb  # noqa: E402
from sycl_blas_amd import _lib  # noqa: E402

from test_gemm_half_gpu import PAIRS, run_case  # noqa: E402


def walk_half(h, rng, cases, log=None):
    """Returns {family that ran: count}."""
    seen = {}
    for i in range(cases):
        pair = PAIRS[int(rng.integers(0, len(PAIRS)))]
        m = int(rng.integers(1, 80)) * 8 if rng.integers(0, 4) else int(rng.integers(1, 600))
        n = int(rng.integers(1, 80)) * 8 if rng.integers(0, 4) else int(rng.integers(1, 600))
        k = int(rng.choice([int(rng.integers(1, 40)) * 32, int(rng.integers(1, 150)) * 8, int(rng.integers(1, 1500))]))
        ta, tb = str(rng.choice(["n", "t"])), str(rng.choice(["n", "t"]))
        alpha = float(rng.choice([1.0, 1.5, -0.75]))
        beta = float(rng.choice([0.0, 1.0, 0.5]))
        ldm = int(rng.choice([1, 1, 2]))
        offset = int(rng.choice([0, 0, 8, 7]))
        family, param = [(_lib.KERNEL_AUTO, 0), (_lib.KERNEL_HUGE, 0), (_lib.KERNEL_LARGE, 0), (_lib.KERNEL_LARGE, 1),
                         (_lib.KERNEL_SMALL, 0)][int(rng.integers(0, 5))]
        batch = int(rng.integers(2, 5)) if rng.integers(0, 6) == 0 else 1
        if batch > 1:
            m, n, k = min(m, 200), min(n, 200), min(k, 400)
        run_case(h, pair, m, n, k, ta, tb, alpha, beta, ldm=ldm, offset=offset, batch=batch, family=family, param=param)
        ran = h.last_kernel()
        seen[ran] = seen.get(ran, 0) + 1
        if log and (i + 1) % 50 == 0:
            log(f"{i + 1} cases ok; kernels that ran: {dict(sorted(seen.items()))}")
    return seen


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    rng = np.random.default_rng(seed)
    with sb.SB_Handle(0) as h:
        walk_half(h, rng, cases, lambda line: print(line, flush=True))
    print("fuzz_half ok", seed, cases)


if __name__ == "__main__":
    main()
