"""Randomised GEMM shapes through whatever the planner picks (tile family, K cut), bit for bit
against the oracle.  The fixed grids of the reference's tests (test_gemm_gpu.py) pin the
documented cases; this file walks the planner's decision space: shapes whose tiles hang over
both edges, K ranges with and without a tail, vector-aligned and unaligned leading
dimensions, split and unsplit launches."""
import numpy as np
import pytest

from sycl_blas_amd import _lib

from common import verify_gemm

pytestmark = pytest.mark.gpu


def _draw(rng, hi_mn, hi_k):
    # mix of round and ragged sizes
    def size(hi):
        kind = rng.integers(0, 4)
        if kind == 0:
            return int(rng.choice([1, 2, 3, 4, 8, 16, 32, 64, 128, 256, 512]))
        if kind == 1:
            return int(rng.integers(1, 70))
        return int(rng.integers(1, hi + 1))
    return min(size(hi_mn), hi_mn), min(size(hi_mn), hi_mn), min(size(hi_k) * int(rng.choice([1, 1, 4])), hi_k)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["float32", "float64"])
def test_random_shapes(sb_handle, dtype):
    rng = np.random.default_rng(20240607)
    sb_handle.set_kernel(_lib.KERNEL_AUTO)
    families, splits = set(), 0
    for _ in range(150):
        m, n, k = _draw(rng, 600, 6000)
        ta, tb = rng.choice(["n", "t"]), rng.choice(["n", "t"])
        alpha = float(rng.choice([1.0, 1.5, -0.75]))
        beta = float(rng.choice([0.0, 1.0, 0.5]))
        ldm = int(rng.choice([1, 1, 2, 3]))
        offset = int(rng.choice([0, 0, 4, 7]))
        verify_gemm(sb_handle, dtype, offset, 1, m, n, k, str(ta), str(tb), alpha, beta, ldm, ldm, ldm)
        families.add(sb_handle.last_kernel())
        splits += sb_handle.last_splitk()[0] > 1 and sb_handle.last_kernel() == _lib.KERNEL_SPLITK
    # the walk must actually have exercised both single launches and K cuts
    assert _lib.KERNEL_SPLITK in families and len(families) >= 2 and splits >= 10


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["float32", "float64"])
def test_random_batched_shapes(sb_handle, dtype):
    rng = np.random.default_rng(77)
    sb_handle.set_kernel(_lib.KERNEL_AUTO)
    for _ in range(40):
        m, n, k = _draw(rng, 200, 300)
        batch = int(rng.integers(2, 9))
        ta, tb = rng.choice(["n", "t"]), rng.choice(["n", "t"])
        verify_gemm(sb_handle, dtype, int(rng.choice([0, 5])), batch, m, n, k, str(ta), str(tb), 1.5, float(rng.choice([0.0, 0.5])),
                    int(rng.choice([1, 2])), 1, 1)
        assert sb_handle.last_kernel() != _lib.KERNEL_SPLITK  # K is never cut for batches
