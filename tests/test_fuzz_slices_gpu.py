"""Fixed-seed slices of the hand-run randomised walks (tests/fuzz_extended.py, tests/fuzz_half.py), so the
driver-run suite reaches the kernel paths those walks were written for -- and asserts that it did: the 96 / 192
tiles through the planner's own choice, pinned K cuts of every kind, the batch-walking kernel, every 16-bit family.
Every case is checked like the fixed grids: fp32 / fp64 whole-buffer bitwise against the oracle plus the
reference's predicate against OpenBLAS; 16-bit pairs the reference predicate plus the fp32 accumulation bound."""
import numpy as np
import pytest

import sycl_blas_amd as sb
from sycl_blas_amd import _lib

import fuzz_extended
import fuzz_half

pytestmark = pytest.mark.gpu


def test_general_walk_slice(sb_handle):
    seen = fuzz_extended.walk_general(sb_handle, np.random.default_rng(3), 60)
    # pinned 64 x 64, pinned 96 / 192, K cuts and plain single launches all occurred
    assert seen.get(_lib.KERNEL_SPLITK, 0) >= 5 and seen.get(_lib.KERNEL_MEDIUM, 0) >= 3 and seen.get(_lib.KERNEL_SMALL, 0) >= 3, seen


def test_planner_reaches_the_medium_tiles_by_itself(sb_handle):
    seen = fuzz_extended.walk_mid(sb_handle, np.random.default_rng(4), 14)
    assert seen.get(_lib.KERNEL_MEDIUM, 0) >= 3, seen  # KERNEL_AUTO took the 96- or 192-row tile


def test_long_batches_walk_slice(sb_handle):
    seen = fuzz_extended.walk_long_batches(sb_handle, np.random.default_rng(5), 8)
    assert seen.get(_lib.KERNEL_STREAM, 0) >= 2, seen


def test_half_walk_slice(sb_handle):
    seen = fuzz_half.walk_half(sb_handle, np.random.default_rng(6), 50)
    assert len(seen) >= 3 and (seen.get(_lib.KERNEL_HUGE, 0) + seen.get(_lib.KERNEL_LARGE, 0)) >= 5, seen
