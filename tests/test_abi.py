"""The C-ABI library loads and exports every symbol include/sbamd_gemm.h and include/sbamd_dist.h
declare (no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

import pytest

import sycl_blas_amd as sb
from sycl_blas_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="sbamd_gemm.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sbamd_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    names = declared_symbols()
    assert len(names) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in names:
        assert hasattr(lib, name), f"{name} is declared in the header but not exported"
    assert sorted(_lib.SIGNATURES) == names, "Python prototypes and the header disagree"
    dist = declared_symbols("sbamd_dist.h")
    assert len(dist) >= 18
    for name in dist:
        assert hasattr(lib, name), f"{name} is declared in sbamd_dist.h but not exported"
    assert sorted(_lib.DIST_SIGNATURES) == dist, "Python prototypes and sbamd_dist.h disagree"


def test_partition_helpers_need_no_device():
    # the C side of sycl_blas_amd.distributed's partition arithmetic (grid, balanced ranges, bands)
    from sycl_blas_amd import distributed as d

    assert [d.grid_for(w) for w in (1, 2, 4, 8, 6, 9)] == [(1, 1), (1, 2), (2, 2), (2, 4), (2, 3), (3, 3)]
    assert [d.split_range(10, 4, i) for i in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert d.block_of(5, 2, 4, 65536, 65536) == ((32768, 65536), (16384, 32768))
    # shrinking halves, multiples of 64, a small last band; one rank multiplies in one piece
    assert d.band_widths(1024, 0, 8) == [512, 256, 128, 128]
    assert d.band_widths(4096, 0, 2) == [2048, 1024, 512, 512]
    assert d.band_widths(1000, 0, 4) == [512, 256, 232]
    assert d.band_widths(1000, 3, 4) == [334, 333, 333]
    assert d.band_widths(8192, 0, 1) == [8192]
    assert d.band_widths(100, 0, 8) == [100]


def test_status_strings_match_reference_messages():
    # src/interface/gemm_interface.hpp:144-165
    assert _lib.status_string(_lib.INVALID_TRANSA) == "invalid _TransA"
    assert _lib.status_string(_lib.INVALID_TRANSB) == "invalid _TransB"
    assert _lib.status_string(_lib.INVALID_STRIDEA) == "invalid _stridea"
    assert _lib.status_string(_lib.INVALID_STRIDEB) == "invalid _strideb"
    assert _lib.status_string(_lib.INVALID_STRIDEC) == "invalid _stridec"
    assert _lib.status_string(_lib.OK) == "success"


def test_null_handle_is_rejected_without_touching_a_device():
    lib = _lib.load()
    assert lib.sbamd_wait(None) == _lib.INVALID_HANDLE
    assert lib.sbamd_sgemm(None, b"n", b"n", 1, 1, 1, 1.0, None, 1, None, 1, 0.0, None, 1) == _lib.INVALID_HANDLE
    assert lib.sbamd_last_kernel(None) == -1


def test_no_silent_cpu_fallback():
    """Without a GPU the handle cannot be created and says so; nothing computes on the host."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(sb.SbamdError):
        sb.SB_Handle(0)


def test_uniform_reference_is_deterministic():
    a = sb.uniform_reference(1000, 12345)
    b = sb.uniform_reference(1000, 12345)
    assert (a == b).all() and a.min() >= -2.0 and a.max() < 5.0
    assert abs(a.mean() - 1.5) < 0.3
    assert (sb.uniform_reference(10, 12345, start=990) == a[990:]).all()
