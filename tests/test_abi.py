"""The C-ABI library loads and exports every symbol include/sbamd_gemm.h declares
(no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

import pytest

import sycl_blas_amd as sb
from sycl_blas_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sbamd_gemm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sbamd_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    names = declared_symbols()
    assert len(names) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in names:
        assert hasattr(lib, name), f"{name} is declared in the header but not exported"
    assert sorted(_lib.SIGNATURES) == names, "Python prototypes and the header disagree"


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
