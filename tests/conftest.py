import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def sb_handle():
    """One handle for the whole session, like the reference's cached queue
    (test/blas_test.hpp:96-100)."""
    import sycl_blas_amd as sb

    h = sb.SB_Handle(0)
    yield h
    h.close()
