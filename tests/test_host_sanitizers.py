"""The C ABI's host logic under AddressSanitizer + UndefinedBehaviorSanitizer, without a GPU: the library's sources
compiled for the host only and linked against a stand-in HIP runtime whose device memory is host memory
(tests/host_san/).  Any sanitizer report, leaked device buffer, illegal launch configuration or unexpected status
fails the run."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_san")


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("make") is None, reason="needs hipcc and make")
def test_c_abi_host_logic_is_clean_under_asan_and_ubsan():
    build = subprocess.run(["make", "-C", HERE, "-j8", "all"], capture_output=True, text=True)
    assert build.returncode == 0, build.stdout[-3000:] + build.stderr[-3000:]
    run = subprocess.run(["make", "-C", HERE, "run"], capture_output=True, text=True, timeout=300)
    out = run.stdout + run.stderr
    assert run.returncode == 0, out[-6000:]
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out and "LeakSanitizer" not in out, out[-6000:]
    assert "0 expectations failed" in out
