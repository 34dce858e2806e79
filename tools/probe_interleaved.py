"""Interleaved batches through the raw C ABI: achieved HBM rate (algorithmic bytes = A + B + C written once)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sycl_blas_amd import _lib  # noqa: E402

lib = _lib.load()
side = torch.cuda.Stream()
torch.cuda.set_stream(side)
h = C.c_void_p()
assert lib.sbamd_create(C.byref(h), 0, C.c_void_p(side.cuda_stream)) == 0
for (m, n, k, batch) in [(4, 4, 4, 1 << 20), (8, 8, 8, 1 << 18), (16, 16, 16, 1 << 16), (32, 32, 32, 1 << 13), (64, 64, 64, 4096), (16, 16, 64, 1 << 15)]:
    a = torch.rand(m * k * batch, device="cuda") - 0.5
    b = torch.rand(k * n * batch, device="cuda") - 0.5
    c = torch.zeros(m * n * batch, device="cuda")

    def call():
        st = lib.sbamd_sgemm_batched(h, b"n", b"n", m, n, k, C.c_float(1.0), C.c_void_p(a.data_ptr()), m, C.c_void_p(b.data_ptr()), k,
                                     C.c_float(0.0), C.c_void_p(c.data_ptr()), m, batch, 1)
        assert st == 0, st

    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    bytes_ = 4.0 * batch * (m * k + k * n + m * n)
    print(f"interleaved f32 {m}x{n}x{k} x{batch}: {us:8.1f} us  {bytes_ / us / 1e6:7.2f} TB/s  {2.0 * m * n * k * batch / us / 1e6:7.2f} TFLOP/s  kernel={lib.sbamd_last_kernel(h)}", flush=True)
lib.sbamd_destroy(h)
