"""Strided batches through the raw C ABI: achieved rate against the algorithmic bytes (A + B + C written once) and flops."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sycl_blas_amd import _lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "f32"
lib = _lib.load()
side = torch.cuda.Stream()
torch.cuda.set_stream(side)
h = C.c_void_p()
assert lib.sbamd_create(C.byref(h), 0, C.c_void_p(side.cuda_stream)) == 0
dt, fn, sc, isz = {"f32": (torch.float32, lib.sbamd_sgemm_batched, C.c_float, 4), "f64": (torch.float64, lib.sbamd_dgemm_batched, C.c_double, 8),
                   "f16": (torch.float16, lib.sbamd_hgemm_batched, C.c_float, 2)}[name]
for (m, n, k, batch) in [(4, 4, 4, 1 << 20), (8, 8, 8, 1 << 18), (16, 16, 16, 1 << 16), (32, 32, 32, 1 << 13), (48, 48, 48, 4096),
                         (64, 64, 64, 4096), (96, 96, 96, 2048), (128, 128, 128, 1024), (256, 256, 256, 256), (512, 512, 512, 64),
                         (64, 64, 512, 2048), (32, 32, 1024, 4096), (100, 100, 100, 2000)]:
    a = (torch.rand(m * k * batch, device="cuda") - 0.5).to(dt)
    b = (torch.rand(k * n * batch, device="cuda") - 0.5).to(dt)
    c = torch.zeros(m * n * batch, device="cuda", dtype=dt)

    def call():
        st = fn(h, b"n", b"n", m, n, k, sc(1.0), C.c_void_p(a.data_ptr()), m, C.c_void_p(b.data_ptr()), k, sc(0.0),
                C.c_void_p(c.data_ptr()), m, batch, 0)
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
    bytes_ = float(isz) * batch * (m * k + k * n + m * n)
    # the vendor's strided-batched GEMM on the same memory (row-major view of the same buffers)
    av, bv, cv = a.view(batch, k, m), b.view(batch, n, k), c.view(batch, n, m)
    for _ in range(3):
        torch.bmm(bv, av, out=cv)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        torch.bmm(bv, av, out=cv)
    e1.record()
    torch.cuda.synchronize()
    us_v = e0.elapsed_time(e1) / 20 * 1e3
    print(f"strided {name} {m}x{n}x{k} x{batch}: {us:8.1f} us  {bytes_ / us / 1e6:6.2f} TB/s  {2.0 * m * n * k * batch / us / 1e6:7.2f} TFLOP/s  "
          f"kernel={lib.sbamd_last_kernel(h)} | vendor {us_v:8.1f} us  ours/vendor {us / us_v:5.2f}", flush=True)
lib.sbamd_destroy(h)
