"""Time the 16-bit K cut over depths and both slab kernels (ring 128 x 128 / register-staged 64 x 64) on thin shapes
through the raw C ABI.  Usage: python tools/sweep16_split.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sycl_blas_amd import _lib  # noqa: E402

SHAPES = [(64, 3, 401408), (128, 128, 100352), (128, 64, 100352), (256, 128, 25088), (512, 512, 6272), (128, 576, 401408),
          (32, 32, 1048576), (1024, 1024, 16384), (64, 4096, 4096)]


def main():
    lib = _lib.load()
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    h = C.c_void_p()
    assert lib.sbamd_create(C.byref(h), 0, C.c_void_p(side.cuda_stream)) == 0
    for (m, n, k) in SHAPES:
        a = (torch.rand(m * k, device="cuda") * 7 - 2).to(torch.bfloat16)
        b = (torch.rand(k * n, device="cuda") * 7 - 2).to(torch.bfloat16)
        c = torch.empty(m * n, device="cuda", dtype=torch.float32)
        line = f"bf16 {m:5d} x {n:5d} x {k:7d}:"
        cands = [("auto", _lib.KERNEL_AUTO, 0), ("large", _lib.KERNEL_LARGE, 0), ("small", _lib.KERNEL_SMALL, 0)]
        cands += [(f"ring/{d}", _lib.KERNEL_SPLITK, d) for d in (8, 32, 64, 128, 256, 512)]
        cands += [(f"staged/{d}", _lib.KERNEL_SPLITK, d | (1 << 16)) for d in (32, 128, 256)]
        for label, fam, param in cands:
            lib.sbamd_set_kernel(h, fam, param)

            def call():
                st = lib.sbamd_bsgemm(h, b"n", b"n", m, n, k, C.c_float(1.0), C.c_void_p(a.data_ptr()), m,
                                      C.c_void_p(b.data_ptr()), k, C.c_float(0.0), C.c_void_p(c.data_ptr()), m)
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
            d = C.c_int64()
            kc = C.c_int64()
            lib.sbamd_last_splitk(h, C.byref(d), C.byref(kc))
            line += f"  {label} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} (d{d.value})"
        print(line, flush=True)
    lib.sbamd_set_kernel(h, _lib.KERNEL_AUTO, 0)
    lib.sbamd_destroy(h)


if __name__ == "__main__":
    main()
