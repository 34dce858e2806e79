"""Time every 16-bit tile family (pinned) on a list of shapes through the raw C ABI: which family should the
dispatcher take where?  Usage: python tools/sweep16.py [bf16|f16] [MxNxK ...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sycl_blas_amd import _lib  # noqa: E402

SHAPES = [(s, s, s) for s in (1024, 1536, 2048, 3072, 4096, 6144)] + [
    (3000, 3000, 3000), (512, 4096, 4096), (4096, 512, 4096), (2048, 4096, 4096), (8192, 8192, 64), (8192, 8192, 256),
    (1024, 1024, 16384), (768, 768, 768), (2304, 2304, 2304), (2560, 2560, 2560), (5120, 5120, 5120)]


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[name]
    out_dt = torch.float32 if name == "bf16" else dt
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[2:]] or SHAPES
    lib = _lib.load()
    fn = lib.sbamd_bsgemm if name == "bf16" else lib.sbamd_hgemm
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    h = C.c_void_p()
    assert lib.sbamd_create(C.byref(h), 0, C.c_void_p(side.cuda_stream)) == 0
    # (label, family, parameter); HUGE with bit 1 set: eight wavefronts per 256 x 256 tile instead of four
    fams = [("auto", _lib.KERNEL_AUTO, 0), ("huge", _lib.KERNEL_HUGE, 0), ("huge/8 waves", _lib.KERNEL_HUGE, 2),
            ("large", _lib.KERNEL_LARGE, 0), ("small", _lib.KERNEL_SMALL, 0), ("K cut 8", _lib.KERNEL_SPLITK, 8),
            ("K cut 16", _lib.KERNEL_SPLITK, 16)]
    for (m, n, k) in shapes:
        a = (torch.rand(m * k, device="cuda") * 7 - 2).to(dt)
        b = (torch.rand(k * n, device="cuda") * 7 - 2).to(dt)
        c = torch.empty(m * n, device="cuda", dtype=out_dt)
        reps = max(3, min(200, int(4e11 / (2.0 * m * n * k))))
        line = f"{name} {m:5d} x {n:5d} x {k:6d}:"
        for label, fam, param in fams:
            lib.sbamd_set_kernel(h, fam, param)

            def call():
                st = fn(h, b"n", b"n", m, n, k, C.c_float(1.0), C.c_void_p(a.data_ptr()), m, C.c_void_p(b.data_ptr()), k,
                        C.c_float(0.0), C.c_void_p(c.data_ptr()), m)
                assert st == 0, st

            for _ in range(3):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                call()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            ran = lib.sbamd_last_kernel(h)
            line += f"  {label} {us:8.1f} us (k{ran})"
        print(line, flush=True)
    lib.sbamd_set_kernel(h, _lib.KERNEL_AUTO, 0)
    lib.sbamd_destroy(h)


if __name__ == "__main__":
    main()
