#!/usr/bin/env python3
"""What one rank of the strong-scaling schedule multiplies, timed on ONE GPU: for 2, 4 and 8 ranks the band
GEMMs of rank 0 (sbamd_dist_band_widths over its n / ranks columns, or --bands equal bands) back to back through the
raw C ABI, against the share of the whole product's time an ideal split would take.  A multi-GPU projection
adds the link time to these.   usage: python tools/probe_bands.py [f32|f64] [size] [bands]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sycl_blas_amd as sb  # noqa: E402
from sycl_blas_amd import _lib, helper  # noqa: E402
from sycl_blas_amd.distributed import band_widths, split_range  # noqa: E402

dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
bands = int(sys.argv[3]) if len(sys.argv) > 3 else 0
name, sc, fn = ("float32", C.c_float, "sbamd_sgemm") if dt == "f32" else ("float64", C.c_double, "sbamd_dgemm")
lib = _lib.load()
gemm = getattr(lib, fn)
with sb.SB_Handle(0) as h:
    a, b, c = (helper.allocate(h, name, size * size) for _ in range(3))
    helper.fill_uniform(h, a, size * size, 1)
    helper.fill_uniform(h, b, size * size, 2)

    def timed(calls, reps=10, warm=4):
        for _ in range(warm):
            for call in calls:
                call()
        h.wait()
        e0 = h.record_event()
        for _ in range(reps):
            for call in calls:
                call()
        e1 = h.record_event()
        e1.wait()
        return e0.elapsed_ms(e1) / reps * 1e3

    def call_for(c0, c1):
        return lambda: gemm(h._h, b"n", b"n", size, c1 - c0, size, sc(1.0), a.ptr, size, b.ptr + c0 * size * a.itemsize, size, sc(0.0),
                            c.ptr + c0 * size * a.itemsize, size)

    whole = timed([call_for(0, size)], reps=5, warm=3)
    print(f"{dt} {size}^3 on one GPU: {whole:.1f} us = {2 * size ** 3 / whole / 1e6:.1f} TFLOP/s")
    for world in (2, 4, 8):
        c0, c1 = split_range(size, world, 0)
        widths = band_widths(c1 - c0, bands, world)
        calls, col, each = [], c0, []
        for w in widths:
            calls.append(call_for(col, col + w))
            us = timed([calls[-1]])
            each.append((w, round(us, 1), h.last_kernel(), h.last_splitk()[0]))
            col += w
        together = timed(calls)
        one_piece = timed([call_for(c0, c1)])
        ideal = whole / world
        print(f"  {world} ranks: bands {widths}: {together:.1f} us back to back (one piece {one_piece:.1f}, ideal {ideal:.1f}) -> "
              f"{100 * ideal / together:.1f} % ; per band (cols, us, family, K slabs): {each}")
