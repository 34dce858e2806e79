"""Developer probe: what reading C costs -- the same product with beta = 0 and beta != 0, raw C ABI.
usage: probe_beta.py [f32|f64]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sycl_blas_amd as sb  # noqa: E402
from sycl_blas_amd import _lib, helper  # noqa: E402

dtype = np.float64 if len(sys.argv) > 1 and sys.argv[1] == "f64" else np.float32
lib = _lib.load()
fn = lib.sbamd_sgemm if dtype == np.float32 else lib.sbamd_dgemm
sc = C.c_float if dtype == np.float32 else C.c_double
SHAPES = [(8192, 8192, 8192), (4096, 4096, 4096), (2048, 2048, 2048), (1024, 1024, 1024), (8192, 8192, 256), (8192, 8192, 1024),
          (4096, 4096, 512)]
with sb.SB_Handle(0) as h:
    for m, n, k in SHAPES:
        a_g = helper.allocate(h, dtype, m * k)
        b_g = helper.allocate(h, dtype, k * n)
        c_g = helper.allocate(h, dtype, m * n)
        helper.fill_uniform(h, a_g, m * k, 1)
        helper.fill_uniform(h, b_g, k * n, 2)
        out = []
        for beta in (0.0, 0.5):
            helper.fill_uniform(h, c_g, m * n, 3)
            call = lambda: fn(h._h, b"n", b"n", m, n, k, sc(1.0), a_g.ptr, m, b_g.ptr, k, sc(beta), c_g.ptr, m)
            reps = max(5, min(200, int(4e11 / (2.0 * m * n * k))))
            for _ in range(5):
                call()
            h.wait()
            e0 = h.record_event()
            for _ in range(reps):
                call()
            e1 = h.record_event()
            e1.wait()
            out.append(e0.elapsed_ms(e1) / reps * 1e3)
        print(f"{np.dtype(dtype).name} {m}x{n}x{k}: beta=0 {out[0]:9.1f} us  beta=0.5 {out[1]:9.1f} us  (+{out[1] - out[0]:.1f} us, "
              f"C read at {m * n * np.dtype(dtype).itemsize / max(out[1] - out[0], 1e-3) / 1e6:.2f} TB/s) kernel={h.last_kernel()} cut={h.last_splitk()}", flush=True)
        del a_g, b_g, c_g
