"""Developer probe: one GEMM shape, one kernel family, many launches (for counter passes).
usage: probe_one.py M N K [family=0] [param=0] [reps=200] [dtype=f32] [trans=nn]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sycl_blas_amd as sb  # noqa: E402
from sycl_blas_amd import helper  # noqa: E402

m, n, k = (int(x) for x in sys.argv[1:4])
fam = int(sys.argv[4]) if len(sys.argv) > 4 else 0
par = int(sys.argv[5]) if len(sys.argv) > 5 else 0
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 200
dtype = np.float64 if len(sys.argv) > 7 and sys.argv[7] == "f64" else np.float32
ta, tb = sys.argv[8] if len(sys.argv) > 8 else "nn"
with sb.SB_Handle(0) as h:
    a_g = helper.allocate(h, dtype, m * k)
    b_g = helper.allocate(h, dtype, k * n)
    c_g = helper.allocate(h, dtype, m * n)
    helper.fill_uniform(h, a_g, m * k, 1)
    helper.fill_uniform(h, b_g, k * n, 2)
    h.set_kernel(fam, par)
    lda = k if ta != "n" else m
    ldb = n if tb != "n" else k
    call = lambda: sb._gemm(h, ta, tb, m, n, k, 1.0, a_g, lda, b_g, ldb, 0.0, c_g, m)
    for _ in range(10):
        call()
    h.wait()
    e0 = h.record_event()
    for _ in range(reps):
        call()
    e1 = h.record_event()
    e1.wait()
    us = e0.elapsed_ms(e1) / reps * 1e3
    print(f"{np.dtype(dtype).name} {ta}{tb} {m}x{n}x{k} family {fam} param {par}: {us:.1f} us {2.0 * m * n * k / us / 1e6:.1f} TFLOP/s "
          f"kernel={h.last_kernel()} cut={h.last_splitk()}")
