"""Developer probe: the tall-skinny SGEMM 256 x 256 x 4194304 (all four layouts), planner's choice."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sycl_blas_amd as sb  # noqa: E402
from sycl_blas_amd import _lib, helper  # noqa: E402

layouts = sys.argv[1].split(",") if len(sys.argv) > 1 else ["nn", "nt", "tn", "tt"]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
pad = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # extra elements on a K-long leading dimension
m = n = 256
k = 1 << 22
with sb.SB_Handle(0) as h:
    a_g = helper.allocate(h, np.float32, m * (k + pad))
    b_g = helper.allocate(h, np.float32, (k + pad) * n)
    c_g = helper.allocate(h, np.float32, m * n)
    helper.fill_uniform(h, a_g, m * (k + pad), 1)
    helper.fill_uniform(h, b_g, (k + pad) * n, 2)
    depth = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # pin the K cut on the biggest tile
    if depth:
        h.set_kernel(_lib.KERNEL_SPLITK, depth | (3 << 16))
    else:
        h.set_kernel(_lib.KERNEL_AUTO)
    for lay in layouts:
        ta, tb = lay
        lda = k + pad if ta != "n" else m
        ldb = n if tb != "n" else k + pad
        call = lambda: sb._gemm(h, ta, tb, m, n, k, 1.0, a_g, lda, b_g, ldb, 0.0, c_g, m)
        for _ in range(2):
            call()
        h.wait()
        e0 = h.record_event()
        for _ in range(reps):
            call()
        e1 = h.record_event()
        e1.wait()
        ms = e0.elapsed_ms(e1) / reps
        print(f"tall-skinny {lay}: {ms:.3f} ms {2.0 * m * n * k / ms / 1e9:.1f} TFLOP/s split={h.last_splitk()} pad={pad}", flush=True)
