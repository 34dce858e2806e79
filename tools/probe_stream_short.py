"""Developer probe: the 64x64x64 batches only (fp32 and fp64, NN and TN), planner's choice."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sycl_blas_amd as sb  # noqa: E402
from sycl_blas_amd import _lib  # noqa: E402
from probe_stream import run  # noqa: E402

with sb.SB_Handle(0) as h:
    for dtype in (np.float32, np.float64):
        for batch in (4096, 16384):
            for ta, tb in (("n", "n"), ("t", "n")):
                for beta in (0.0, 0.5):
                    us = min(run(h, dtype, 64, 64, 64, batch, ta, tb, beta, _lib.KERNEL_AUTO, 20)[0] for _ in range(3))
                    nbytes = 64 * 64 * (3 if beta == 0 else 4) * batch * np.dtype(dtype).itemsize
                    print(f"{np.dtype(dtype).name} x{batch} {ta}{tb} beta={beta}: {us:7.1f} us {nbytes / us / 1e6:5.2f} TB/s", flush=True)
