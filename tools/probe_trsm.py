"""Developer probe: _trsm 8192^2 (left / lower / no transpose / non-unit and right / upper / transposed), fp32 and fp64.
usage: probe_trsm.py [size=8192]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sycl_blas_amd as sb  # noqa: E402
from sycl_blas_amd import DeviceBuffer  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
s = torch.cuda.Stream()
with torch.cuda.stream(s), sb.SB_Handle(0, stream=s.cuda_stream) as h:
    for tdt, name in ((torch.float32, "float32"), (torch.float64, "float64")):
        ta_t = torch.rand(size, size, dtype=tdt, device="cuda").mul_(1.0 / size)
        ta_t.diagonal().fill_(2.0)
        tb_t = torch.rand(size * size, dtype=tdt, device="cuda")
        a = DeviceBuffer.from_pointer(ta_t.data_ptr(), name, size * size, ta_t)
        b = DeviceBuffer.from_pointer(tb_t.data_ptr(), name, size * size, tb_t)
        torch.cuda.synchronize()
        for side, uplo, trans in (("l", "l", "n"), ("r", "u", "t"), ("l", "u", "t")):
            call = lambda: sb._trsm(h, side, uplo, trans, "n", size, size, 1.0, a, size, b, size)
            call()
            h.wait()
            e0 = h.record_event()
            for _ in range(3):
                call()
            e1 = h.record_event()
            e1.wait()
            ms = e0.elapsed_ms(e1) / 3
            print(f"{name} trsm {side}{uplo}{trans}n {size}^2: {ms:.3f} ms  {float(size) ** 3 / ms / 1e9:.1f} TFLOP/s launches={h.last_trsm_launches()}",
                  flush=True)
        del a, b, ta_t, tb_t
