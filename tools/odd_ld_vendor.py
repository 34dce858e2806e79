"""Yardstick: the vendor GEMM behind torch.matmul on odd sizes (row-major M x K times K x N, every leading dimension odd)."""
import sys

import torch

torch.backends.cuda.matmul.allow_tf32 = False
for name, dt in (("f32", torch.float32), ("f64", torch.float64), ("bf16", torch.bfloat16)):
    for s in (4096, 4099):
        a = (torch.rand(s, s, device="cuda") - 0.5).to(dt)
        b = (torch.rand(s, s, device="cuda") - 0.5).to(dt)
        c = torch.empty(s, s, device="cuda", dtype=dt)
        for _ in range(3):
            torch.matmul(a, b, out=c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            torch.matmul(a, b, out=c)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"vendor {name} {s}^3: {ms * 1e3:9.1f} us {2.0 * s ** 3 / ms / 1e9:8.1f} TFLOP/s", flush=True)
