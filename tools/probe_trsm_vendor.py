"""Yardstick: the vendor's triangular solve behind torch.linalg.solve_triangular (left, lower, no transpose) at the
sizes bench.py times _trsm at.  Not part of the library or its tests."""
import torch

for dt in (torch.float32, torch.float64):
    for n in (2048, 4096, 8192):
        a = torch.tril(torch.rand(n, n, device="cuda", dtype=dt)) + n * torch.eye(n, device="cuda", dtype=dt)
        b = torch.rand(n, n, device="cuda", dtype=dt)
        for _ in range(2):
            x = torch.linalg.solve_triangular(a, b, upper=False, left=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            x = torch.linalg.solve_triangular(a, b, upper=False, left=True)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"vendor trsm {dt} {n}x{n}: {ms:8.3f} ms  {float(n) ** 3 / ms / 1e9:7.1f} TFLOP/s", flush=True)
