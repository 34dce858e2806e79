"""Yardstick aid (not part of the library or its tests): run the vendor GEMM behind torch.matmul on a few shapes so a
kernel trace (rocprofv3 --kernel-trace --stats -- python3 tools/vendor_kernel_names.py) shows which of its kernels,
with which tile, serve the shapes this library is compared on.  Usage: python tools/vendor_kernel_names.py [f32|f64]"""
import sys

import torch

SHAPES = [(8192, 8192, 8192), (1024, 1024, 1024), (512, 4096, 4096), (128, 8192, 8192), (1024, 1024, 16384), (512, 512, 512)]
dt = torch.float64 if len(sys.argv) > 1 and sys.argv[1] == "f64" else torch.float32
torch.backends.cuda.matmul.allow_tf32 = False
for (M, N, K) in SHAPES:
    a = torch.rand(M, K, device="cuda", dtype=dt)
    b = torch.rand(K, N, device="cuda", dtype=dt)
    c = torch.empty(M, N, device="cuda", dtype=dt)
    for _ in range(5):
        torch.matmul(a, b, out=c)
    torch.cuda.synchronize()
    print(M, N, K, flush=True)
