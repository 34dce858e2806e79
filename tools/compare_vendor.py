"""Yardstick sweep (not part of the library or its tests): this library against the vendor GEMM behind
torch.matmul on the same GPU, same memory problem, over a list of shapes -- to find the shapes
where a kernel or the planner leaves the most on the table.
Usage: python tools/compare_vendor.py [f32|f64|bf16|f16] [--csv out.csv]
(bf16: bfloat16 operands, float result on this side -- sbamd_bsgemm -- against the vendor's bfloat16 result;
f16: half operands and result on both sides.)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sycl_blas_amd as sb  # noqa: E402
from sycl_blas_amd import DeviceBuffer, _lib  # noqa: E402

SHAPES = (
    [(s, s, s) for s in (256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144, 8192)]
    + [(1000, 1000, 1000), (3000, 3000, 3000), (4100, 4100, 4100)]
    # rectangular: (M, N, K) of row-major torch (M, K) @ (K, N)
    + [(512, 4096, 4096), (2048, 4096, 4096), (8192, 4096, 1024), (4096, 11008, 4096), (4096, 4096, 11008),
       (128, 8192, 8192), (8192, 128, 8192), (64, 4096, 4096), (4096, 64, 4096), (16384, 1024, 1024),
       (1024, 16384, 1024), (1024, 1024, 16384), (512, 512, 65536), (256, 256, 262144), (2048, 2048, 512),
       (8192, 8192, 256), (8192, 8192, 64), (32, 32, 1048576), (100, 100, 100000), (5000, 64, 64)]
)


def time_ms(fn, sync, reps):
    """Mean time of `reps` back-to-back calls, after 20 ms of the same call: the clock the chip holds depends on
    what it ran in the last milliseconds, so each side is timed in its own steady state."""
    import time
    t0 = time.perf_counter()
    while True:
        for _ in range(4):
            fn()
        sync()
        if time.perf_counter() - t0 > 0.02:
            break
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "f32"
    dt = {"f32": torch.float32, "f64": torch.float64, "bf16": torch.bfloat16, "f16": torch.float16}[name]
    np_name = {"f32": "float32", "f64": "float64", "bf16": "bfloat16", "f16": "float16"}[name]
    out_dt = torch.float32 if name == "bf16" else dt
    out_np = "float32" if name == "bf16" else np_name
    csv = open(sys.argv[sys.argv.index("--csv") + 1], "w") if "--csv" in sys.argv else None
    torch.backends.cuda.matmul.allow_tf32 = False
    side = torch.cuda.Stream()  # one stream for both libraries, so the torch events see this library's launches
    rows = []
    lib = _lib.load()
    with torch.cuda.stream(side), sb.SB_Handle(0, stream=side.cuda_stream) as h:
        for (M, N, K) in SHAPES:
            a = (torch.rand(M, K, device="cuda", dtype=torch.float32) * 7 - 2).to(dt)
            b = (torch.rand(K, N, device="cuda", dtype=torch.float32) * 7 - 2).to(dt)
            c = torch.empty(M, N, device="cuda", dtype=dt)
            c_ours = c if out_dt == dt else torch.empty(M, N, device="cuda", dtype=out_dt)
            reps = max(3, min(200, int(2e11 / (2.0 * M * N * K))))
            vendor = lambda: torch.matmul(a, b, out=c)
            ms_v = time_ms(vendor, torch.cuda.synchronize, reps)
            # the same memory problem in column-major terms: C^T (N x M) = B^T (N x K) * A^T (K x M)
            a_o = DeviceBuffer.from_pointer(b.data_ptr(), np_name, K * N, b)
            b_o = DeviceBuffer.from_pointer(a.data_ptr(), np_name, M * K, a)
            c_o = DeviceBuffer.from_pointer(c_ours.data_ptr(), out_np, M * N, c_ours)
            # straight through the C ABI: the Python mirror records an event per call, which at 20 us per
            # launch makes the host the bottleneck and the comparison a comparison of launch paths
            fn = {"f32": lib.sbamd_sgemm, "f64": lib.sbamd_dgemm, "bf16": lib.sbamd_bsgemm, "f16": lib.sbamd_hgemm}[name]
            sc = C.c_double if name == "f64" else C.c_float
            args = (h._h, b"n", b"n", N, M, K, sc(1.0), a_o.ptr, N, b_o.ptr, K, sc(0.0), c_o.ptr, N)
            ms_o = time_ms(lambda: fn(*args), torch.cuda.synchronize, reps)
            # second round in the same order; each side keeps its better mean
            ms_v = min(ms_v, time_ms(vendor, torch.cuda.synchronize, reps))
            ms_o = min(ms_o, time_ms(lambda: fn(*args), torch.cuda.synchronize, reps))
            tf = lambda ms: 2.0 * M * N * K / ms / 1e9
            depth = h.last_splitk()[0]
            rows.append((ms_o / ms_v, M, N, K, ms_o, ms_v, h.last_kernel(), depth))
            line = (f"{name} {M:6d} x {N:6d} x {K:7d}: ours {ms_o * 1e3:9.1f} us {tf(ms_o):7.1f} TF | vendor {ms_v * 1e3:9.1f} us "
                    f"{tf(ms_v):7.1f} TF | ours/vendor time {ms_o / ms_v:5.2f}  kernel={h.last_kernel()} depth={depth}")
            print(line, flush=True)
            if csv:
                csv.write(f"{name},{M},{N},{K},{ms_o * 1e3:.2f},{ms_v * 1e3:.2f},{h.last_kernel()},{depth}\n")
            del a, b, c, c_ours
    rows.sort(reverse=True)
    print("worst five (time ratio ours / vendor):")
    for r in rows[:5]:
        print(f"   {r[0]:5.2f}  {r[1]} x {r[2]} x {r[3]}  kernel={r[6]} depth={r[7]}")


if __name__ == "__main__":
    main()
