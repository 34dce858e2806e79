"""Yardstick only (not part of the library or its tests): what the vendor GEMM behind torch.matmul
reaches on this GPU for the bench shapes, with the bench's data range, to tell kernel headroom from
power / clock limits.  Usage: python tools/yardstick_hipblaslt.py"""
import torch


def time_ms(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    n = 8192
    for dt, name in ((torch.bfloat16, "bf16"), (torch.float16, "f16"), (torch.float32, "f32"), (torch.float64, "f64")):
        a = (torch.rand(n, n, device="cuda", dtype=torch.float32) * 7 - 2).to(dt)
        b = (torch.rand(n, n, device="cuda", dtype=torch.float32) * 7 - 2).to(dt)
        for label, fn in (("nn", lambda: a @ b), ("nt", lambda: a @ b.t()), ("tn", lambda: a.t() @ b)):
            ms = time_ms(fn, 10)
            print(f"vendor {name} {label} {n}^3: {ms:.3f} ms  {2 * n ** 3 / ms / 1e9:.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
