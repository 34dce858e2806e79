"""Developer probe: what the chip reaches on pure writes, pure reads and a copy (torch kernels), for scale."""
import torch

n = 512 * 1024 * 1024 // 4
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


us = timed(lambda: x.fill_(1.0))
print(f"fill 512 MiB: {us:.1f} us  {n * 4 / us / 1e6:.2f} TB/s written")
us = timed(lambda: x.zero_())
print(f"zero 512 MiB (memset): {us:.1f} us  {n * 4 / us / 1e6:.2f} TB/s written")
us = timed(lambda: y.copy_(x))
print(f"copy 512 MiB: {us:.1f} us  {2 * n * 4 / us / 1e6:.2f} TB/s read + written")
us = timed(lambda: x.sum())
print(f"sum 512 MiB: {us:.1f} us  {n * 4 / us / 1e6:.2f} TB/s read")
us = timed(lambda: torch.add(x, 1.0, out=x))
print(f"in-place add 512 MiB: {us:.1f} us  {2 * n * 4 / us / 1e6:.2f} TB/s read + written")
