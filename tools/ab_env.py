"""A/B harness: time C = A*B through the raw C ABI under the current environment and print a digest of C.

Run it twice, with and without a switch in the environment, and compare the lines: equal digests mean bitwise
equal results.  Usage: ab_env.py DTYPE M N K [TA TB] [REPS]
"""
import ctypes as C
import hashlib
import sys

import torch

from sycl_blas_amd import _lib


def main():
    dt, m, n, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    ta, tb = (sys.argv[5], sys.argv[6]) if len(sys.argv) > 6 else ("n", "n")
    reps = int(sys.argv[7]) if len(sys.argv) > 7 else 20
    lib = _lib.load()
    tdt = {"f32": torch.float32, "f64": torch.float64}[dt]
    fn = {"f32": lib.sbamd_sgemm, "f64": lib.sbamd_dgemm}[dt]
    sc = {"f32": C.c_float, "f64": C.c_double}[dt]
    g = torch.Generator(device="cuda").manual_seed(7)
    a = torch.rand(m * k, dtype=tdt, device="cuda", generator=g) - 0.5
    b = torch.rand(k * n, dtype=tdt, device="cuda", generator=g) - 0.5
    c = torch.zeros(m * n, dtype=tdt, device="cuda")
    side = torch.cuda.Stream()  # an explicit stream: the torch events below must see this library's launches
    torch.cuda.set_stream(side)
    stream = side.cuda_stream
    h = C.c_void_p()
    assert lib.sbamd_create(C.byref(h), 0, C.c_void_p(stream)) == 0
    lda = m if ta == "n" else k
    ldb = k if tb == "n" else n

    def call():
        st = fn(h, ta.encode(), tb.encode(), m, n, k, sc(1.0), C.c_void_p(a.data_ptr()), lda,
                C.c_void_p(b.data_ptr()), ldb, sc(0.0), C.c_void_p(c.data_ptr()), m)
        assert st == 0, st

    for _ in range(3):
        call()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    digest = hashlib.sha256(c.cpu().numpy().tobytes()).hexdigest()[:16]
    print(f"{dt} {ta}{tb} {m}x{n}x{k}: {best * 1e3:9.1f} us  {2.0 * m * n * k / best / 1e9:7.1f} TFLOP/s  digest {digest}",
          flush=True)
    lib.sbamd_destroy(h)


if __name__ == "__main__":
    main()
