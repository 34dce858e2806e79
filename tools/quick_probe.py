"""Developer probe: a few parity cases with verbose mismatch output, then a
quick throughput reading per kernel family.  Not part of the test suite."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import oracle  # noqa: E402
import sycl_blas_amd as sb  # noqa: E402
from sycl_blas_amd import _lib, helper  # noqa: E402


def run_case(h, dtype, m, n, k, ta, tb, alpha, beta, fam, param=0):
    rng = np.random.default_rng(1)
    lda = (k if ta != "n" else m)
    ldb = (n if tb != "n" else k)
    a = rng.uniform(-2, 5, m * k).astype(dtype)
    b = rng.uniform(-2, 5, k * n).astype(dtype)
    c = rng.uniform(-2, 5, m * n).astype(dtype)
    want = c.copy()
    oracle.gemm_backend(ta, tb, m, n, k, alpha, a, lda, 0, b, ldb, 0, beta, want, m, 0)
    h.set_kernel(fam, param)
    a_g, b_g, c_g = (helper.allocate(h, dtype, x.size) for x in (a, b, c))
    for host, dev in ((a, a_g), (b, b_g), (c, c_g)):
        helper.copy_to_device(h, host, dev)
    sb._gemm(h, ta, tb, m, n, k, alpha, a_g, lda, b_g, ldb, beta, c_g, m)
    got = np.empty_like(c)
    h.wait(helper.copy_to_host(h, c_g, got))
    exact = np.array_equal(got.view(np.uint8), want.view(np.uint8))
    err = np.abs(got.astype(np.float64) - want.astype(np.float64)).max()
    nbad = int((got != want).sum())
    print(f"{np.dtype(dtype).name} fam={fam} {m}x{n}x{k} {ta}{tb} a={alpha} b={beta}: exact={exact} "
          f"maxerr={err:.3e} nbad={nbad}/{got.size} kernel={h.last_kernel()}", flush=True)
    if not exact and nbad:
        bad = np.flatnonzero(got != want)[:8]
        print("   first bad idx (row,col):", [(int(i % m), int(i // m)) for i in bad])
        print("   got ", got[bad], "\n   want", want[bad])
    return exact


def bench(h, dtype, m, n, k, ta, tb, fam, reps=5, param=0):
    a_g = helper.allocate(h, dtype, m * k)
    b_g = helper.allocate(h, dtype, k * n)
    c_g = helper.allocate(h, dtype, m * n)
    helper.fill_uniform(h, a_g, m * k, 1)
    helper.fill_uniform(h, b_g, k * n, 2)
    lda = (k if ta != "n" else m)
    ldb = (n if tb != "n" else k)
    h.set_kernel(fam, param)
    for _ in range(2):
        sb._gemm(h, ta, tb, m, n, k, 1.0, a_g, lda, b_g, ldb, 0.0, c_g, m)
    h.wait()
    e0 = h.record_event()
    for _ in range(reps):
        sb._gemm(h, ta, tb, m, n, k, 1.0, a_g, lda, b_g, ldb, 0.0, c_g, m)
    e1 = h.record_event()
    e1.wait()
    ms = e0.elapsed_ms(e1) / reps
    print(f"BENCH {np.dtype(dtype).name} fam={fam} {m}x{n}x{k} {ta}{tb}: {ms:.3f} ms  "
          f"{2.0 * m * n * k / ms / 1e9:.1f} TFLOP/s", flush=True)
    for buf in (a_g, b_g, c_g):
        helper.deallocate(buf)


def bench_batched(h, dtype, m, n, k, batch, reps=20):
    sa, sb_, sc = m * k, k * n, m * n
    a_g = helper.allocate(h, dtype, sa * batch)
    b_g = helper.allocate(h, dtype, sb_ * batch)
    c_g = helper.allocate(h, dtype, sc * batch)
    helper.fill_uniform(h, a_g, sa * batch, 1)
    helper.fill_uniform(h, b_g, sb_ * batch, 2)
    h.set_kernel(_lib.KERNEL_AUTO)
    run = lambda: sb._gemm_strided_batched(h, "n", "n", m, n, k, 1.0, a_g, m, sa, b_g, k, sb_, 0.0, c_g, m, sc, batch)
    for _ in range(3):
        run()
    h.wait()
    e0 = h.record_event()
    for _ in range(reps):
        run()
    e1 = h.record_event()
    e1.wait()
    ms = e0.elapsed_ms(e1) / reps
    nbytes = (sa + sb_ + sc) * batch * np.dtype(dtype).itemsize
    print(f"BENCH batched {np.dtype(dtype).name} {batch}x{m}x{n}x{k}: {ms * 1e3:.1f} us  "
          f"{2.0 * m * n * k * batch / ms / 1e9:.1f} TFLOP/s  {nbytes / ms / 1e6:.0f} GB/s  kernel={h.last_kernel()}", flush=True)
    for buf in (a_g, b_g, c_g):
        helper.deallocate(buf)


def main():
    h = sb.SB_Handle(0)
    print("CUs", h.get_num_compute_units(), flush=True)
    ok = True
    for dtype in (np.float32, np.float64):
        for fam in (_lib.KERNEL_NAIVE, _lib.KERNEL_SMALL, _lib.KERNEL_LARGE, _lib.KERNEL_HUGE):
            for ta in "nt":
                for tb in "nt":
                    ok &= run_case(h, dtype, 96, 80, 40, ta, tb, 1.5, 0.5, fam)
            ok &= run_case(h, dtype, 129, 67, 37, "n", "n", 1.5, 0.0, fam)
            ok &= run_case(h, dtype, 256, 256, 256, "n", "t", 1.0, 0.0, fam)
    print("ALL EXACT" if ok else "SOME MISMATCH", flush=True)
    if "--bench" in sys.argv:
        for dtype in (np.float32, np.float64):
            for fam in (_lib.KERNEL_LARGE, _lib.KERNEL_HUGE):
                for ta, tb in (("n", "n"), ("n", "t"), ("t", "n"), ("t", "t")):
                    bench(h, dtype, 8192, 8192, 8192, ta, tb, fam, reps=3)
            for n in (1024, 2048, 4096):
                bench(h, dtype, n, n, n, "n", "n", _lib.KERNEL_AUTO, reps=10)
        bench_batched(h, np.float32, 64, 64, 64, 4096)
        bench_batched(h, np.float64, 64, 64, 64, 4096)
        for kk in (65536, 1 << 20, 1 << 22):
            bench(h, np.float32, 256, 256, kk, "n", "n", _lib.KERNEL_AUTO, reps=3)
            print("   splitk cut", h.last_splitk(), flush=True)
        for depth in (64, 128, 256, 384, 512, 1024):
            bench(h, np.float32, 256, 256, 1 << 22, "n", "n", _lib.KERNEL_SPLITK, reps=3, param=depth)
            print("   splitk cut", h.last_splitk(), flush=True)
        for name in ("float16", "bfloat16"):
            for n_ in (4096, 8192):
                a_g = helper.allocate(h, name, n_ * n_)
                b_g = helper.allocate(h, name, n_ * n_)
                c_g = helper.allocate(h, "float32", n_ * n_)
                helper.fill_uniform(h, a_g, n_ * n_, 1)
                helper.fill_uniform(h, b_g, n_ * n_, 2)
                for ta, tb in (("n", "n"), ("n", "t"), ("t", "n")):
                    run = lambda: sb._gemm(h, ta, tb, n_, n_, n_, 1.0, a_g, n_, b_g, n_, 0.0, c_g, n_)
                    run(); h.wait()
                    e0 = h.record_event()
                    for _ in range(3):
                        run()
                    e1 = h.record_event(); e1.wait()
                    ms = e0.elapsed_ms(e1) / 3
                    print(f"BENCH {name}->f32 {n_}^3 {ta}{tb}: {ms:.3f} ms {2.0 * n_ ** 3 / ms / 1e9:.1f} TFLOP/s", flush=True)
                for buf in (a_g, b_g, c_g):
                    helper.deallocate(buf)
    h.close()


if __name__ == "__main__":
    main()
