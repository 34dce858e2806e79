#!/usr/bin/env python3
"""Headline benchmark: SGEMM / DGEMM at 8192^3 on MI355X through the C ABI.

    python bench.py --gpus N --steps K --warmup W

One "step" is one GEMM call (C <- alpha*A*B + beta*C, alpha = 1, beta = 0, the
reference benchmark's convention: benchmark/config_csv/blas3/gemm/gemm_square.csv)
on operands that are already resident in HBM.  Inputs are uniform in [-2, 5]
(test/blas_test.hpp:137-140), generated on the device from a counter-based
generator with fixed seeds.

N = 1: the workload is BASELINE config #2, SGEMM 8192 x 8192 x 8192 NN.
N > 1: one process per GPU (launched by torch.distributed.run), operands resident on
every GPU.
  --mode strong (default) the job is ONE 8192^3 GEMM whose full column-major C ends up
                on every GPU, inside the timed region, through the native multi-GPU
                layer (include/sbamd_dist.h, sbamd_dist_gathered_{s,d}gemm): the columns
                of C are dealt to the ranks, each rank multiplies its columns band by
                band straight into their place, and a finished band goes directly into
                the memory of every peer over the xGMI mesh (a copy kernel on a second
                stream, ordered by flag words) while the next band is multiplied.
                --transport rccl uses in-place ncclAllGather per band instead, --transport
                torch the same schedule over torch.distributed.  --bands overrides the
                number of bands, --no-gather leaves C distributed.
  --mode weak   every rank multiplies a full 8192^3 block of a (Pr*8192) x (Pc*8192) x 8192
                job (2-D C blocks, no exchange at all); also reported as an extra of
                the strong run, never as its `value`.
Rank 0 prints ONE JSON line.  `value` is whole-job TFLOP/s = 2*M*N*K of the job /
max-over-ranks wall time of the K timed steps.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Dense MFMA peaks of one MI355X (TFLOP/s) and its HBM3E peak (GB/s).
PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6, "bf16": 2500.0}  # dense MFMA peaks, TFLOP/s
PEAK_HBM_GBS = 8000.0
NP_DTYPE = {"f32": np.float32, "f64": np.float64}
GRIDS = {1: (1, 1), 2: (1, 2), 4: (2, 2), 8: (2, 4)}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)  # the reference warms up 10 calls (common_utils.hpp:1840-1845)
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--trans", choices=["nn", "nt", "tn", "tt"], default="nn")
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--mode", choices=["weak", "strong"], default="strong")
    ap.add_argument("--no-gather", action="store_true", help="strong mode: leave C distributed (no collective)")
    ap.add_argument("--gather", action="store_true", help="accepted for compatibility: gathering is the default")
    ap.add_argument("--bands", type=int, default=0, help="column bands of the overlapped gather (0 = choose)")
    ap.add_argument("--transport", choices=["auto", "mesh", "rccl", "torch"], default="auto",
                    help="how a finished band reaches the peers: direct stores over the xGMI mesh (auto: tried first), "
                         "RCCL all-gathers from the native layer, or torch.distributed collectives")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed side measurements")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--one-rank-rccl", action="store_true",
                    help="one process, one GPU, but through the N>1 code path of the native layer on a one-rank RCCL "
                         "communicator: the in-place band gathers are issued from the second stream behind events exactly "
                         "as the multi-GPU run issues them (not a measurement)")
    ap.add_argument("--rehearse", action="store_true",
                    help="rank logic check on a box with fewer GPUs than ranks: every rank uses GPU 0 (the mesh transport "
                         "then maps the peers' buffers of the same GPU through HIP IPC), the barrier / max-over-ranks go "
                         "through gloo; the number it prints is not a multi-GPU measurement")
    return ap.parse_args()


class Operands:
    """A, B, C of one rank's block product, resident in HBM (torch owns the memory)."""

    def __init__(self, torch, sb, handle, dtype, ta, tb, m, n, k, seed):
        from sycl_blas_amd import DeviceBuffer, helper

        tdt = torch.float32 if dtype == "f32" else torch.float64
        name = "float32" if dtype == "f32" else "float64"
        self.m, self.n, self.k, self.ta, self.tb = m, n, k, ta, tb
        self.lda = k if ta != "n" else m
        self.ldb = n if tb != "n" else k
        self.ldc = m
        self.ta_t = torch.empty(m * k, dtype=tdt, device="cuda")
        self.tb_t = torch.empty(k * n, dtype=tdt, device="cuda")
        self.tc_t = torch.zeros(m * n, dtype=tdt, device="cuda")
        self.a = DeviceBuffer.from_pointer(self.ta_t.data_ptr(), name, m * k, self.ta_t)
        self.b = DeviceBuffer.from_pointer(self.tb_t.data_ptr(), name, k * n, self.tb_t)
        self.c = DeviceBuffer.from_pointer(self.tc_t.data_ptr(), name, m * n, self.tc_t)
        helper.fill_uniform(handle, self.a, m * k, seed)
        helper.fill_uniform(handle, self.b, k * n, seed + 1)
        handle.wait()

    def run(self, sb, handle):
        return sb._gemm(handle, self.ta, self.tb, self.m, self.n, self.k, 1.0, self.a, self.lda, self.b, self.ldb,
                        0.0, self.c, self.ldc)


def device_time_ms(sb, handle, fn, reps, warm=2):
    """Average device time of `fn` from HIP events recorded on the stream the kernels run on."""
    for _ in range(warm):
        fn()
    handle.wait()
    e0 = handle.record_event()
    for _ in range(reps):
        fn()
    e1 = handle.record_event()
    e1.wait()
    return e0.elapsed_ms(e1) / reps


def side_measurements(torch, sb, handle, size):
    """Untimed-region companions of the headline number (device time from HIP events):
    the other 8192^3 variants of config #2, config #3 (strided-batched 64^3 x 4096, HBM-bound),
    config #4 (256 x 256 x 4194304 split-K) and the bf16-input variant of config #5's kernel."""
    from sycl_blas_amd import helper

    extras = {}
    for dt, tr in (("f32", "nt"), ("f64", "nn"), ("f64", "nt")):
        o = Operands(torch, sb, handle, dt, tr[0], tr[1], size, size, size, 777)
        ms = device_time_ms(sb, handle, lambda: o.run(sb, handle), 5)
        tf = 2.0 * size ** 3 / ms / 1e9
        extras[f"{dt}_{tr}_{size}"] = {"ms": round(ms, 4), "tflops": round(tf, 2),
                                       "pct_mfma_peak": round(100 * tf / PEAK_TFLOPS[dt], 2)}
        del o

    # config #1: the sample's call, SGEMM 1024^3 NN (the reference's CPU-runnable case)
    # (launched straight through the C ABI: the Python mirror of blas::_gemm records an event per call, and
    # at 20-odd microseconds per launch that makes the host the bottleneck)
    import ctypes
    from sycl_blas_amd import _lib
    o = Operands(torch, sb, handle, "f32", "n", "n", 1024, 1024, 1024, 778)
    raw = _lib.load().sbamd_sgemm
    raw_args = (handle._h, b"n", b"n", 1024, 1024, 1024, ctypes.c_float(1.0), o.a.ptr, 1024, o.b.ptr, 1024,
                ctypes.c_float(0.0), o.c.ptr, 1024)
    ms = device_time_ms(sb, handle, lambda: raw(*raw_args), 200, warm=20)
    extras["sgemm_1024_nn"] = {"us": round(ms * 1e3, 2), "tflops": round(2.0 * 1024 ** 3 / ms / 1e9, 2),
                               "split_k": list(handle.last_splitk())}
    del o

    # config #3: batch 4096 of 64x64x64, strides 4096 elements, beta = 0
    bm, batch = 64, 4096
    per = bm * bm
    a = helper.allocate(handle, "float32", per * batch)
    b = helper.allocate(handle, "float32", per * batch)
    c = helper.allocate(handle, "float32", per * batch)
    helper.fill_uniform(handle, a, per * batch, 31)
    helper.fill_uniform(handle, b, per * batch, 32)
    ms = device_time_ms(sb, handle, lambda: sb._gemm_strided_batched(handle, "n", "n", bm, bm, bm, 1.0, a, bm, per, b, bm,
                                                                      per, 0.0, c, bm, per, batch), 20, warm=5)
    nbytes = 3.0 * per * batch * 4  # (MK + KN + MN) * sizeof * batch
    extras["sgemm_strided_batched_4096x64^3"] = {
        "us": round(ms * 1e3, 2), "tflops": round(2.0 * bm ** 3 * batch / ms / 1e9, 2),
        "algorithmic_GBps": round(nbytes / ms / 1e6, 1), "pct_hbm_peak": round(100 * nbytes / ms / 1e6 / PEAK_HBM_GBS, 2),
        "bound": "hbm"}
    for buf in (a, b, c):
        helper.deallocate(buf)

    # config #4: 256 x 256 x 4194304, NN (A 4 GiB, B 4 GiB)
    m = n = 256
    k = 1 << 22
    a = helper.allocate(handle, "float32", m * k)
    b = helper.allocate(handle, "float32", k * n)
    c = helper.allocate(handle, "float32", m * n)
    helper.fill_uniform(handle, a, m * k, 41)
    helper.fill_uniform(handle, b, k * n, 42)
    # the reference's protocol: 10 warm-up calls (common_utils.hpp:1840-1845); the clock is still
    # rising over the first few 4 ms launches after an idle spell
    ms = device_time_ms(sb, handle, lambda: sb._gemm(handle, "n", "n", m, n, k, 1.0, a, m, b, k, 0.0, c, m), 10, warm=10)
    depth, k_chunk = handle.last_splitk()
    tf = 2.0 * m * n * k / ms / 1e9
    extras["sgemm_tall_skinny_256x256x4194304"] = {
        "ms": round(ms, 3), "tflops": round(tf, 2), "pct_mfma_peak": round(100 * tf / PEAK_TFLOPS["f32"], 2),
        "algorithmic_GBps": round((m * k + k * n + m * n) * 4 / ms / 1e6, 1), "split_k": [depth, k_chunk]}
    for buf in (a, b, c):
        helper.deallocate(buf)

    # bf16 inputs, fp32 output (kernel of config #5's bf16 variant), 8192^3 TN and NN
    a = helper.allocate(handle, "bfloat16", size * size)
    b = helper.allocate(handle, "bfloat16", size * size)
    c = helper.allocate(handle, "float32", size * size)
    helper.fill_uniform(handle, a, size * size, 51)
    helper.fill_uniform(handle, b, size * size, 52)
    for tr in ("tn", "nn"):
        ms = device_time_ms(sb, handle, lambda: sb._gemm(handle, tr[0], tr[1], size, size, size, 1.0, a, size, b, size,
                                                          0.0, c, size), 10, warm=10)
        tf = 2.0 * size ** 3 / ms / 1e9
        extras[f"bf16_f32out_{tr}_{size}"] = {"ms": round(ms, 4), "tflops": round(tf, 1),
                                              "pct_mfma_peak": round(100 * tf / PEAK_TFLOPS["bf16"], 2)}
    for buf in (a, b, c):
        helper.deallocate(buf)

    # a caller of the GEMM path: STRSM, left / lower / no transpose / non-unit, size x size.
    # Off-diagonal entries are scaled so that the solution stays O(1) over thousands of rows.
    ta_t = torch.rand(size, size, dtype=torch.float32, device="cuda").mul_(1.0 / size)
    ta_t.diagonal().fill_(2.0)
    tb_t = torch.rand(size * size, dtype=torch.float32, device="cuda")
    from sycl_blas_amd import DeviceBuffer
    a = DeviceBuffer.from_pointer(ta_t.data_ptr(), "float32", size * size, ta_t)
    b = DeviceBuffer.from_pointer(tb_t.data_ptr(), "float32", size * size, tb_t)
    torch.cuda.synchronize()
    ms = device_time_ms(sb, handle, lambda: sb._trsm(handle, "l", "l", "n", "n", size, size, 1.0, a, size, b, size),
                        3, warm=1)
    diag_launches, panel_launches = handle.last_trsm_launches()
    extras[f"strsm_llnn_{size}"] = {"ms": round(ms, 3), "tflops": round(float(size) ** 3 / ms / 1e9, 2),
                                    "launches": {"diagonal": diag_launches, "panel": panel_launches}}
    del a, b, ta_t, tb_t
    return extras


def cpu_baseline(dtype, ta, tb, size, seed):
    """The CPU restatement (oracle) timed on the host cores on a bounded sample:
    the first `cols` columns of C of the same product (same A, a column slab of B)."""
    import oracle
    from sycl_blas_amd import uniform_reference

    npdt = NP_DTYPE[dtype]
    m = k = size
    # about 10 s of CPU work on 16 cores at 8192^3 (0.1-0.2 TFLOP/s): the first 8192 columns
    cols = min(size, 8192)
    a = uniform_reference(m * k, seed).astype(npdt)
    lda = k if ta != "n" else m
    if tb == "n":  # B is k x n, a column slab is contiguous
        b = uniform_reference(k * cols, seed + 1).astype(npdt)
        ldb = k
    else:          # B is stored n x k: take the first `cols` rows of every column
        full = uniform_reference(size * k, seed + 1).astype(npdt).reshape(k, size)
        b = np.ascontiguousarray(full[:, :cols]).ravel()
        ldb = cols
    c = np.zeros(m * cols, dtype=npdt)
    oracle.gemm_backend(ta, tb, m, min(cols, 8), k, 1.0, a, lda, 0, b, ldb, 0, 0.0, c, m, 0)  # page in, spin up threads
    t0 = time.perf_counter()
    st = oracle.gemm_backend(ta, tb, m, cols, k, 1.0, a, lda, 0, b, ldb, 0, 0.0, c, m, 0)
    dt = time.perf_counter() - t0
    assert st == 0
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    # the reference's own ground-truth library (system CBLAS) on the same host cores, for scale
    openblas = None
    try:
        from scipy.linalg import blas
        fn = blas.sgemm if npdt == np.float32 else blas.dgemm
        sz = 4096
        x = np.asfortranarray(a[: sz * sz].reshape(sz, sz))
        y = np.asfortranarray(b[: sz * sz].reshape(sz, sz))
        fn(1.0, x, y)
        t1 = time.perf_counter()
        fn(1.0, x, y)
        openblas = round(2.0 * sz ** 3 / (time.perf_counter() - t1) / 1e12, 4)
    except Exception:
        pass
    # BASELINE config #1 (SGEMM 1024^3 NN on the reference's CPU back end): the restatement of the kernel
    # that back end picks for it (no_local, partial, Tile<4,4,4,4>; src/interface/blas3/backend/default.hpp:98-112),
    # whole product, best of three
    n1 = 1024
    a1 = uniform_reference(n1 * n1, 778).astype(np.float32)
    b1 = uniform_reference(n1 * n1, 779).astype(np.float32)
    c1 = np.zeros(n1 * n1, dtype=np.float32)
    best = 1e30
    for _ in range(4):
        t1 = time.perf_counter()
        oracle.gemm_default_tile("n", "n", n1, n1, n1, 1.0, a1, n1, b1, n1, 0.0, c1, n1)
        best = min(best, time.perf_counter() - t1)
    config1 = {"value": round(2.0 * n1 ** 3 / best / 1e12, 5), "unit": "TFLOP/s", "cores": threads, "kind": "port",
               "ms": round(best * 1e3, 3),
               "sample": "whole SGEMM 1024^3 NN (config #1), restatement of the reference's DEFAULT-back-end kernel "
                         "(no_local partial, Tile<4,4,4,4>, 16x16 C tile per work-group), OpenMP over work-groups"}
    return {
        "config1_sgemm_1024": config1,
        "openblas_tflops_4096": openblas,
        "value": round(2.0 * m * cols * k / dt / 1e12, 5),
        "unit": "TFLOP/s",
        "cores": threads,
        "kind": "port",
        "sample": f"first {cols} of {size} columns of C of the same {dtype} {size}^3 {ta}{tb} product "
                  f"({2.0 * m * cols * k / 1e9:.0f} GFLOP, {dt:.2f} s), OpenMP over columns",
    }, c, cols, c1


def load_traffic(dtype, trans, size):
    """HBM bytes per launch from a separate rocprofv3 --pmc pass (profiles/*.json), or None."""
    for path in (os.path.join(ROOT, "profiles", "r03", "r03_pmc_traffic.json"), os.path.join(ROOT, "profiles", "r02", "r02_pmc_traffic.json")):
        try:
            with open(path) as f:
                table = json.load(f)
            if f"{dtype}_{trans}_{size}" in table:
                return table[f"{dtype}_{trans}_{size}"]
        except Exception:
            pass
    return None


class StagedGather:
    """--rehearse only: gloo cannot gather GPU tensors, so stage the collective through the host.
    Same call shape as torch.distributed for what gathered_gemm uses."""

    def __init__(self, dist, torch):
        self.dist, self.torch = dist, torch

    def all_gather_into_tensor(self, out, inp, group=None, async_op=False):
        self.torch.cuda.current_stream().synchronize()
        host_out = self.torch.empty(out.numel(), dtype=out.dtype)
        self.dist.all_gather_into_tensor(host_out, inp.cpu().contiguous(), group=group)
        out.copy_(host_out)
        return None


def timed_steps(torch, dist, dist_on, step, steps, handle=None, events=None):
    """EXACTLY `steps` steps between a barrier + device synchronisation on both sides; seconds on this rank.
    With `handle` and `events`, a HIP event is recorded on the handle's stream right before the first and right
    after the last step (device time of the same region)."""
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    if events is not None:
        events.append(handle.record_event())
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if events is not None:
        events.append(handle.record_event())
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def max_over_ranks(torch, dist, dist_on, seconds, on_cpu):
    if not dist_on:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if on_cpu else "cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the GEMM path has no CPU fallback")
    if args.rehearse:
        local = 0
    torch.cuda.set_device(local)
    dist_on = world > 1 or args.one_rank_rccl
    if args.one_rank_rccl:
        if world != 1:
            raise SystemExit("--one-rank-rccl is a single-process run")
        for key, val in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29561"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(key, val)
    if dist_on:
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import sycl_blas_amd as sb
    from sycl_blas_amd import _lib
    from sycl_blas_amd.distributed import (Communicator, DistError, TorchStreams, band_widths, block_of, gathered_gemm, grid_for,
                                           pick_bands, split_range, torch_allgather)

    # Everything of this rank -- allocations, fills, GEMMs -- is ordered on ONE explicit stream that
    # torch and the library's handle share; the traffic runs on a second stream behind events.
    compute = torch.cuda.Stream()
    torch.cuda.set_stream(compute)
    handle = sb.SB_Handle(local, compute.cuda_stream)
    ta, tb = args.trans[0], args.trans[1]
    size = args.size
    pr, pc = grid_for(world)
    strong = dist_on and args.mode == "strong"
    gather = strong and not args.no_gather
    staged = StagedGather(dist, torch) if args.rehearse else dist
    extras = {}
    native, window, transport = None, -1, "none"

    if strong:
        # ONE size^3 product; A, B (replicated, same seeds everywhere) and the full C live on every rank.
        gm = gn = gk = size
        ops = Operands(torch, sb, handle, args.dtype, ta, tb, gm, gn, gk, 12345)
        gemm_calls = []

        def local_gemm(tra, trb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc):
            gemm_calls.append((m, n, k))
            sb._gemm(handle, tra, trb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc)

        # Transport: the native layer's mesh first, its RCCL route second, torch.distributed last.  Every rank
        # must end up with the same choice, so a rank that fails says so to all.
        def agree(ok):
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if args.rehearse else "cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(flag.item())

        want = args.transport
        if args.one_rank_rccl:
            want = "rccl"
            os.environ["SBAMD_DIST_PREFER"] = "rccl"
        # a peer that never answers must end a device-side wait within seconds, not hang the run
        os.environ.setdefault("SBAMD_DIST_TIMEOUT_MS", "5000")
        state = {"native": None, "window": -1, "transport": "none", "bands": 1, "my_cols": []}

        def close_native():
            if state["native"] is not None:
                try:
                    state["native"].close()
                except DistError:
                    pass
            state["native"], state["window"] = None, -1

        def set_up(candidate):
            """Create the communicator of one transport; every rank learns whether all of them succeeded."""
            ok = True
            try:
                if candidate == "mesh":
                    flags = _lib.DIST_MESH if args.rehearse else (_lib.DIST_MESH | _lib.DIST_RCCL)
                    state["native"] = Communicator(handle, rank, world, torch_allgather(dist), flags)
                    state["window"] = state["native"].register(ops.c)
                elif candidate == "rccl":
                    if want == "rccl":
                        os.environ["SBAMD_DIST_PREFER"] = "rccl"
                    state["native"] = Communicator(handle, rank, world, torch_allgather(dist), _lib.DIST_RCCL)
            except DistError as exc:
                ok, extras[f"{candidate}_setup_error"] = False, str(exc)
            if not agree(ok):
                close_native()
                return False
            state["transport"] = candidate
            if candidate == "mesh":
                # rank r owns the columns split_range(gn, world, r), multiplied in bands that shrink towards the end
                my0, my1 = split_range(gn, world, rank)
                widths = band_widths(my1 - my0, args.bands, world)
                state["bands"], state["my_cols"] = len(widths), []
                for w in widths:
                    state["my_cols"].append((my0, my0 + w))
                    my0 += w
            else:
                nb = args.bands or pick_bands(world, gm, gn, gk)
                if gn % (nb * world):
                    raise SystemExit(f"--size must be divisible by bands * ranks = {nb * world}")
                w = gn // nb // world
                state["bands"] = nb
                state["my_cols"] = [(bi * (gn // nb) + rank * w, bi * (gn // nb) + (rank + 1) * w) for bi in range(nb)]
            return True

        def c_digest_agrees():
            digest = ops.tc_t.view(torch.int32 if args.dtype == "f32" else torch.int64).sum(dtype=torch.int64).reshape(1)
            lo, hi = digest.clone(), digest.clone()
            if args.rehearse:
                lo, hi = lo.cpu(), hi.cpu()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            return bool(lo.item() == hi.item())

        def proves_itself():
            """One untimed step of the chosen transport before anything is timed: no wait for a peer timed out on any
            rank and every rank holds the same C.  A transport that fails here is dropped for the next one."""
            ok = True
            try:
                step()
                if state["native"] is not None:
                    state["native"].status()
                torch.cuda.synchronize()
            except DistError as exc:
                ok, extras[f"{state['transport']}_first_step_error"] = False, str(exc)
            same = c_digest_agrees()
            if ok and not same:
                extras[f"{state['transport']}_first_step_error"] = "the ranks' C differ after the first step"
            return agree(ok and same)

        streams = TorchStreams(torch, compute)

        def own_columns_only():
            # this rank's band multiplications with nothing travelling behind them
            for c0, c1 in state["my_cols"]:
                local_gemm(ta, tb, gm, c1 - c0, gk, 1.0, ops.a, ops.lda, ops.b + (c0 * ops.ldb if tb == "n" else c0), ops.ldb, 0.0,
                           ops.c + c0 * gm, gm)

        def step():
            if not gather:
                own_columns_only()
            elif state["native"] is not None:
                state["native"].gathered_gemm(ta, tb, gm, gn, gk, 1.0, ops.a, ops.lda, ops.b, ops.ldb, 0.0, ops.c, gm, state["window"],
                                              args.bands if state["transport"] == "mesh" else state["bands"])
            else:
                gathered_gemm(staged, local_gemm, rank, world, ta, tb, gm, gn, gk, 1.0, ops.a, ops.lda, ops.b, ops.ldb, 0.0,
                              ops.c, ops.tc_t, bands=state["bands"], streams=streams)

        order = []
        if gather and want in ("auto", "mesh"):
            order.append("mesh")
        if gather and want in ("auto", "rccl") and not args.rehearse:
            order.append("rccl")
        order.append("torch")
        for candidate in order:
            if set_up(candidate) and (not gather or proves_itself()):
                break
            close_native()
        native, window, transport, bands, my_cols = (state[k] for k in ("native", "window", "transport", "bands", "my_cols"))
        if not gather:
            transport = "none"
        bm, bn = gm, gn // world
    else:
        if world > 1:  # weak: 2-D C blocks, every block a full size^3 product, nothing exchanged
            gm, gn, gk = pr * size, pc * size, size
        else:
            gm, gn, gk = size, size, size
        (r0, r1), (c0, c1) = block_of(rank, pr, pc, gm, gn)
        bm, bn = r1 - r0, c1 - c0
        ops = Operands(torch, sb, handle, args.dtype, ta, tb, bm, bn, gk, 12345 + 2 * rank)
        bands = 1

        def step():
            ops.run(sb, handle)

    for _ in range(args.warmup):
        step()
    region = []
    wall = timed_steps(torch, dist, dist_on, step, args.steps, handle, region)
    wall = max_over_ranks(torch, dist, dist_on, wall, args.rehearse)
    flops_job = 2.0 * gm * gn * gk
    value = flops_job * args.steps / wall / 1e12

    # Device time of this rank's multiplications alone (HIP events on the stream the kernels run on).
    if strong:
        kernel_ms = device_time_ms(sb, handle, own_columns_only, max(3, args.steps // 2))
        shapes = list(dict.fromkeys(gemm_calls))
        extras["strong"] = {"bands": bands, "band_gemm_shapes": shapes, "gemm_only_ms_per_step": round(kernel_ms, 4),
                            "gather": gather}
        if native is not None:
            native.status()  # raises if a rank ever waited for a peer in vain
            extras["strong"]["transport"] = transport
            extras["strong"]["band_columns_of_rank0"] = [list(c) for c in my_cols]
        if gather:
            # the traffic alone (this rank's whole column panel at once, nothing to hide behind)
            def gather_only():
                if transport == "mesh":
                    native.all_gather_c(1, world, gm, gn, ops.c, gm, window)
                    return
                works = []
                width = gn // bands
                for bi in range(bands):
                    col = bi * width + rank * (width // world)
                    out = ops.tc_t[bi * width * gm:(bi + 1) * width * gm]
                    works.append(staged.all_gather_into_tensor(out, ops.tc_t[col * gm:(col + width // world) * gm],
                                                               async_op=True))
                for wk in works:
                    if wk is not None:
                        wk.wait()
            for _ in range(2):
                gather_only()
            t_g = timed_steps(torch, dist, dist_on, gather_only, max(3, args.steps // 2)) / max(3, args.steps // 2)
            extras["strong"]["gather_only_ms_per_step"] = round(max_over_ranks(torch, dist, dist_on, t_g, args.rehearse) * 1e3, 4)
            # every rank must hold the same C, and a foreign part must equal this rank's own recomputation
            # of it (same kernel, same shape: bit for bit)
            step()
            torch.cuda.synchronize()
            if dist_on:
                dist.barrier()
            digest = ops.tc_t.view(torch.int32 if args.dtype == "f32" else torch.int64).sum(dtype=torch.int64).reshape(1)
            lo, hi = digest.clone(), digest.clone()
            if args.rehearse:
                lo, hi = lo.cpu(), hi.cpu()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            peer = (rank + 1) % world
            if transport == "mesh":
                col = split_range(gn, world, peer)[0]  # the peer's first band
                w = band_widths(split_range(gn, world, peer)[1] - col, args.bands, world)[0]
            else:
                w = gn // bands // world
                col = peer * w  # the peer's part of band 0
            theirs = ops.tc_t[col * gm:(col + w) * gm].clone()
            sb._gemm(handle, ta, tb, gm, w, gk, 1.0, ops.a, ops.lda, ops.b + (col * ops.ldb if tb == "n" else col), ops.ldb, 0.0,
                     ops.c + col * gm, gm)
            torch.cuda.synchronize()
            extras["strong"]["gathered_c_identical_on_all_ranks"] = bool(lo.item() == hi.item())
            extras["strong"]["peer_part_equals_local_recompute"] = bool(torch.equal(theirs, ops.tc_t[col * gm:(col + w) * gm]))
            if dist_on:
                dist.barrier()  # nobody starts the next collective while a peer still recomputes into its C
        # weak scaling, reported beside the headline, never as `value`
        wops = Operands(torch, sb, handle, args.dtype, ta, tb, size, size, size, 999 + rank)
        for _ in range(3):
            wops.run(sb, handle)
        t_w = timed_steps(torch, dist, dist_on, lambda: wops.run(sb, handle), max(3, args.steps // 2)) / max(3, args.steps // 2)
        t_w = max_over_ranks(torch, dist, dist_on, t_w, args.rehearse)
        extras["weak_no_exchange"] = {"job": f"{pr * size}x{pc * size}x{size}", "tflops": round(2.0 * size ** 3 * world / t_w / 1e12, 2),
                                      "ms_per_step": round(t_w * 1e3, 4)}
        del wops
    else:
        kernel_ms = region[0].elapsed_ms(region[1]) / args.steps  # device time per launch over the timed region itself

    if rank == 0:
        achieved = 2.0 * bm * bn * gk / (kernel_ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.dtype]
        sd = "s" if args.dtype == "f32" else "d"
        if strong:
            workload = (f"{sd.upper()}GEMM {size}x{size}x{size} {args.trans.upper()} cut over {world} GPUs (columns of C dealt in {bands} "
                        f"band(s)), alpha=1 beta=0, through sbamd_{sd}gemm" +
                        (f", full C gathered on every GPU inside the timed region (transport: {transport}, overlapped)" if gather
                         else ", C left distributed"))
        else:
            workload = (f"{sd.upper()}GEMM {size}x{size}x{size} {args.trans.upper()} per GPU C-block, alpha=1 beta=0, "
                        f"through sbamd_{sd}gemm")
        out = {
            "metric": "SGEMM/DGEMM TFLOP/s at 8192^3 + % of MI355X MFMA peak, 1/2/4/8 GPUs",
            "value": round(value, 3),
            "unit": "TFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic: uniform [-2, 5], counter-based generator on the device, seed 12345",
            "config": {
                "workload": workload,
                "job": f"{gm}x{gn}x{gk}",
                "grid": f"1x{world} column panels" if strong else f"{pr}x{pc} C blocks",
                "transport": transport if strong else "none",
                "gather": bool(gather),
                "pct_of_mfma_peak_per_gpu": round(100.0 * value / world / peak, 2),
            },
            "roofline": {
                "bound": "mfma",
                "achieved": round(achieved, 3),
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4),
                "traffic": load_traffic(args.dtype, args.trans, size) if not dist_on else None,
                "kernel": f"gemm_mfma_kernel<{'float' if args.dtype == 'f32' else 'double'}> (tile family "
                          f"{handle.last_kernel()}), {kernel_ms:.4f} ms per step of this rank's multiplications",
            },
        }
        if args.rehearse:
            out["config"]["rehearsal"] = "all ranks on GPU 0 (peers mapped through HIP IPC on one device): NOT a multi-GPU measurement"
        if args.one_rank_rccl:
            out["config"]["rehearsal"] = "the N>1 code path of the native layer on a one-rank RCCL communicator: NOT a multi-GPU measurement"
        if extras:
            out["extras"] = extras

        if not args.no_extras and not dist_on:
            out["extras"] = side_measurements(torch, sb, handle, size)

        if not args.no_cpu and not dist_on:
            try:
                base, c_cpu, cols, c_1024 = cpu_baseline(args.dtype, ta, tb, size, 12345)
                out["cpu_baseline"] = base
                # config #1 on the GPU (same seeds as the CPU leg) must give the CPU kernel's bits
                # (a single-launch family is pinned for this check: the planner may cut K in two for this
                # shape, which is the same product with one more rounding)
                from sycl_blas_amd import _lib
                o1 = Operands(torch, sb, handle, "f32", "n", "n", 1024, 1024, 1024, 778)
                handle.set_kernel(_lib.KERNEL_SMALL)
                o1.run(sb, handle)
                handle.set_kernel(_lib.KERNEL_AUTO)
                got1 = np.empty(1024 * 1024, dtype=np.float32)
                handle.wait(sb.helper.copy_to_host(handle, o1.c, got1, got1.size))
                base["config1_sgemm_1024"]["gpu_bitwise_equal"] = bool(np.array_equal(got1.view(np.uint8), c_1024.view(np.uint8)))
                # the GPU result of the same product must equal the CPU restatement bit for bit
                got = np.empty(size * cols, dtype=NP_DTYPE[args.dtype])
                handle.wait(sb.helper.copy_to_host(handle, ops.c, got, got.size))
                out["config"]["cpu_sample_bitwise_equal"] = bool(np.array_equal(got.view(np.uint8), c_cpu.view(np.uint8)))
            except Exception as exc:  # the baseline is a reported extra, never the product path
                out["cpu_baseline"] = {"value": None, "unit": "TFLOP/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {exc!r}"}
        print(json.dumps(out), flush=True)

    if native is not None:
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        native.close()
    handle.close()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
