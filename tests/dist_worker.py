"""One rank of the multi-process tests of the native multi-GPU layer (started by tests/test_dist_native_gpu.py
through torch.distributed.run; not collected by pytest).  Every rank uses GPU 0 -- the mesh transport then maps the
peers' buffers of the same GPU through HIP IPC, which exercises the real exchange, flag and push code on a one-GPU
box -- and the out-of-band exchange goes through the gloo group.  Results are checked bit for bit against the CPU
oracle replaying each multiplication's K cut."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import torch.distributed as dist

    import oracle
    import sycl_blas_amd as sb
    from sycl_blas_amd import _lib, helper
    from sycl_blas_amd.distributed import Communicator, block_of, grid_for, split_range, torch_allgather

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dtype = np.float32 if (len(sys.argv) < 2 or sys.argv[1] == "float32") else np.float64
    m, n, k = 200, 96 * world + 40, 1500
    rng = np.random.default_rng(777)  # the same operands on every rank
    a = rng.uniform(-2, 5, m * k).astype(dtype)
    b = rng.uniform(-2, 5, k * n).astype(dtype)
    c0 = rng.uniform(-2, 5, m * n).astype(dtype)
    with sb.SB_Handle(0) as h:
        a_g, b_g, c_g = (helper.allocate(h, dtype, x.size) for x in (a, b, c0))
        for host, dev in ((a, a_g), (b, b_g), (c0, c_g)):
            helper.copy_to_device(h, host, dev)
        h.wait()
        with Communicator(h, rank, world, torch_allgather(dist), _lib.DIST_MESH, workspace_bytes=m * n * a.itemsize) as comm:
            win = comm.register(c_g)

            def download():
                out = np.empty_like(c0)
                h.wait(helper.copy_to_host(h, c_g, out))
                return out

            # --- strong-scaling schedule: whole C on every rank; the oracle replays every rank's bands and cuts
            comm.gathered_gemm("n", "n", m, n, k, 1.5, a_g, m, b_g, k, 0.5, c_g, m, win, 0)
            h.wait()
            got = download()
            plans = [None] * world
            dist.all_gather_object(plans, comm.last_bands())
            want = c0.copy()
            for bands in plans:
                for (col0, col1, depth, k_chunk) in bands:
                    form = oracle.FORM_SPLITK if depth > 1 else oracle.FORM_REF
                    assert oracle.gemm_backend("n", "n", m, col1 - col0, k, 1.5, a, m, 0, b[col0 * k:], k, 0, 0.5,
                                               want[col0 * m:], m, 0, form=form, depth=depth, k_chunk=k_chunk) == 0
            covered = sorted((b0, b1) for bands in plans for (b0, b1, _, _) in bands)
            assert covered[0][0] == 0 and covered[-1][1] == n and all(x[1] == y[0] for x, y in zip(covered, covered[1:]))
            assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), f"rank {rank}: gathered GEMM differs from the oracle"

            # --- C blocks on the process grid, then the all-gather of the blocks
            pr, pc = grid_for(world)
            helper.copy_to_device(h, c0, c_g)
            h.wait()
            comm.barrier()
            h.set_kernel(_lib.KERNEL_LARGE)  # a family that never cuts K: the plain chain
            comm.block_gemm(pr, pc, "t", "n", m, n, k, 2.0, a_g, k, b_g, k, 0.0, c_g, m)
            h.set_kernel(_lib.KERNEL_AUTO)
            comm.all_gather_c(pr, pc, m, n, c_g, m, win)
            h.wait()
            got = download()
            want = c0.copy()
            a_t = a  # read as k x m with lda = k
            for r in range(world):
                (r0, r1), (q0, q1) = block_of(r, pr, pc, m, n)
                blk = np.zeros((r1 - r0) * (q1 - q0), dtype)
                assert oracle.gemm_backend("t", "n", r1 - r0, q1 - q0, k, 2.0, a_t[r0 * k:], k, 0, b[q0 * k:], k, 0, 0.0, blk,
                                           r1 - r0, 0) == 0
                want.reshape(n, m)[q0:q1, r0:r1] = blk.reshape(q1 - q0, r1 - r0)
            assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), f"rank {rank}: gathered blocks differ from the oracle"

            # --- K cut across the ranks, slabs gathered: partial products summed in rank order
            helper.copy_to_device(h, c0, c_g)
            h.wait()
            comm.barrier()
            h.set_kernel(_lib.KERNEL_LARGE)
            comm.splitk_gemm("n", "n", m, n, k, 1.5, a_g, m, b_g, k, 0.5, c_g, m, win, gather=True)
            h.set_kernel(_lib.KERNEL_AUTO)
            h.wait()
            got = download()
            want = c0.copy()
            chunk = -(-k // world)
            assert oracle.gemm_backend("n", "n", m, n, k, 1.5, a, m, 0, b, k, 0, 0.5, want, m, 0, form=oracle.FORM_SPLITK,
                                       depth=world, k_chunk=chunk) == 0
            assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), f"rank {rank}: K cut across ranks differs from the oracle"
            comm.status()
            h.wait()
            comm.barrier()
            h.wait()
            comm.unregister(win)
        for buf in (a_g, b_g, c_g):
            helper.deallocate(buf)
    dist.barrier()
    if rank == 0:
        print(f"dist worker ok: {world} ranks, {np.dtype(dtype).name}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
