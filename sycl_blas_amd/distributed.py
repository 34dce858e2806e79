"""Multi-GPU GEMM: 2-D C-block partition over the GPUs of one node.

The reference has no multi-device code (one sycl::queue per SB_Handle,
include/sb_handle/portblas_handle.h:46-60), so this layer is new.  GEMM shards
naturally: C blocks are independent.  One process drives one GPU; the ranks form
a Pr x Pc grid, rank (i, j) owns C block (i, j) and needs only the A row-panel i
and the B column-panel j, so with the panels resident the block products need NO
exchange.  Data moves only where an output has to be assembled.

The product path is native: `Communicator` binds the C ABI of include/sbamd_dist.h
(sycl_blas_amd/csrc/sbamd_dist.hip) -- C blocks, the strong-scaling schedule with
the whole C on every rank, the K cut across ranks -- over two transports: direct
stores into the peers' memory across the xGMI mesh (buffers exchanged once as HIP
IPC handles) and RCCL collectives.  The partition arithmetic below calls the same
C functions, so there is one implementation of it.

The functions further down that take a `dist` module and a `local_gemm` callable
(all_gather_c, gathered_gemm, splitk_gemm) are the same schedules over
torch.distributed; they exist so that the rank logic can be exercised by CPU ranks
over gloo, without a GPU.
"""
import ctypes as C

from . import _lib

GRIDS = {1: (1, 1), 2: (1, 2), 4: (2, 2), 8: (2, 4)}


def grid_for(world):
    """Process grid Pr x Pc (Pr <= Pc) for `world` ranks (sbamd_dist_grid)."""
    pr, pc = C.c_int(), C.c_int()
    _lib.load().sbamd_dist_grid(int(world), C.byref(pr), C.byref(pc))
    return pr.value, pc.value


def split_range(extent, parts, index):
    """Balanced [begin, end) of piece `index` when `extent` is cut into `parts` (sbamd_dist_split_range)."""
    b, e = C.c_int64(), C.c_int64()
    _lib.load().sbamd_dist_split_range(int(extent), int(parts), int(index), C.byref(b), C.byref(e))
    return b.value, e.value


def band_widths(cols, bands, world):
    """Widths of the column bands one rank's `cols` columns are multiplied in by Communicator.gathered_gemm
    (sbamd_dist_band_widths): `bands` equal parts, or for bands == 0 halves that shrink towards the end."""
    out = (C.c_int64 * _lib.DIST_MAX_BANDS)()
    count = _lib.load().sbamd_dist_band_widths(int(cols), int(bands), int(world), out)
    return [out[i] for i in range(count)]


def coords_of(rank, pr, pc):
    """Grid coordinates (i, j) of a rank: row-major over the Pr x Pc grid."""
    if not 0 <= rank < pr * pc:
        raise ValueError("rank outside the process grid")
    return rank // pc, rank % pc


def block_of(rank, pr, pc, m, n):
    """((row_begin, row_end), (col_begin, col_end)) of the C block a rank owns (sbamd_dist_block_of)."""
    r0, r1, c0, c1 = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    if _lib.load().sbamd_dist_block_of(int(rank), int(pr), int(pc), int(m), int(n), C.byref(r0), C.byref(r1), C.byref(c0),
                                       C.byref(c1)) != _lib.OK:
        raise ValueError("rank outside the process grid")
    return (r0.value, r1.value), (c0.value, c1.value)


def panel_offsets(rank, pr, pc, transa, transb, m, n, lda, ldb):
    """Element offsets of a rank's A row-panel and B column-panel inside FULL
    column-major A and B (for callers that keep the whole operands on every GPU).

    op(A) row r0 starts at A + r0 ('n') or A + r0*lda (transposed); op(B) column
    c0 starts at B + c0*ldb ('n') or B + c0 (transposed).
    """
    (r0, _), (c0, _) = block_of(rank, pr, pc, m, n)
    a_off = r0 if transa.lower() == "n" else r0 * lda
    b_off = c0 * ldb if transb.lower() == "n" else c0
    return a_off, b_off


def block_gemm(local_gemm, rank, pr, pc, transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc,
               full_operands=True):
    """Compute this rank's C block of the global m x n x k product.

    local_gemm(transa, transb, bm, bn, k, alpha, a, lda, b, ldb, beta, c, ldc) runs one
    GEMM on views; a, b, c must support `+ element_offset`.
    full_operands=True : a, b, c address the FULL matrices (replicated inputs,
                         C written in place into the rank's block).
    full_operands=False: a is the rank's row-panel, b its column-panel and c its
                         block (ldc is the block's leading dimension).
    Returns ((r0, r1), (c0, c1)).
    """
    (r0, r1), (c0, c1) = block_of(rank, pr, pc, m, n)
    bm, bn = r1 - r0, c1 - c0
    if bm == 0 or bn == 0:
        return (r0, r1), (c0, c1)
    if full_operands:
        a_off, b_off = panel_offsets(rank, pr, pc, transa, transb, m, n, lda, ldb)
        local_gemm(transa, transb, bm, bn, k, alpha, a + a_off, lda, b + b_off, ldb, beta, c + (r0 + c0 * ldc), ldc)
    else:
        local_gemm(transa, transb, bm, bn, k, alpha, a, lda, b, ldb, beta, c, ldc)
    return (r0, r1), (c0, c1)


def all_gather_c(dist, torch, c_block, rank, pr, pc, m, n, group=None):
    """Assemble the full column-major C (a flat tensor of m*n) on every rank.

    c_block is the rank's block packed column-major (bm*bn elements).  Blocks may
    have different sizes (ragged m or n), so they travel padded to the largest
    block and are trimmed on arrival.  One all-gather per call: on xGMI each rank
    sends its block to its 7 peers over separate links.
    """
    world = pr * pc
    sizes = []
    for r in range(world):
        (a0, a1), (b0, b1) = block_of(r, pr, pc, m, n)
        sizes.append((a1 - a0) * (b1 - b0))
    biggest = max(sizes)
    send = c_block
    if c_block.numel() != biggest:
        send = torch.zeros(biggest, dtype=c_block.dtype, device=c_block.device)
        send[:c_block.numel()] = c_block
    recv = torch.empty(world * biggest, dtype=c_block.dtype, device=c_block.device)
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    full = torch.empty(m * n, dtype=c_block.dtype, device=c_block.device)
    full_2d = full.view(n, m)  # row j of this view = column j of C
    for r in range(world):
        (a0, a1), (b0, b1) = block_of(r, pr, pc, m, n)
        if a1 > a0 and b1 > b0:
            blk = recv[r * biggest:r * biggest + sizes[r]].view(b1 - b0, a1 - a0)
            full_2d[b0:b1, a0:a1] = blk
    return full


def splitk_gemm(dist, torch, local_gemm, make_buffer, rank, world, transa, transb, m, n, k, alpha, a, lda, b, ldb,
                group=None):
    """K cut across ranks: C = alpha * sum_r op(A)[:, K_r] * op(B)[K_r, :].

    a, b address the FULL operands; rank r multiplies its K range into a full-size
    partial (beta = 0), then a reduce-scatter sums the partials and leaves rank r
    the columns split_range(n, world, r) of C, packed column-major.
    make_buffer(count) -> (torch tensor, view usable by local_gemm) allocates the
    partial.  Needs n % world == 0 (equal slabs for reduce_scatter_tensor).
    Returns (slab tensor, (col_begin, col_end)).
    """
    if n % world:
        raise ValueError("splitk_gemm needs n divisible by the number of ranks")
    k0, k1 = split_range(k, world, rank)
    part_t, part_v = make_buffer(m * n)
    if k1 > k0:
        a_off = k0 * lda if transa.lower() == "n" else k0
        b_off = k0 if transb.lower() == "n" else k0 * ldb
        local_gemm(transa, transb, m, n, k1 - k0, alpha, a + a_off, lda, b + b_off, ldb, 0.0, part_v, m)
    else:
        part_t.zero_()
    slab = torch.empty(m * (n // world), dtype=part_t.dtype, device=part_t.device)
    dist.reduce_scatter_tensor(slab, part_t, op=dist.ReduceOp.SUM, group=group)
    return slab, split_range(n, world, rank)


def band_columns(rank, world, n, bands):
    """Column ranges [(begin, end), ...] a rank computes in gathered_gemm, one per band.

    Band b covers the global columns [b*W, (b+1)*W), W = n / bands, and is dealt to the
    ranks in order: rank r takes [b*W + r*w, b*W + (r+1)*w), w = W / world.  In a
    column-major C a band is then one contiguous run of m*W elements in which rank r's
    part sits at r*(m*w): exactly the layout an in-place all-gather produces.
    Needs n % (bands * world) == 0.
    """
    if n % (bands * world):
        raise ValueError("band_columns needs n divisible by bands * ranks")
    width = n // bands
    w = width // world
    return [(b * width + rank * w, b * width + (rank + 1) * w) for b in range(bands)]


def pick_bands(world, m, n, k, max_bands=4, min_cols=512):
    """How many column bands gathered_gemm cuts the product into: as many as divide n evenly
    while every band GEMM (m x n/(bands*world) x k) keeps at least `min_cols` columns, so that it
    still fills the chip with 64x64 / 128x128 tiles.  More bands hide more of the gather behind
    the multiplication; fewer keep the multiplications efficient."""
    bands = 1
    for cand in range(2, max_bands + 1):
        if n % (cand * world) == 0 and n // (cand * world) >= min_cols:
            bands = cand
    return bands


class _NoStreams:
    """Stream plumbing for ranks without a GPU (gloo tests): everything is already ordered."""

    def gemm_done(self):
        return None

    def comm_section(self, after):
        import contextlib
        return contextlib.nullcontext()

    def finish(self, works):
        for w in works:
            if w is not None:
                w.wait()


class TorchStreams:
    """Two HIP streams per rank: the GEMMs run on `compute` (the stream the library's handle was
    created on), the collectives are launched from `comm`, which waits for each band's event."""

    def __init__(self, torch, compute, comm=None):
        self.torch = torch
        self.compute = compute
        self.comm = comm if comm is not None else torch.cuda.Stream()

    def gemm_done(self):
        ev = self.torch.cuda.Event()
        ev.record(self.compute)
        return ev

    def comm_section(self, after):
        self.comm.wait_event(after)
        return self.torch.cuda.stream(self.comm)

    def finish(self, works):
        with self.torch.cuda.stream(self.comm):
            for w in works:
                if w is not None:
                    w.wait()  # the comm stream waits for the collective
        # the next multiplication into these slots must not start before the gathers have read them
        self.compute.wait_stream(self.comm)


def gathered_gemm(dist, local_gemm, rank, world, transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, c_tensor,
                  bands=1, streams=None, group=None, in_place=True):
    """C = alpha * op(A) * op(B) + beta * C with the full column-major C (ldc = m) assembled on
    every rank, multiplication and gather overlapped.

    a, b address the FULL operands (replicated; a rank reads all of A and its own columns of
    op(B)); c is the view of the rank's full C that local_gemm writes through, c_tensor the same
    memory as a flat torch tensor of m*n elements for the collectives.  For band 0, 1, ...: the
    rank multiplies its columns of the band straight into their final place, records an event,
    and the band's all-gather is launched from the communication stream behind that event while
    the next band is multiplied.  No re-layout: a band is contiguous and the ranks' parts lie in
    rank order (band_columns).  With beta != 0 every rank must hold the same C on entry (its own
    columns are read, the others are overwritten by the gather).
    in_place=False sends a copy of the rank's part instead of the slice of the output itself
    (for back ends that do not accept an input that aliases the output).
    Returns the list of (begin, end) column ranges this rank multiplied.
    """
    streams = streams or _NoStreams()
    cols = band_columns(rank, world, n, bands)
    width = n // bands
    works = []
    for bi, (c0, c1) in enumerate(cols):
        b_off = c0 * ldb if transb.lower() == "n" else c0
        local_gemm(transa, transb, m, c1 - c0, k, alpha, a, lda, b + b_off, ldb, beta, c + c0 * m, m)
        done = streams.gemm_done()
        if dist is None:  # no process group at all; a one-rank group still issues its collectives
            continue
        out = c_tensor[bi * width * m:(bi + 1) * width * m]
        mine = c_tensor[c0 * m:c1 * m]
        with streams.comm_section(done):
            works.append(dist.all_gather_into_tensor(out, mine if in_place else mine.clone(), group=group, async_op=True))
    streams.finish(works)
    return cols


# ----------------------------------------------------------------- the native layer


class DistError(RuntimeError):
    """A failure reported by the native multi-GPU layer (HIP, RCCL, a peer that never arrived)."""


def torch_allgather(dist, group=None):
    """Out-of-band exchange for Communicator over a torch.distributed process group (any back end)."""
    def exchange(payload):
        world = dist.get_world_size(group)
        out = [None] * world
        dist.all_gather_object(out, payload, group=group)
        return out
    return exchange


class Communicator:
    """sbamd_comm_t: the ranks of one node, one process per GPU (include/sbamd_dist.h).

    handle     the rank's SB_Handle (its device and stream; must outlive the communicator)
    allgather  callable(bytes) -> list of every rank's bytes in rank order: the out-of-band exchange used
               while the communicator and its windows are set up (torch_allgather(dist) makes one)
    transports _lib.DIST_MESH | _lib.DIST_RCCL
    Every method is collective: all ranks call it, same arguments apart from pointers, same order.
    """

    _SUFFIX = {"float32": "s", "float64": "d"}

    def __init__(self, handle, rank, world, allgather=None, transports=_lib.DIST_MESH, workspace_bytes=0):
        self._c = None
        self.handle, self.rank, self.world, self.transports = handle, int(rank), int(world), int(transports)
        self._exchange = allgather
        self._failure = None

        def callback(_ctx, send, recv, nbytes):
            try:
                parts = self._exchange(C.string_at(send, nbytes))
                if len(parts) != self.world or any(len(p) != nbytes for p in parts):
                    raise ValueError("the all-gather returned the wrong number or size of parts")
                for i, part in enumerate(parts):
                    C.memmove(recv + i * nbytes, part, nbytes)
                return 0
            except Exception as exc:  # never let an exception cross the C boundary
                self._failure = exc
                return 1

        self._cb = _lib.ALLGATHER_FN(callback)  # kept alive with the object
        raw = C.c_void_p()
        st = _lib.load().sbamd_dist_create(C.byref(raw), handle._h, self.rank, self.world, self._cb, None, self.transports,
                                           int(workspace_bytes))
        self._check(st, "sbamd_dist_create")
        self._c = raw

    def _check(self, st, what):
        if st == _lib.OK:
            return
        detail = _lib.load().sbamd_last_error(self.handle._h).decode() if self.handle._h else ""
        if self._failure is not None:
            detail += f" ({self._failure!r})"
        raise DistError(f"{what}: {_lib.status_string(st)} [{st}] {detail}")

    def close(self):
        if self._c:
            _lib.load().sbamd_dist_destroy(self._c)
            self._c = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def register(self, buf, nbytes=None):
        """Make a device buffer (DeviceBuffer or address) writable by every peer; returns the window id."""
        ptr = getattr(buf, "ptr", buf)
        if nbytes is None:
            nbytes = buf.count * buf.itemsize
        win = C.c_int(-1)
        self._check(_lib.load().sbamd_dist_register(self._c, C.c_void_p(int(ptr)), int(nbytes), C.byref(win)), "sbamd_dist_register")
        return win.value

    def unregister(self, window):
        self._check(_lib.load().sbamd_dist_unregister(self._c, int(window)), "sbamd_dist_unregister")

    def barrier(self):
        self._check(_lib.load().sbamd_dist_barrier(self._c), "sbamd_dist_barrier")

    def status(self):
        """Raises if a device-side wait of this communicator has timed out (synchronises the stream)."""
        self._check(_lib.load().sbamd_dist_status(self._c), "sbamd_dist_status")

    def block_gemm(self, pr, pc, transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc):
        """This rank's C block of the global product; a, b, c address the FULL operands.  No exchange."""
        suffix = "bs" if a.dtype == "bfloat16" else self._SUFFIX[c.dtype]
        sc = C.c_double if suffix == "d" else C.c_float
        fn = getattr(_lib.load(), f"sbamd_dist_block_{suffix}gemm")
        self._check(fn(self._c, int(pr), int(pc), transa.encode(), transb.encode(), m, n, k, sc(alpha), a.ptr, lda, b.ptr, ldb,
                       sc(beta), c.ptr, ldc), "sbamd_dist_block_gemm")

    def all_gather_c(self, pr, pc, m, n, c, ldc, window=-1):
        """Assemble the full C on every rank from the Pr x Pc blocks, in place, nothing re-laid out on the host."""
        self._check(_lib.load().sbamd_dist_all_gather_c(self._c, int(pr), int(pc), m, n, c.ptr, ldc, c.itemsize, int(window)),
                    "sbamd_dist_all_gather_c")

    def gathered_gemm(self, transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, window=-1, bands=0):
        """Strong scaling with the whole C on every rank (multiplication and traffic overlapped)."""
        suffix = self._SUFFIX[c.dtype]
        sc = C.c_double if suffix == "d" else C.c_float
        fn = getattr(_lib.load(), f"sbamd_dist_gathered_{suffix}gemm")
        self._check(fn(self._c, transa.encode(), transb.encode(), m, n, k, sc(alpha), a.ptr, lda, b.ptr, ldb, sc(beta), c.ptr, ldc,
                       int(window), int(bands)), "sbamd_dist_gathered_gemm")

    def splitk_gemm(self, transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, window=-1, gather=False):
        """K cut across the ranks; rank r ends with the columns split_range(n, world, r) (all of C with gather)."""
        suffix = self._SUFFIX[c.dtype]
        sc = C.c_double if suffix == "d" else C.c_float
        fn = getattr(_lib.load(), f"sbamd_dist_splitk_{suffix}gemm")
        self._check(fn(self._c, transa.encode(), transb.encode(), m, n, k, sc(alpha), a.ptr, lda, b.ptr, ldb, sc(beta), c.ptr, ldc,
                       int(window), int(bool(gather))), "sbamd_dist_splitk_gemm")

    def last_bands(self):
        """[(col_begin, col_end, depth, k_chunk), ...]: this rank's multiplications in the last gathered call."""
        arrays = [(C.c_int64 * _lib.DIST_MAX_BANDS)() for _ in range(4)]
        count = _lib.load().sbamd_dist_last_bands(self._c, *arrays)
        return [tuple(arr[i] for arr in arrays) for i in range(count)]
