"""Multi-GPU GEMM: 2-D C-block partition over the GPUs of one node.

The reference has no multi-device code (one sycl::queue per SB_Handle,
include/sb_handle/portblas_handle.h:46-60), so this layer is new.  GEMM shards
naturally: C blocks are independent.  One process drives one GPU; the ranks form
a Pr x Pc grid, rank (i, j) owns C block (i, j) and needs only the A row-panel i
and the B column-panel j, so with the panels resident the block products need NO
exchange.  Collectives (RCCL through torch.distributed) appear only where data
really moves:

  * all_gather_c      -- assemble the full C on every rank (optional);
  * splitk_gemm       -- K is cut across ranks instead, every rank produces a
                         full-size partial C, and a reduce-scatter sums them and
                         leaves each rank one column slab of the result.

Every function takes the local GEMM as a callable, so the partition and the
collectives can be exercised on CPU ranks (gloo) without the HIP library.
"""
import math

GRIDS = {1: (1, 1), 2: (1, 2), 4: (2, 2), 8: (2, 4)}


def grid_for(world):
    """Process grid Pr x Pc (Pr <= Pc) for `world` ranks."""
    if world in GRIDS:
        return GRIDS[world]
    pr = int(math.isqrt(world))
    while world % pr:
        pr -= 1
    return pr, world // pr


def split_range(extent, parts, index):
    """Balanced [begin, end) of piece `index` when `extent` is cut into `parts`."""
    base, rem = divmod(extent, parts)
    begin = index * base + min(index, rem)
    return begin, begin + base + (1 if index < rem else 0)


def coords_of(rank, pr, pc):
    """Grid coordinates (i, j) of a rank: row-major over the Pr x Pc grid."""
    if not 0 <= rank < pr * pc:
        raise ValueError("rank outside the process grid")
    return rank // pc, rank % pc


def block_of(rank, pr, pc, m, n):
    """((row_begin, row_end), (col_begin, col_end)) of the C block a rank owns."""
    i, j = coords_of(rank, pr, pc)
    return split_range(m, pr, i), split_range(n, pc, j)


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
