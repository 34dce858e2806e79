/*
 * sbamd_dist.h -- C ABI of the multi-GPU layer of the MI355X GEMM engine: one
 * process per GPU, the GPUs of one node, C blocks / column bands / K ranges dealt
 * to the ranks, output assembled over xGMI.
 *
 * The reference has no counterpart: a blas::SB_Handle owns exactly one
 * sycl::queue (include/sb_handle/portblas_handle.h:46-60, :192-196) and nothing in
 * the reference addresses a second device.  These entry points are what a caller
 * of the single-GPU ABI (include/sbamd_gemm.h, include/portblas_compat.hpp) adds
 * to run one GEMM on all GPUs of a node.
 *
 * Two transports, chosen at create time (both may be on):
 *   SBAMD_DIST_MESH  xGMI is a full mesh of point-to-point links, so data goes
 *                    straight from the producing GPU into the memory of each of
 *                    its peers (buffers are exchanged once as HIP IPC handles;
 *                    a copy kernel drives all links at once), ordered by flag
 *                    words in uncached device memory.  Nothing is re-laid out.
 *   SBAMD_DIST_RCCL  RCCL collectives (ncclAllGather / ncclReduceScatter) on a
 *                    second stream; librccl is loaded on demand.
 * Where both are available the mesh transport is used unless
 * SBAMD_DIST_PREFER=rccl is set in the environment.  Other environment switches,
 * read when a communicator is created: SBAMD_DIST_TIMEOUT_MS (how long a device-side
 * wait for a peer may last, default 20000), SBAMD_DIST_PUSH_BLOCKS (workgroups per peer
 * of the push kernel, default 16), SBAMD_DIST_FAULT=mute (fault injection for callers'
 * fallback paths: this rank never signals, so every mesh barrier times out).
 *
 * Every entry point is collective: all ranks of the communicator call it with the
 * same arguments (apart from pointers) in the same order.  Calls are asynchronous
 * on the stream of the handle the communicator was created on; on return from
 * sbamd_wait(handle) the rank's outputs are complete, including what its peers
 * wrote into them.
 */
#ifndef SBAMD_DIST_H
#define SBAMD_DIST_H

#include "sbamd_gemm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Additional status codes of this layer (continuing the enum of sbamd_gemm.h). */
enum {
  SBAMD_RCCL_ERROR = 19,       /* an RCCL call failed, or librccl could not be loaded */
  SBAMD_DIST_TIMEOUT = 20,     /* a peer did not arrive within SBAMD_DIST_TIMEOUT_MS */
  SBAMD_DIST_BOOTSTRAP = 21,   /* the caller's all-gather callback failed */
  SBAMD_DIST_NO_TRANSPORT = 22 /* the operation needs a transport / window this communicator lacks */
};

enum { SBAMD_DIST_MESH = 1, SBAMD_DIST_RCCL = 2 };
#define SBAMD_DIST_MAX_RANKS 16
#define SBAMD_DIST_MAX_BANDS 16

typedef struct sbamd_comm_s* sbamd_comm_t;

/* Out-of-band exchange used only while a communicator or a window is set up: every rank
 * contributes `bytes` from `send` and receives all ranks' contributions, in rank order, in
 * `recv` (world * bytes).  Return 0 on success.  (MPI_Allgather, a torch.distributed gloo
 * all-gather, a shared-memory segment -- whatever the launcher offers.) */
typedef int (*sbamd_allgather_fn)(void* ctx, const void* send, void* recv, size_t bytes);

/* Create the communicator of `world` ranks on the GPU and stream of `h` (which must outlive it).
 * `transports`: SBAMD_DIST_MESH | SBAMD_DIST_RCCL.  `workspace_bytes` of device memory are
 * allocated per rank and mapped into every peer (mesh: partial products of the K cut across
 * ranks; RCCL: staging of block gathers); 0 asks for none. */
int sbamd_dist_create(sbamd_comm_t* out, sbamd_handle_t h, int rank, int world, sbamd_allgather_fn allgather,
                      void* ctx, int transports, size_t workspace_bytes);
int sbamd_dist_destroy(sbamd_comm_t comm);
int sbamd_dist_rank(sbamd_comm_t comm, int* rank, int* world, int* transports);

/* Mesh transport: make `bytes` at `ptr` (device memory of this rank; the peers pass their own
 * buffer of the same size and role) writable by every peer.  Returns a window id >= 0 in *window.
 * The memory must stay allocated until sbamd_dist_unregister or sbamd_dist_destroy. */
int sbamd_dist_register(sbamd_comm_t comm, void* ptr, size_t bytes, int* window);
int sbamd_dist_unregister(sbamd_comm_t comm, int window);

/* Stream-ordered barrier over all ranks (work queued before it on every rank's stream has
 * completed before work queued after it on any rank's stream starts). */
int sbamd_dist_barrier(sbamd_comm_t comm);

/* 0 when no device-side wait of this communicator has timed out so far (synchronises the stream). */
int sbamd_dist_status(sbamd_comm_t comm);

/* ------------------------------------------------------------- partition */

/* Process grid Pr x Pc (Pr <= Pc) for `world` ranks: 1x1, 1x2, 2x2, 2x4, else the most square. */
void sbamd_dist_grid(int world, int* pr, int* pc);
/* Balanced [begin, end) of piece `index` when `extent` is cut into `parts`. */
void sbamd_dist_split_range(int64_t extent, int parts, int index, int64_t* begin, int64_t* end);
/* C block of a rank on the Pr x Pc grid (ranks row-major over the grid). */
int sbamd_dist_block_of(int rank, int pr, int pc, int64_t m, int64_t n, int64_t* r0, int64_t* r1, int64_t* c0,
                        int64_t* c1);
/* Column bands of one rank's column range [c0, c1) in sbamd_dist_gathered_*gemm: `bands` > 0 cuts it
 * into that many equal parts, `bands` == 0 chooses (halves that shrink towards the end, so that
 * only a small last band is sent with nothing left to hide behind).  Writes up to
 * SBAMD_DIST_MAX_BANDS widths and returns how many. */
int sbamd_dist_band_widths(int64_t cols, int bands, int world, int64_t* widths);

/* ------------------------------------------------------------------ GEMM */

/* Config "C stays distributed": this rank's C block of the global m x n x k product on a Pr x Pc
 * grid; a, b, c address the FULL column-major operands (replicated inputs; C written in place into
 * the rank's block).  No exchange.  s: float, d: double, bs: bfloat16 in / float out. */
int sbamd_dist_block_sgemm(sbamd_comm_t comm, int pr, int pc, char transa, char transb, int64_t m, int64_t n,
                           int64_t k, float alpha, const float* a, int64_t lda, const float* b, int64_t ldb,
                           float beta, float* c, int64_t ldc);
int sbamd_dist_block_dgemm(sbamd_comm_t comm, int pr, int pc, char transa, char transb, int64_t m, int64_t n,
                           int64_t k, double alpha, const double* a, int64_t lda, const double* b, int64_t ldb,
                           double beta, double* c, int64_t ldc);
int sbamd_dist_block_bsgemm(sbamd_comm_t comm, int pr, int pc, char transa, char transb, int64_t m, int64_t n,
                            int64_t k, float alpha, const uint16_t* a, int64_t lda, const uint16_t* b, int64_t ldb,
                            float beta, float* c, int64_t ldc);

/* Assemble the full C on every rank from the Pr x Pc blocks: `c` addresses the rank's FULL m x n
 * matrix (leading dimension ldc) whose own block is filled in.  `elem_bytes` 4 or 8.
 * Mesh: `window` is the window `c` lies in; each rank's block goes straight into its place in
 * every peer's C (strided, no staging).  RCCL (window < 0): column stripes of the blocks are
 * packed into the workspace, gathered with ncclAllGather and unpacked. */
int sbamd_dist_all_gather_c(sbamd_comm_t comm, int pr, int pc, int64_t m, int64_t n, void* c, int64_t ldc,
                            int elem_bytes, int window);

/* Strong scaling with the whole C on every rank: C <- alpha*op(A)*op(B) + beta*C, a, b, c the FULL
 * operands on every rank (with beta != 0 every rank holds the same C on entry).  The columns of C
 * are dealt to the ranks; each rank multiplies its columns band by band straight into their place
 * and band i travels to the peers while band i + 1 is multiplied.  `bands`: 0 = choose.
 * Mesh (`window` >= 0): rank r owns the columns split_range(n, world, r), cut into bands by
 * sbamd_dist_band_widths; the bands are pushed into the peers' windows.
 * RCCL (window < 0): band i covers columns [i*W, (i+1)*W), W = n / bands, rank r takes the r-th
 * part of it, and the band's ncclAllGather is in place (needs ldc == m and n % (bands*world) == 0). */
int sbamd_dist_gathered_sgemm(sbamd_comm_t comm, char transa, char transb, int64_t m, int64_t n, int64_t k,
                              float alpha, const float* a, int64_t lda, const float* b, int64_t ldb, float beta,
                              float* c, int64_t ldc, int window, int bands);
int sbamd_dist_gathered_dgemm(sbamd_comm_t comm, char transa, char transb, int64_t m, int64_t n, int64_t k,
                              double alpha, const double* a, int64_t lda, const double* b, int64_t ldb, double beta,
                              double* c, int64_t ldc, int window, int bands);

/* The multiplications of the last gathered call on this rank: column range, and how the library cut
 * K for each (depth slabs of k_chunk; depth 1 = one k-ordered chain) -- what a checker needs to
 * reproduce the result exactly.  Arrays of SBAMD_DIST_MAX_BANDS; returns the band count. */
int sbamd_dist_last_bands(sbamd_comm_t comm, int64_t* c0, int64_t* c1, int64_t* depth, int64_t* k_chunk);

/* K cut across the ranks (the multi-GPU form of the tall-skinny split): rank r multiplies the
 * K range split_range(k, world, r) into a full-size partial product in its workspace (needs
 * m*n elements of it), then the partials are summed in rank order -- the same order on every
 * rank, so the result does not depend on who adds -- and + beta*C is applied once:
 *   C[:, slab_r] <- sum_r alpha*op(A)[:, K_r]*op(B)[K_r, :] + beta*C[:, slab_r],  slab_r = split_range(n, world, r).
 * gather != 0 (mesh, window >= 0): every rank ends with all slabs, i.e. the whole C.
 * Mesh: each rank reads its slab of every peer's partial over xGMI.  RCCL: ncclReduceScatter
 * (needs n % world == 0; the order of the additions is RCCL's). */
int sbamd_dist_splitk_sgemm(sbamd_comm_t comm, char transa, char transb, int64_t m, int64_t n, int64_t k,
                            float alpha, const float* a, int64_t lda, const float* b, int64_t ldb, float beta,
                            float* c, int64_t ldc, int window, int gather);
int sbamd_dist_splitk_dgemm(sbamd_comm_t comm, char transa, char transb, int64_t m, int64_t n, int64_t k,
                            double alpha, const double* a, int64_t lda, const double* b, int64_t ldb, double beta,
                            double* c, int64_t ldc, int window, int gather);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* SBAMD_DIST_H */
