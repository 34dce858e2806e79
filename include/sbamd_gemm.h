/*
 * sbamd_gemm.h -- C ABI of the MI355X (gfx950) GEMM engine that stands in for
 * the BLAS-3 GEMM path of portBLAS / sycl-blas.
 *
 * Every entry point is `extern "C"`, takes plain pointers and 64-bit sizes and
 * returns an `int` status (0 = success); nothing throws.  Device pointers are
 * raw HIP device addresses (the reference's `BufferIterator + offset` /
 * `usm_ptr + offset` collapse into a pointer that already includes the element
 * offset).  All matrices are column-major.  Calls are asynchronous on the
 * handle's HIP stream, like the reference's calls on its `sycl::queue`.
 *
 * Reference interfaces replaced (paths relative to the reference tree):
 *   include/interface/blas3_interface.h:86-123   blas::_gemm / _gemm_batched /
 *                                                _gemm_strided_batched
 *   src/interface/gemm_interface.hpp:102-240     argument rules (alpha==0,
 *                                                trans flags, stride checks)
 *   src/interface/blas3/gemm.cpp.in:33-137       the instantiated type pairs
 *   include/sb_handle/portblas_handle.h:46-200   SB_Handle (queue, wait)
 *   include/portblas_helper.h:54-67,139-191      allocate / copy helpers
 */
#ifndef SBAMD_GEMM_H
#define SBAMD_GEMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */

/* Status codes.  The three INVALID_TRANS / INVALID_STRIDE groups correspond
 * one-to-one to the std::invalid_argument messages of
 * src/interface/gemm_interface.hpp:144-165. */
enum {
  SBAMD_OK = 0,
  SBAMD_INVALID_TRANSA = 1,  /* "invalid _TransA"  gemm_interface.hpp:145 */
  SBAMD_INVALID_TRANSB = 2,  /* "invalid _TransB"  gemm_interface.hpp:147 */
  SBAMD_INVALID_STRIDEC = 3, /* "invalid _stridec" gemm_interface.hpp:160 */
  SBAMD_INVALID_STRIDEA = 4, /* "invalid _stridea" gemm_interface.hpp:162 */
  SBAMD_INVALID_STRIDEB = 5, /* "invalid _strideb" gemm_interface.hpp:164 */
  SBAMD_INVALID_HANDLE = 6,
  SBAMD_INVALID_VALUE = 7,   /* negative size, null pointer with work to do */
  SBAMD_HIP_ERROR = 8,       /* a HIP runtime call failed; see last_error */
  SBAMD_ALLOC_FAILED = 9,
  SBAMD_NOT_SUPPORTED = 10,
  SBAMD_INVALID_UPLO = 11,       /* "invalid _uplo"  symm_interface.hpp:52 */
  SBAMD_INVALID_SIDE = 12,       /* "invalid _side"  symm_interface.hpp:71 */
  SBAMD_TRSM_INVALID_SIZE = 13,  /* "invalid matrix size argument" trsm_interface.hpp:115 */
  SBAMD_TRSM_INVALID_SIDE = 14,  /* "invalid Side argument"        trsm_interface.hpp:124 */
  SBAMD_TRSM_INVALID_UPLO = 15,  /* "invalid Triangle argument"    trsm_interface.hpp:126 */
  SBAMD_TRSM_INVALID_TRANS = 16, /* "invalid Transpose argument"   trsm_interface.hpp:128 */
  SBAMD_TRSM_INVALID_DIAG = 17,  /* "invalid Diagonal argument"    trsm_interface.hpp:130 */
  SBAMD_SCRATCH_TOO_SMALL = 18   /* sbamd_sgemm_noalloc only: a pinned split-K cut whose cube exceeds the fixed array */
};

/* Batch layouts: include/operations/blas3_trees.h:59 (gemm_batch_type_t). */
enum { SBAMD_BATCH_STRIDED = 0, SBAMD_BATCH_INTERLEAVED = 1 };

/* Kernel families the dispatcher can pick; sbamd_set_kernel() pins one (tests
 * and tuning), SBAMD_KERNEL_AUTO restores the shape-class heuristic that
 * stands in for src/interface/blas3/backend/amd_gpu.hpp:33-290. */
enum {
  SBAMD_KERNEL_AUTO = 0,
  SBAMD_KERNEL_LARGE = 1,    /* 128x128 macro tile, LDS double buffer      */
  SBAMD_KERNEL_SMALL = 2,    /* 64x64 macro tile (batched / small shapes)  */
  SBAMD_KERNEL_SPLITK = 3,   /* tall-skinny: K slabs -> cube -> reduction  */
  SBAMD_KERNEL_NAIVE = 4,    /* one thread per C element, sequential k (chosen by itself for batches of items up to 16x16: a 4x4 block of C per thread) */
  SBAMD_KERNEL_HUGE = 5,     /* 256x256 (float) / 256x128 (double) macro tile */
  SBAMD_KERNEL_INTERLEAVED = 6, /* batch-contiguous layout; reported, not pinnable */
  SBAMD_KERNEL_TINY = 7,     /* 32x32 macro tile, one wavefront (batches of small matrices) */
  SBAMD_KERNEL_STREAM = 8,   /* long batches of 64x64x64-class items, or the 64x64 tiles of one product with K = 64, on resident workgroups; reported, not pinnable */
  SBAMD_KERNEL_MEDIUM = 9    /* 96x96 macro tile (parameter bit 0: 192x192) on the 16x16x4 matrix instructions (float, double; the 16-bit pairs choose by themselves when it is pinned) */
};

typedef struct sbamd_handle_s* sbamd_handle_t;
typedef struct sbamd_event_s* sbamd_event_t;

/* Human-readable text for a status code (static storage). */
const char* sbamd_status_string(int status);

/* ------------------------------------------------------------------ handle */

/* Replaces blas::SB_Handle(queue) (include/sb_handle/portblas_handle.h:51-60).
 * `device` is the HIP device ordinal.  `stream` is a hipStream_t to enqueue on,
 * or NULL to let the handle create (and own) a non-blocking stream. */
int sbamd_create(sbamd_handle_t* out, int device, void* stream);
int sbamd_destroy(sbamd_handle_t h);

/* SB_Handle::wait() (portblas_handle.h:175): drain the handle's stream. */
int sbamd_wait(sbamd_handle_t h);

/* SB_Handle::get_num_compute_units() (portblas_handle.h:173). */
int sbamd_get_num_compute_units(sbamd_handle_t h, int* out);

/* The hipStream_t the handle enqueues on (SB_Handle::get_queue analogue). */
int sbamd_get_stream(sbamd_handle_t h, void** out_stream);

/* Text of the most recent HIP failure seen through this handle. */
const char* sbamd_last_error(sbamd_handle_t h);

/* Pin a kernel family (SBAMD_KERNEL_*); `param` is family specific.
 * Split-K: bits 0-15 = cube depth (0 = derive from the compute-unit count as
 * src/operations/blas3/gemm_partial_local.hpp:191-199 does), bits 16-23 = tile of
 * the partial products (0 = automatic, 1 = 64x64, 2 = 128x128, 3 = the biggest tile, 6 = 96x96,
 * 7 = 192x192; 4 = 32x32 together with bit 24 only), bit 24 = keep the slabs inside ONE workgroup
 * instead of going through the cube: both halves of K (depth 2) on the 64x64 / 128x128 / 96x96 tile,
 * the four quarters of K (depth 4) on the 32x32 tile (same result as the cube route with two / four
 * slabs; needs K to split into equal parts whose start stays 16-byte aligned, else the cube route runs).
 * SBAMD_KERNEL_SMALL: bit 0 = keep the register-staged kernel where the direct-to-LDS one would run
 * (fp32, A 'n', B 't'; same bits either way -- a switch for A/B timing).
 * 16-bit entry points: SBAMD_KERNEL_HUGE / SBAMD_KERNEL_LARGE are the 256x256 / 128x128 tiles of the
 * direct-to-LDS kernel where the operands allow it (16-byte aligned, leading dimensions multiples of 8, K >= 32,
 * M / N a multiple of 8 along a contiguous operand), else HUGE falls back to the automatic choice and LARGE to
 * the register-staged 128x128 kernel; SBAMD_KERNEL_LARGE bit 0 = keep the register-staged kernel anyway;
 * SBAMD_KERNEL_HUGE bit 1 = eight wavefronts of 128x64 per 256x256 tile instead of four of 128x128;
 * split-K bits 16-23 == 1 = K slabs on the register-staged 64x64 tile instead of the direct-to-LDS 128x128 one.
 * SBAMD_KERNEL_AUTO bit 0 = always copy operands the direct-to-LDS kernel cannot take as they lie (odd leading
 * dimension or offset, M / N not a multiple of 8) into padded aligned buffers, where the library would otherwise
 * do so only for products big enough to pay for the copies.
 * Every 16-bit kernel produces the same bits for the same K cut. */
int sbamd_set_kernel(sbamd_handle_t h, int kernel, int param);

/* Which family the last GEMM call on this handle launched (SBAMD_KERNEL_*),
 * -1 when it launched none (alpha == 0, empty problem). */
int sbamd_last_kernel(sbamd_handle_t h);

/* How the last split-K call cut K: `depth` slabs of `k_chunk` (the last one may
 * be shorter).  depth == 1 when the last call did not split.  The cut decides
 * the rounding of a split-K result (the reference's cut depends on the device's
 * compute-unit count the same way, gemm_partial_local.hpp:174, :191-199), so a
 * checker needs it to reproduce the result exactly. */
int sbamd_last_splitk(sbamd_handle_t h, int64_t* depth, int64_t* k_chunk);

/* The macro tile (rows x columns of C per workgroup) the last sbamd_sgemm / sbamd_dgemm family call ran its
 * matrix-instruction kernel on, and whether a cut in two was done inside one workgroup (`fused`, no cube).
 * 0 x 0 when the call launched no such kernel (the one-element-per-lane and interleaved kernels, alpha == 0) and
 * after the 16-bit entry points.  A tuning hook like sbamd_last_kernel (the reference reports its configuration
 * as a string, gemm_local.hpp:161-178 get_type_string); it does not affect results. */
int sbamd_last_tile(sbamd_handle_t h, int* tile_m, int* tile_n, int* fused);

/* How the last _trsm call on this handle was carried out: `diag` = launches of the
 * kernel that solves a whole diagonal super-block (up to 128 rows) at once, `panel` =
 * launches that apply the updates of many 16-wide blocks to a panel in one pass.  Both
 * are 0 when the call went block by block, GEMM call for GEMM call, as
 * src/interface/trsm_interface.hpp:227-380 does.  The result is the same bit for bit
 * either way; tests use the counts to know which route they exercised. */
int sbamd_last_trsm_launches(sbamd_handle_t h, int64_t* diag, int64_t* panel);

/* ------------------------------------------------------------------ events */

/* The reference returns std::vector<sycl::event> from every call and takes a
 * dependency list (portblas_handle.h:50, kernel_constructor.hpp:199).  Here an
 * event is recorded on the handle's stream after a call, and a dependency is
 * a stream-side wait on such an event. */
int sbamd_event_record(sbamd_handle_t h, sbamd_event_t* out);
int sbamd_stream_wait_event(sbamd_handle_t h, sbamd_event_t ev);
int sbamd_event_synchronize(sbamd_event_t ev);      /* SB_Handle::wait(ev)  */
int sbamd_event_elapsed_ms(sbamd_event_t start, sbamd_event_t stop, float* ms);
int sbamd_event_destroy(sbamd_event_t ev);

/* ------------------------------------------------------------ memory helpers */

/* blas::helper::allocate / deallocate / copy_to_device / copy_to_host
 * (include/portblas_helper.h:54-67, :139-191).  Copies are asynchronous on the
 * handle's stream; sbamd_wait() or an event makes them visible. */
int sbamd_malloc(sbamd_handle_t h, void** out, size_t bytes);
int sbamd_free(sbamd_handle_t h, void* ptr);
/* The same release without a handle (waits for the device's outstanding work, then frees):
 * for owners that may outlive the handle they were allocated through, as a sycl::buffer
 * outlives the SB_Handle in the reference (include/container/sycl_iterator.h:179-181). */
int sbamd_device_free(int device, void* ptr);
int sbamd_copy_to_device(sbamd_handle_t h, void* dst_dev, const void* src_host, size_t bytes);
int sbamd_copy_to_host(sbamd_handle_t h, void* dst_host, const void* src_dev, size_t bytes);
int sbamd_memset(sbamd_handle_t h, void* dst_dev, int byte_value, size_t bytes);

/* Fill a device buffer with uniform reals in [lo, hi) from a counter-based
 * generator (element i depends only on seed and i), so multi-GiB operands are
 * produced without crossing PCIe.  dtype: 0 f32, 1 f64, 2 f16, 3 bf16.
 * The reference's tests and benchmarks fill with uniform [-2, 5]
 * (test/blas_test.hpp:137-140). */
int sbamd_fill_uniform(sbamd_handle_t h, void* dst_dev, int64_t count, int dtype,
                       uint64_t seed, double lo, double hi);

/* -------------------------------------------------------------------- GEMM */

/* C <- alpha * op(A) * op(B) + beta * C, column-major.
 * op(X) = X for trans 'n'/'N', X^T for 't'/'T'/'c'/'C'; any other character is
 * SBAMD_INVALID_TRANSA/B.  op(A) is m x k, op(B) is k x n, C is m x n.
 * alpha == 0: C <- beta * C and A, B are not read (checked before the trans
 * flags, as in gemm_interface.hpp:112-139).  beta == 0: C is never read.
 *
 * Replaces blas::_gemm (blas3_interface.h:86-95). Type pairs as instantiated
 * by src/interface/blas3/gemm.cpp.in:33-137:
 *   s: float  -> float      d: double -> double
 *   h: half   -> half       hs: half  -> float
 * plus bs: bfloat16 -> float, which the reference reaches only through its
 * joint_matrix kernel on float storage
 * (src/operations/blas3/gemm_load_store_joint_matrix.hpp:76-81). */
int sbamd_sgemm(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n, int64_t k,
                float alpha, const float* a, int64_t lda, const float* b, int64_t ldb,
                float beta, float* c, int64_t ldc);
int sbamd_dgemm(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n, int64_t k,
                double alpha, const double* a, int64_t lda, const double* b, int64_t ldb,
                double beta, double* c, int64_t ldc);
/* half storage is passed as uint16_t bit patterns (IEEE binary16 / bfloat16);
 * alpha and beta travel as float and are rounded to the output type. */
int sbamd_hgemm(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n, int64_t k,
                float alpha, const uint16_t* a, int64_t lda, const uint16_t* b, int64_t ldb,
                float beta, uint16_t* c, int64_t ldc);
int sbamd_hsgemm(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n, int64_t k,
                 float alpha, const uint16_t* a, int64_t lda, const uint16_t* b, int64_t ldb,
                 float beta, float* c, int64_t ldc);
int sbamd_bsgemm(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n, int64_t k,
                 float alpha, const uint16_t* a, int64_t lda, const uint16_t* b, int64_t ldb,
                 float beta, float* c, int64_t ldc);

/* sbamd_sgemm for callers that cannot allocate device memory (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has
 * no counterpart in the reference.  Same arguments and same kernels as
 * sbamd_sgemm; the only difference is where a split-K call keeps its partial
 * products: in a fixed array of SBAMD_NOALLOC_SCRATCH_BYTES that every GPU the
 * library runs on holds until the process ends, instead of the handle's
 * on-demand buffer.  The call makes no hipMalloc / hipFree and does not use or
 * touch the handle's own buffers.  The planner only considers K cuts whose cube
 * (m * n * depth floats) fits the array, so a shape whose preferred cut would
 * not fit runs with the best cut that does, or as a single launch; the result
 * equals sbamd_sgemm's bit for bit whenever both cut K the same way
 * (sbamd_last_splitk reports the cut of either).  Only a cut pinned with
 * sbamd_set_kernel(SBAMD_KERNEL_SPLITK, depth) that does not fit is refused:
 * SBAMD_SCRATCH_TOO_SMALL, nothing launched, C untouched.  All handles on one
 * GPU share that GPU's array: calls through this entry point that overlap in
 * time on one GPU (different streams or host threads) race on it; calls on one
 * stream are ordered and safe.  Nothing in the array has to start at zero (every
 * partial product is written before the reduction reads it), so it is never
 * cleared and keeps the last call's contents. */
#define SBAMD_NOALLOC_SCRATCH_BYTES ((size_t)64 << 20)
int sbamd_sgemm_noalloc(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n,
                        int64_t k, float alpha, const float* a, int64_t lda, const float* b,
                        int64_t ldb, float beta, float* c, int64_t ldc);

/* Replaces blas::_gemm_batched (blas3_interface.h:97-109).  For
 * SBAMD_BATCH_STRIDED the strides default to whole matrices
 * (gemm_interface.hpp:213-219); for SBAMD_BATCH_INTERLEAVED element (r, c) of
 * item b lives at c*ld*batch + r*batch + b
 * (test/unittest/blas3/blas3_gemm_common.hpp:54-69). */
int sbamd_sgemm_batched(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n,
                        int64_t k, float alpha, const float* a, int64_t lda, const float* b,
                        int64_t ldb, float beta, float* c, int64_t ldc, int64_t batch,
                        int batch_type);
int sbamd_dgemm_batched(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n,
                        int64_t k, double alpha, const double* a, int64_t lda, const double* b,
                        int64_t ldb, double beta, double* c, int64_t ldc, int64_t batch,
                        int batch_type);

/* The 16-bit pairs of blas::_gemm_batched (instantiated for every type pair incl. half:
 * src/interface/blas3/gemm.cpp.in:56-79; half dispatch src/interface/blas3/backend/default.hpp:150-199),
 * plus bfloat16 -> float.  Both batch layouts. */
int sbamd_hgemm_batched(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n,
                        int64_t k, float alpha, const uint16_t* a, int64_t lda, const uint16_t* b,
                        int64_t ldb, float beta, uint16_t* c, int64_t ldc, int64_t batch,
                        int batch_type);
int sbamd_hsgemm_batched(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n,
                         int64_t k, float alpha, const uint16_t* a, int64_t lda, const uint16_t* b,
                         int64_t ldb, float beta, float* c, int64_t ldc, int64_t batch,
                         int batch_type);
int sbamd_bsgemm_batched(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n,
                         int64_t k, float alpha, const uint16_t* a, int64_t lda, const uint16_t* b,
                         int64_t ldb, float beta, float* c, int64_t ldc, int64_t batch,
                         int batch_type);

/* Replaces blas::_gemm_strided_batched (blas3_interface.h:111-123).  Item b
 * uses A + b*stride_a, B + b*stride_b, C + b*stride_c.  For batch > 1:
 * stride_c < ldc*n or any negative stride is an error
 * (gemm_interface.hpp:158-166); stride_a == 0 / stride_b == 0 broadcast. */
int sbamd_sgemm_strided_batched(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n,
                                int64_t k, float alpha, const float* a, int64_t lda,
                                int64_t stride_a, const float* b, int64_t ldb, int64_t stride_b,
                                float beta, float* c, int64_t ldc, int64_t stride_c,
                                int64_t batch);
int sbamd_dgemm_strided_batched(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n,
                                int64_t k, double alpha, const double* a, int64_t lda,
                                int64_t stride_a, const double* b, int64_t ldb, int64_t stride_b,
                                double beta, double* c, int64_t ldc, int64_t stride_c,
                                int64_t batch);
int sbamd_hgemm_strided_batched(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n,
                                int64_t k, float alpha, const uint16_t* a, int64_t lda,
                                int64_t stride_a, const uint16_t* b, int64_t ldb,
                                int64_t stride_b, float beta, uint16_t* c, int64_t ldc,
                                int64_t stride_c, int64_t batch);
int sbamd_hsgemm_strided_batched(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n,
                                 int64_t k, float alpha, const uint16_t* a, int64_t lda,
                                 int64_t stride_a, const uint16_t* b, int64_t ldb,
                                 int64_t stride_b, float beta, float* c, int64_t ldc,
                                 int64_t stride_c, int64_t batch);
int sbamd_bsgemm_strided_batched(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n,
                                 int64_t k, float alpha, const uint16_t* a, int64_t lda,
                                 int64_t stride_a, const uint16_t* b, int64_t ldb,
                                 int64_t stride_b, float beta, float* c, int64_t ldc,
                                 int64_t stride_c, int64_t batch);

/* ------------------------------------------------- callers of the GEMM path */

/* Replaces blas::_symm (include/interface/blas3_interface.h:136-145,
 * src/interface/symm_interface.hpp:34-73).  side 'l': C <- alpha*A*B + beta*C with
 * A m x m symmetric; side 'r': C <- alpha*B*A + beta*C with A n x n symmetric.
 * Only the `uplo` ('u' / 'l') triangle of A is read.  Errors: SBAMD_INVALID_UPLO,
 * then SBAMD_INVALID_SIDE (the reference's order); alpha == 0 scales C as GEMM does. */
int sbamd_ssymm(sbamd_handle_t h, char side, char uplo, int64_t m, int64_t n, float alpha,
                const float* a, int64_t lda, const float* b, int64_t ldb, float beta, float* c,
                int64_t ldc);
int sbamd_dsymm(sbamd_handle_t h, char side, char uplo, int64_t m, int64_t n, double alpha,
                const double* a, int64_t lda, const double* b, int64_t ldb, double beta, double* c,
                int64_t ldc);

/* Replaces blas::_trsm (include/interface/blas3_interface.h:125-134,
 * src/interface/trsm_interface.hpp:106-396): solves op(A)*X = alpha*B (side 'l') or
 * X*op(A) = alpha*B (side 'r') for X, which overwrites B (m x n).  A is k x k
 * triangular (k = m or n), `uplo` says which triangle is stored, diag 'u' assumes a
 * unit diagonal; the other triangle is never read.  Same algorithm as the reference:
 * 16x16 diagonal blocks are inverted, then two GEMM calls per block row/column. */
int sbamd_strsm(sbamd_handle_t h, char side, char uplo, char trans, char diag, int64_t m, int64_t n,
                float alpha, const float* a, int64_t lda, float* b, int64_t ldb);
int sbamd_dtrsm(sbamd_handle_t h, char side, char uplo, char trans, char diag, int64_t m, int64_t n,
                double alpha, const double* a, int64_t lda, double* b, int64_t ldb);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* SBAMD_GEMM_H */
