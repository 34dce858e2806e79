/*
 * gemm_oracle.c -- CPU restatement of the portBLAS / sycl-blas GEMM path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path may call, link or
 * load this file: it is the checker that the HIP kernels are compared with
 * (tests/, the smoke check, and the CPU-baseline leg of bench.py).
 *
 * PARITY UNPINNED in the strict sense: the reference holds no golden vectors
 * or known-answer files for GEMM (its tests draw random inputs and compare
 * with a system CBLAS within a tolerance, test/blas_test.hpp:107-140,
 * test/unittest/blas3/blas3_gemm_common.hpp:163-225), and the reference
 * itself cannot be built here (SYCL compiler and the py_gen submodule are
 * absent).  What pins this file instead is the reference's own method: it is
 * checked against OpenBLAS sgemm/dgemm on the reference's parameter grids with
 * the reference's pass/fail predicate (tests/test_oracle.py), and against
 * committed fixtures generated from it (tests/golden/).
 *
 * Each function cites the reference lines it follows (paths relative to the
 * reference tree).  Built with -ffp-contract=off so that only the explicit
 * fma() calls fuse.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Status codes shared with include/sbamd_gemm.h (kept numerically equal so
 * the front-end rules can be compared call by call). */
enum {
  ORC_OK = 0,
  ORC_INVALID_TRANSA = 1,
  ORC_INVALID_TRANSB = 2,
  ORC_INVALID_STRIDEC = 3,
  ORC_INVALID_STRIDEA = 4,
  ORC_INVALID_STRIDEB = 5,
  ORC_INVALID_VALUE = 7,
  ORC_ALLOC_FAILED = 9
};

/* Which arithmetic form of the reference to follow. */
enum {
  /* alpha*acc + beta*C, acc = fma chain over k from zero:
   * src/operations/blas3/gemm_ref.hpp:234-251 (naive kernel) */
  ORC_FORM_REF = 0,
  /* alpha * acc, acc = fma chain over k started at (beta/alpha)*C:
   * src/operations/blas3/gemm_local.hpp:182 (beta/alpha on the host),
   * :375-409 (scaling_c), :557-581 (store * alpha); the no_local kernels do
   * the same (gemm_no_local_full_vec.hpp:132, gemm_no_local_partial_vec.hpp:129) */
  ORC_FORM_LOCAL = 1,
  /* split-K: per K slab alpha*sum (gemm_partial_local.hpp:297-355, IsBetaZero
   * forced true by src/sb_handle/portblas_handle.hpp:345-350), slabs summed,
   * then + beta*C (portblas_handle.hpp:383-395).  Slabs are added in index
   * order. */
  ORC_FORM_SPLITK = 2,
  /* as ORC_FORM_SPLITK but slabs are combined pairwise, the shape of the
   * reference's local-memory tree (src/operations/extension/reduction.hpp:209-224) */
  ORC_FORM_SPLITK_TREE = 3
};

#define DEFINE_ORACLE(T, SUF, FMA)                                                              \
  /* op(A) packed as m x k column-major, op(B) as k x n column-major: element   */             \
  /* addressing of gemm_ref.hpp:227-243 ('n' walks lda per k, otherwise 1).     */             \
  static void pack_a_##SUF(int ta, int64_t m, int64_t k, const T* a, int64_t lda, T* out) {     \
    for (int64_t l = 0; l < k; ++l)                                                             \
      for (int64_t i = 0; i < m; ++i) out[i + l * m] = ta ? a[l + i * lda] : a[i + l * lda];    \
  }                                                                                             \
  static void pack_b_##SUF(int tb, int64_t k, int64_t n, const T* b, int64_t ldb, T* out) {     \
    for (int64_t j = 0; j < n; ++j)                                                             \
      for (int64_t l = 0; l < k; ++l) out[l + j * k] = tb ? b[j + l * ldb] : b[l + j * ldb];    \
  }                                                                                             \
                                                                                                \
  /* acc[i] <- fma chain over k in [k0, k1) for one column j, every row at once */             \
  /* (per element this is exactly the sequential-k mad loop of                  */             \
  /* gemm_ref.hpp:239-244 / gemm_local.hpp:752-772).                            */             \
  static void column_chain_##SUF(int64_t m, int64_t k0, int64_t k1, int64_t kld, const T* pa,   \
                                 const T* pbcol, T* acc) {                                      \
    (void)kld;                                                                                  \
    for (int64_t l = k0; l < k1; ++l) {                                                         \
      const T bv = pbcol[l];                                                                    \
      const T* acol = pa + l * m;                                                               \
      for (int64_t i = 0; i < m; ++i) acc[i] = FMA(acol[i], bv, acc[i]);                        \
    }                                                                                           \
  }                                                                                             \
                                                                                                \
  /* One GEMM in the chosen arithmetic form.  depth / k_chunk only matter for   */             \
  /* the split-K forms: slab s covers k in [s*k_chunk, min(k, (s+1)*k_chunk)).  */             \
  int oracle_gemm_##SUF(int form, int ta, int tb, int64_t m, int64_t n, int64_t k, T alpha,     \
                        const T* a, int64_t lda, const T* b, int64_t ldb, T beta, T* c,         \
                        int64_t ldc, int64_t depth, int64_t k_chunk) {                          \
    if (m <= 0 || n <= 0) return ORC_OK;                                                        \
    T* pa = (T*)malloc(sizeof(T) * (size_t)(m * (k > 0 ? k : 1)));                              \
    T* pb = (T*)malloc(sizeof(T) * (size_t)(n * (k > 0 ? k : 1)));                              \
    if (!pa || !pb) {                                                                           \
      free(pa);                                                                                 \
      free(pb);                                                                                 \
      return ORC_ALLOC_FAILED;                                                                  \
    }                                                                                           \
    pack_a_##SUF(ta, m, k, a, lda, pa);                                                         \
    pack_b_##SUF(tb, k, n, b, ldb, pb);                                                         \
    const int beta_zero = (beta == (T)0); /* gemm_interface.hpp:92 */                           \
    int status = ORC_OK;                                                                        \
    _Pragma("omp parallel for schedule(static)") for (int64_t j = 0; j < n; ++j) {              \
      T* acc = (T*)malloc(sizeof(T) * (size_t)m);                                               \
      T* part = NULL;                                                                           \
      T* ccol = c + j * ldc;                                                                    \
      const T* pbcol = pb + j * k;                                                              \
      if (!acc) {                                                                               \
        status = ORC_ALLOC_FAILED;                                                              \
        continue;                                                                               \
      }                                                                                         \
      if (form == ORC_FORM_REF) {                                                               \
        for (int64_t i = 0; i < m; ++i) acc[i] = (T)0;                                          \
        column_chain_##SUF(m, 0, k, k, pa, pbcol, acc);                                         \
        for (int64_t i = 0; i < m; ++i) {                                                       \
          /* gemm_ref.hpp:247-251: beta == 0 never reads C */                                   \
          if (beta_zero) {                                                                      \
            ccol[i] = alpha * acc[i];                                                           \
          } else {                                                                              \
            const T t0 = alpha * acc[i];                                                        \
            const T t1 = beta * ccol[i];                                                        \
            ccol[i] = t0 + t1;                                                                  \
          }                                                                                     \
        }                                                                                       \
      } else if (form == ORC_FORM_LOCAL) {                                                      \
        const T ratio = beta / alpha; /* gemm_local.hpp:182 */                                  \
        for (int64_t i = 0; i < m; ++i) acc[i] = beta_zero ? (T)0 : ratio * ccol[i];            \
        column_chain_##SUF(m, 0, k, k, pa, pbcol, acc);                                         \
        for (int64_t i = 0; i < m; ++i) ccol[i] = alpha * acc[i];                               \
      } else {                                                                                  \
        const int64_t nslab = depth > 0 ? depth : 1;                                            \
        part = (T*)malloc(sizeof(T) * (size_t)(m * nslab));                                     \
        if (!part) {                                                                            \
          status = ORC_ALLOC_FAILED;                                                            \
          free(acc);                                                                            \
          continue;                                                                             \
        }                                                                                       \
        for (int64_t s = 0; s < nslab; ++s) {                                                   \
          int64_t k0 = s * k_chunk, k1 = k0 + k_chunk;                                          \
          if (k0 > k) k0 = k;                                                                   \
          if (k1 > k) k1 = k;                                                                   \
          for (int64_t i = 0; i < m; ++i) acc[i] = (T)0;                                        \
          column_chain_##SUF(m, k0, k1, k, pa, pbcol, acc);                                     \
          for (int64_t i = 0; i < m; ++i) part[i + s * m] = alpha * acc[i];                     \
        }                                                                                       \
        if (form == ORC_FORM_SPLITK) {                                                          \
          for (int64_t i = 0; i < m; ++i) {                                                     \
            T sum = part[i];                                                                    \
            for (int64_t s = 1; s < nslab; ++s) sum = sum + part[i + s * m];                    \
            acc[i] = sum;                                                                       \
          }                                                                                     \
        } else {                                                                                \
          for (int64_t w = 1; w < nslab; w *= 2)                                                \
            for (int64_t s = 0; s + w < nslab; s += 2 * w)                                      \
              for (int64_t i = 0; i < m; ++i) part[i + s * m] = part[i + s * m] + part[i + (s + w) * m]; \
          for (int64_t i = 0; i < m; ++i) acc[i] = part[i];                                     \
        }                                                                                       \
        for (int64_t i = 0; i < m; ++i) {                                                       \
          if (beta_zero) {                                                                      \
            ccol[i] = acc[i];                                                                   \
          } else {                                                                              \
            const T t1 = beta * ccol[i];                                                        \
            ccol[i] = acc[i] + t1;                                                              \
          }                                                                                     \
        }                                                                                       \
        free(part);                                                                             \
      }                                                                                         \
      free(acc);                                                                                \
    }                                                                                           \
    free(pa);                                                                                   \
    free(pb);                                                                                   \
    return status;                                                                              \
  }                                                                                             \
                                                                                                \
  /* C = beta*C over a rows x cols window: _scal_matrix multiplies               */            \
  /* (src/interface/blas1_interface.hpp:498-506), the vector _scal with a zero   */            \
  /* factor assigns zero (:444-448).                                             */            \
  static void scale_window_##SUF(T* c, int64_t rows, int64_t cols, int64_t ld, T beta,          \
                                 int assign_zero) {                                             \
    for (int64_t j = 0; j < cols; ++j)                                                          \
      for (int64_t i = 0; i < rows; ++i)                                                        \
        c[i + j * ld] = assign_zero ? (T)0 : beta * c[i + j * ld];                              \
  }                                                                                             \
                                                                                                \
  /* The whole front end: src/interface/gemm_interface.hpp:102-185.              */            \
  /* batch_type: 0 strided, 1 interleaved (element (r, c) of item b at           */            \
  /* c*ld*batch + r*batch + b, test/unittest/blas3/blas3_gemm_common.hpp:54-69). */            \
  int oracle_gemm_backend_##SUF(int form, char transa, char transb, int64_t m, int64_t n,       \
                                int64_t k, T alpha, const T* a, int64_t lda, int64_t stride_a,  \
                                const T* b, int64_t ldb, int64_t stride_b, T beta, T* c,        \
                                int64_t ldc, int64_t stride_c, int64_t batch, int batch_type,   \
                                int64_t depth, int64_t k_chunk) {                               \
    if (m < 0 || n < 0 || k < 0 || batch < 0) return ORC_INVALID_VALUE;                         \
    if (alpha == (T)0) { /* :112-139 */                                                         \
      if (m == 0 || n == 0 || batch == 0) return ORC_OK;                                        \
      const int64_t size_c = ldc * n;                                                           \
      if (batch_type == 1 && batch > 1) {                                                       \
        if (!(beta == (T)1)) scale_window_##SUF(c, m * batch, n, ldc * batch, beta, 0);         \
      } else if (size_c == stride_c) {                                                          \
        if (ldc == m) {                                                                         \
          scale_window_##SUF(c, size_c * batch, 1, size_c * batch, beta, beta == (T)0);         \
        } else if (!(beta == (T)1)) {                                                           \
          scale_window_##SUF(c, m, n * batch, ldc, beta, 0);                                    \
        }                                                                                       \
      } else if (!(beta == (T)1)) {                                                             \
        for (int64_t z = 0; z < batch; ++z) scale_window_##SUF(c + z * stride_c, m, n, ldc, beta, 0); \
      }                                                                                         \
      return ORC_OK;                                                                            \
    }                                                                                           \
    if (transa >= 'A' && transa <= 'Z') transa = (char)(transa - 'A' + 'a'); /* :141-142 */     \
    if (transb >= 'A' && transb <= 'Z') transb = (char)(transb - 'A' + 'a');                    \
    if (transa != 'n' && transa != 't' && transa != 'c') return ORC_INVALID_TRANSA;             \
    if (transb != 'n' && transb != 't' && transb != 'c') return ORC_INVALID_TRANSB;             \
    const int ta = transa != 'n', tb = transb != 'n'; /* :150-151 */                            \
    if (batch > 1 && batch_type == 0) {               /* :158-166 */                            \
      if (stride_c < ldc * n || stride_c < 0) return ORC_INVALID_STRIDEC;                       \
      if (stride_a < 0) return ORC_INVALID_STRIDEA;                                             \
      if (stride_b < 0) return ORC_INVALID_STRIDEB;                                             \
    }                                                                                           \
    if (m == 0 || n == 0 || batch == 0) return ORC_OK;                                          \
    if (batch_type == 1 && batch > 1) {                                                         \
      /* interleaved: item z is the strided problem with every index scaled by   */            \
      /* batch; gather it, solve it, scatter C back.                             */            \
      const int64_t ar = ta ? k : m, ac = ta ? m : k;                                           \
      const int64_t br = tb ? n : k, bc = tb ? k : n;                                           \
      T* ta_ = (T*)malloc(sizeof(T) * (size_t)(ar * ac > 0 ? ar * ac : 1));                     \
      T* tb_ = (T*)malloc(sizeof(T) * (size_t)(br * bc > 0 ? br * bc : 1));                     \
      T* tc_ = (T*)malloc(sizeof(T) * (size_t)(m * n));                                         \
      if (!ta_ || !tb_ || !tc_) {                                                               \
        free(ta_);                                                                              \
        free(tb_);                                                                              \
        free(tc_);                                                                              \
        return ORC_ALLOC_FAILED;                                                                \
      }                                                                                         \
      int st = ORC_OK;                                                                          \
      for (int64_t z = 0; z < batch && st == ORC_OK; ++z) {                                     \
        for (int64_t cc = 0; cc < ac; ++cc)                                                     \
          for (int64_t r = 0; r < ar; ++r) ta_[r + cc * ar] = a[cc * lda * batch + r * batch + z]; \
        for (int64_t cc = 0; cc < bc; ++cc)                                                     \
          for (int64_t r = 0; r < br; ++r) tb_[r + cc * br] = b[cc * ldb * batch + r * batch + z]; \
        for (int64_t cc = 0; cc < n; ++cc)                                                      \
          for (int64_t r = 0; r < m; ++r) tc_[r + cc * m] = c[cc * ldc * batch + r * batch + z]; \
        st = oracle_gemm_##SUF(form, ta, tb, m, n, k, alpha, ta_, ar, tb_, br, beta, tc_, m,    \
                               depth, k_chunk);                                                 \
        for (int64_t cc = 0; cc < n; ++cc)                                                      \
          for (int64_t r = 0; r < m; ++r) c[cc * ldc * batch + r * batch + z] = tc_[r + cc * m]; \
      }                                                                                         \
      free(ta_);                                                                                \
      free(tb_);                                                                                \
      free(tc_);                                                                                \
      return st;                                                                                \
    }                                                                                           \
    for (int64_t z = 0; z < batch; ++z) { /* gemm_local.hpp:290-292 */                          \
      int st = oracle_gemm_##SUF(form, ta, tb, m, n, k, alpha, a + z * stride_a, lda,           \
                                 b + z * stride_b, ldb, beta, c + z * stride_c, ldc, depth,     \
                                 k_chunk);                                                      \
      if (st != ORC_OK) return st;                                                              \
    }                                                                                           \
    return ORC_OK;                                                                              \
  }

DEFINE_ORACLE(float, f32, fmaf)
DEFINE_ORACLE(double, f64, fma)

/* The kernel the reference's DEFAULT (CPU) back end picks when M*N >= 524288, i.e. for
 * BASELINE config #1 (SGEMM 1024^3): src/interface/blas3/backend/default.hpp:98-112 --
 * Gemm<no_local, standard, partial> with Tile<4, 4, 4, 4>, vector size 1, no double buffer.
 * Restated loop for loop from src/operations/blas3/gemm_no_local_partial_vec.hpp:
 *   - work-groups of wg_rows*wg_cols = 16 items own block_rows x block_cols = 16 x 16 tiles of C,
 *     numbered row-tile first (:215-228);
 *   - item (item_id % wg_rows, item_id / wg_rows) holds an item_rows x item_cols = 4 x 4 register
 *     block whose rows / columns are wg_rows / wg_cols apart (:243-262, :321-333);
 *   - reg_res starts at (beta/alpha)*C, beta/alpha formed on the host (:129, :313-335), or 0 when
 *     beta == 0 (:340-347);
 *   - one k per trip: item_rows elements of A and item_cols of B straight from global memory
 *     (zero past the edge, :454-486), 16 mul_add (:374-401, :495-505);
 *   - C <- alpha * reg_res (:544-579).
 * The work-groups run as an OpenMP loop and the 16 items of a group one after the other, which
 * is how an OpenMP SYCL back end executes a barrier-free kernel.  This is the CPU baseline for
 * config #1: same arithmetic per element as ORC_FORM_LOCAL (asserted in the tests), the
 * reference's memory access pattern and blocking. */
#define DEFINE_DEFAULT_TILE(T, SUF, FMA)                                                        \
  int oracle_gemm_default_tile_##SUF(int ta, int tb, int64_t m, int64_t n, int64_t k, T alpha,  \
                                     const T* a, int64_t lda, const T* b, int64_t ldb, T beta,  \
                                     T* c, int64_t ldc) {                                       \
    enum { IR = 4, IC = 4, WR = 4, WC = 4, BR = IR * WR, BC = IC * WC };                        \
    if (m <= 0 || n <= 0) return ORC_OK;                                                        \
    const int beta_zero = (beta == (T)0);                                                       \
    const T ratio = beta / alpha; /* :129 */                                                    \
    const int64_t blocks_per_row = (m - 1) / BR + 1;                                            \
    const int64_t cluster = blocks_per_row * ((n - 1) / BC + 1); /* :153-156 */                 \
    const int64_t a_row = ta ? lda : 1, a_k = ta ? 1 : lda;      /* :260, :399 */               \
    const int64_t b_col = tb ? 1 : ldb, b_k = tb ? ldb : 1;      /* :261, :400 */               \
    _Pragma("omp parallel for schedule(static)") for (int64_t wg = 0; wg < cluster; ++wg) {     \
      const int64_t wg_row = (wg % blocks_per_row) * BR, wg_col = (wg / blocks_per_row) * BC;   \
      for (int item = 0; item < WR * WC; ++item) {                                              \
        const int64_t row0 = wg_row + item % WR, col0 = wg_col + item / WR;                     \
        if (row0 >= m || col0 >= n) continue; /* out_of_range, :250 */                          \
        T reg_res[IR * IC], reg_a[IR], reg_b[IC];                                               \
        for (int i = 0; i < IC; ++i)                                                            \
          for (int j = 0; j < IR; ++j) {                                                        \
            const int64_t r = row0 + j * WR, cc = col0 + i * WC;                                \
            reg_res[i * IR + j] = (!beta_zero && r < m && cc < n) ? ratio * c[r + cc * ldc] : (T)0; \
          }                                                                                     \
        const T* pa = a + row0 * a_row;                                                         \
        const T* pb = b + col0 * b_col;                                                         \
        for (int64_t l = 0; l < k; ++l) {                                                       \
          for (int j = 0; j < IR; ++j) reg_a[j] = (row0 + j * WR < m) ? pa[j * WR * a_row] : (T)0; \
          for (int i = 0; i < IC; ++i) reg_b[i] = (col0 + i * WC < n) ? pb[i * WC * b_col] : (T)0; \
          for (int i = 0; i < IC; ++i)                                                          \
            for (int j = 0; j < IR; ++j) reg_res[i * IR + j] = FMA(reg_a[j], reg_b[i], reg_res[i * IR + j]); \
          pa += a_k;                                                                            \
          pb += b_k;                                                                            \
        }                                                                                       \
        for (int i = 0; i < IC; ++i)                                                            \
          for (int j = 0; j < IR; ++j) {                                                        \
            const int64_t r = row0 + j * WR, cc = col0 + i * WC;                                \
            if (r < m && cc < n) c[r + cc * ldc] = alpha * reg_res[i * IR + j];                 \
          }                                                                                     \
      }                                                                                         \
    }                                                                                           \
    return ORC_OK;                                                                              \
  }

DEFINE_DEFAULT_TILE(float, f32, fmaf)
DEFINE_DEFAULT_TILE(double, f64, fma)

/* Status codes of the GEMM callers (numerically equal to include/sbamd_gemm.h). */
enum {
  ORC_INVALID_UPLO = 11,
  ORC_INVALID_SIDE = 12,
  ORC_TRSM_INVALID_SIZE = 13,
  ORC_TRSM_INVALID_SIDE = 14,
  ORC_TRSM_INVALID_UPLO = 15,
  ORC_TRSM_INVALID_TRANS = 16,
  ORC_TRSM_INVALID_DIAG = 17
};

static char lower_(char c) { return (c >= 'A' && c <= 'Z') ? (char)(c - 'A' + 'a') : c; }

#define DEFINE_CALLERS(T, SUF, FMA)                                                             \
  /* _symm: src/interface/symm_interface.hpp:34-73.  One GEMM whose symmetric    */            \
  /* operand is read through the stored triangle (gemm_local.hpp:813-873); the   */            \
  /* per-element arithmetic is the GEMM's k-ordered chain over the mirrored      */            \
  /* matrix, so the mirror is materialised and the GEMM front end called.        */            \
  /* depth / k_chunk: the K cut for the split-K forms (1 / k otherwise).         */            \
  int oracle_symm_##SUF(int form, char side, char uplo, int64_t m, int64_t n, T alpha,          \
                        const T* a, int64_t lda, const T* b, int64_t ldb, T beta, T* c,         \
                        int64_t ldc, int64_t depth, int64_t k_chunk) {                          \
    side = lower_(side);                                                                        \
    uplo = lower_(uplo);                                                                        \
    if (uplo != 'u' && uplo != 'l') return ORC_INVALID_UPLO; /* :51-53 */                       \
    if (side != 'l' && side != 'r') return ORC_INVALID_SIDE; /* :70-72 */                       \
    const int64_t k = side == 'l' ? m : n;                                                      \
    T* full = (T*)malloc(sizeof(T) * (size_t)(k * k > 0 ? k * k : 1));                          \
    if (!full) return ORC_ALLOC_FAILED;                                                         \
    if (!(alpha == (T)0))                                                                       \
      for (int64_t j = 0; j < k; ++j)                                                           \
        for (int64_t i = 0; i < k; ++i) {                                                       \
          const int stored = (uplo == 'u') ? (i <= j) : (i >= j);                               \
          full[i + j * k] = stored ? a[i + j * lda] : a[j + i * lda];                           \
        }                                                                                       \
    int st;                                                                                     \
    if (side == 'l') /* C <- alpha*A*B + beta*C, :54-59 */                                      \
      st = oracle_gemm_backend_##SUF(form, 'n', 'n', m, n, k, alpha, full, k, 0, b, ldb, 0,     \
                                     beta, c, ldc, 0, 1, 0, depth, k_chunk);                    \
    else /* C <- alpha*B*A + beta*C, :60-69 */                                                  \
      st = oracle_gemm_backend_##SUF(form, 'n', 'n', m, n, k, alpha, b, ldb, 0, full, k, 0,     \
                                     beta, c, ldc, 0, 1, 0, depth, k_chunk);                    \
    free(full);                                                                                 \
    return st;                                                                                  \
  }                                                                                             \
                                                                                                \
  /* Inverse of one 16x16 diagonal block of a triangular matrix:                 */            \
  /* src/operations/blas3/trsm.hpp:93-153.  The work-item index i of the         */            \
  /* reference becomes the inner loop; each barrier-separated phase reads only   */            \
  /* values written in earlier phases, so running the items in turn is exact.    */            \
  static void invert_block_##SUF(int unit, int upper, const T* a, int64_t lda, int64_t n,       \
                                 int64_t origin, T* dst) {                                      \
    enum { B = 16 };                                                                            \
    T local[B * B]; /* local[j + i*B] = (row i, column j) */                                    \
    T sum[B];                                                                                   \
    for (int i = 0; i < B; ++i)                                                                 \
      for (int j = 0; j < B; ++j) {                                                             \
        const int in_range = upper ? (i <= j) && (origin + j < n) : (i >= j) && (origin + i < n); \
        T v = (T)0;                                                                             \
        if (in_range) v = (unit && i == j) ? (T)1 : a[(origin + j) * lda + origin + i];         \
        local[j + i * B] = v;                                                                   \
      }                                                                                         \
    if (!unit)                                                                                  \
      for (int i = 0; i < B; ++i) local[i + i * B] = (T)1 / local[i + i * B];                   \
    if (upper) {                                                                                \
      for (int j = 1; j < B; ++j) {                                                             \
        for (int i = 0; i < j; ++i) {                                                           \
          T s = (T)0;                                                                           \
          for (int k = 0; k < j; ++k) s = FMA(local[k + i * B], local[j + k * B], s);           \
          sum[i] = s;                                                                           \
        }                                                                                       \
        for (int i = 0; i < j; ++i) local[j + i * B] = sum[i] * (unit ? (T)-1 : -local[j + j * B]); \
      }                                                                                         \
    } else {                                                                                    \
      for (int j = B - 2; j >= 0; --j) {                                                        \
        for (int i = j + 1; i < B; ++i) {                                                       \
          T s = (T)0;                                                                           \
          for (int k = j + 1; k < B; ++k) s = FMA(local[k + i * B], local[j + k * B], s);       \
          sum[i] = s;                                                                           \
        }                                                                                       \
        for (int i = j + 1; i < B; ++i) local[j + i * B] = sum[i] * (unit ? (T)-1 : -local[j + j * B]); \
      }                                                                                         \
    }                                                                                           \
    for (int j = 0; j < B; ++j)                                                                 \
      for (int i = 0; i < B; ++i) dst[j * B + i] = local[j + i * B];                            \
  }                                                                                             \
                                                                                                \
  /* _trsm: src/interface/trsm_interface.hpp:106-396, GEMM call for GEMM call.   */            \
  int oracle_trsm_##SUF(int form, char side, char uplo, char trans, char diag, int64_t M,       \
                        int64_t N, T alpha, const T* A, int64_t lda, T* B, int64_t ldb) {       \
    if (M == 0 || N == 0 || lda == 0 || ldb == 0) return ORC_TRSM_INVALID_SIZE; /* :114 */      \
    side = lower_(side);                                                                        \
    uplo = lower_(uplo);                                                                        \
    trans = lower_(trans);                                                                      \
    diag = lower_(diag);                                                                        \
    if (side != 'l' && side != 'r') return ORC_TRSM_INVALID_SIDE; /* :123-131 */                \
    if (uplo != 'u' && uplo != 'l') return ORC_TRSM_INVALID_UPLO;                               \
    if (trans != 'n' && trans != 't') return ORC_TRSM_INVALID_TRANS;                            \
    if (diag != 'u' && diag != 'n') return ORC_TRSM_INVALID_DIAG;                               \
    const int64_t K = side == 'l' ? M : N, bs = 16;                                             \
    const int unit = diag == 'u', upper = uplo == 'u', left = side == 'l', tr = trans == 't';   \
    const int64_t nblocks = (K + bs - 1) / bs;                                                  \
    const int64_t b_elems = ldb * (N - 1) + M; /* :199 */                                       \
    const int64_t ldx = ldb;                                                                    \
    T* invA = (T*)calloc((size_t)(nblocks * bs * bs), sizeof(T)); /* :148-155 */                \
    T* X = (T*)malloc(sizeof(T) * (size_t)b_elems);                                             \
    if (!invA || !X) {                                                                          \
      free(invA);                                                                               \
      free(X);                                                                                  \
      return ORC_ALLOC_FAILED;                                                                  \
    }                                                                                           \
    for (int64_t blk = 0; blk < nblocks; ++blk) /* :165-194 */                                  \
      invert_block_##SUF(unit, upper, A, lda, K, blk * bs, invA + blk * bs * bs);               \
    memcpy(X, B, sizeof(T) * (size_t)b_elems); /* :205-208 */                                   \
    const char tc = tr ? 't' : 'n';                                                             \
    const T one = (T)1, zero = (T)0, neg = (T)-1;                                               \
    int st = ORC_OK;                                                                            \
    /* every call is internal::_gemm: batch 1, strides 0 */                                     \
    if (left) {                                                                                 \
      if ((upper && tr) || (!upper && !tr)) { /* :227-251 */                                    \
        for (int64_t i = 0; i < M && st == ORC_OK; i += bs) {                                   \
          const int64_t cb = (M - i < bs) ? M - i : bs;                                         \
          st = oracle_gemm_backend_##SUF(form, tc, 'n', cb, N, cb, i == 0 ? alpha : one,        \
                                         invA + i * bs, bs, 0, B + i, ldb, 0, zero, X + i, ldx, \
                                         0, 1, 0, 1, cb);                                       \
          if (st != ORC_OK || i + bs >= M) break;                                               \
          const int64_t off = !tr ? ((i + bs) + i * lda) : (i + (bs + i) * lda);                \
          st = oracle_gemm_backend_##SUF(form, tc, 'n', M - i - bs, N, bs, neg, A + off, lda,   \
                                         0, X + i, ldx, 0, i == 0 ? alpha : one, B + i + bs,    \
                                         ldb, 0, 1, 0, 1, bs);                                  \
        }                                                                                       \
      } else { /* :267-292 */                                                                   \
        const int64_t special = (M % bs == 0) ? bs : (M % bs);                                  \
        const int64_t start = M - special;                                                      \
        for (int64_t i = start; i >= 0 && st == ORC_OK; i -= bs) {                              \
          const int64_t cb = (i == start) ? special : bs;                                       \
          st = oracle_gemm_backend_##SUF(form, tc, 'n', cb, N, cb, i == start ? alpha : one,    \
                                         invA + i * bs, bs, 0, B + i, ldb, 0, zero, X + i, ldx, \
                                         0, 1, 0, 1, cb);                                       \
          if (st != ORC_OK || i - bs < 0) break;                                                \
          st = oracle_gemm_backend_##SUF(form, tc, 'n', i, N, cb, neg, A + (!tr ? i * lda : i), \
                                         lda, 0, X + i, ldx, 0, i == start ? alpha : one, B,    \
                                         ldb, 0, 1, 0, 1, cb);                                  \
        }                                                                                       \
      }                                                                                         \
    } else {                                                                                    \
      if ((upper && tr) || (!upper && !tr)) { /* :312-337 */                                    \
        const int64_t special = (N % bs == 0) ? bs : (N % bs);                                  \
        const int64_t start = N - special;                                                      \
        for (int64_t i = start; i >= 0 && st == ORC_OK; i -= bs) {                              \
          const int64_t cb = (i == start) ? special : bs;                                       \
          st = oracle_gemm_backend_##SUF(form, 'n', tc, M, cb, cb, i == start ? alpha : one,    \
                                         B + i * ldb, ldb, 0, invA + i * bs, bs, 0, zero,       \
                                         X + i * ldx, ldx, 0, 1, 0, 1, cb);                     \
          if (st != ORC_OK || i - bs < 0) break;                                                \
          st = oracle_gemm_backend_##SUF(form, 'n', tc, M, i, cb, neg, X + i * ldx, ldx, 0,     \
                                         A + (!tr ? i : i * lda), lda, 0,                       \
                                         i == start ? alpha : one, B, ldb, 0, 1, 0, 1, cb);     \
        }                                                                                       \
      } else { /* :354-380 */                                                                   \
        for (int64_t i = 0; i < N && st == ORC_OK; i += bs) {                                   \
          const int64_t cb = (N - i < bs) ? N - i : bs;                                         \
          st = oracle_gemm_backend_##SUF(form, 'n', tc, M, cb, cb, i == 0 ? alpha : one,        \
                                         B + i * ldb, ldb, 0, invA + i * bs, bs, 0, zero,       \
                                         X + i * ldx, ldx, 0, 1, 0, 1, cb);                     \
          if (st != ORC_OK || i + bs > N) break;                                                \
          const int64_t off = !tr ? (i + (bs + i) * lda) : ((i + bs) + i * lda);                \
          st = oracle_gemm_backend_##SUF(form, 'n', tc, M, N - i - bs, bs, neg, X + i * ldx,    \
                                         ldx, 0, A + off, lda, 0, i == 0 ? alpha : one,         \
                                         B + (i + bs) * ldb, ldb, 0, 1, 0, 1, bs);              \
        }                                                                                       \
      }                                                                                         \
    }                                                                                           \
    if (st == ORC_OK) memcpy(B, X, sizeof(T) * (size_t)b_elems); /* :384-389 */                 \
    free(invA);                                                                                 \
    free(X);                                                                                    \
    return st;                                                                                  \
  }

DEFINE_CALLERS(float, f32, fmaf)
DEFINE_CALLERS(double, f64, fma)

/* Default strides of _gemm_batched (src/interface/gemm_interface.hpp:213-219). */
void oracle_default_strides(char transa, char transb, int64_t m, int64_t n, int64_t k, int64_t lda,
                            int64_t ldb, int64_t ldc, int64_t* sa, int64_t* sb, int64_t* sc) {
  if (transa >= 'A' && transa <= 'Z') transa = (char)(transa - 'A' + 'a');
  if (transb >= 'A' && transb <= 'Z') transb = (char)(transb - 'A' + 'a');
  *sa = (transa != 'n') ? m * lda : k * lda;
  *sb = (transb != 'n') ? ldb * k : n * ldb;
  *sc = ldc * n;
}

/* High-precision truth for error bounds: C_exact = alpha*sum + beta*C in long
 * double from double-converted inputs, and the magnitude sum
 * |alpha|*sum|a||b| + |beta||c| that scales the k-dependent bound.  Inputs are
 * given as doubles (exact images of the float / half data). */
int oracle_gemm_truth(int ta, int tb, int64_t m, int64_t n, int64_t k, double alpha,
                      const double* a, int64_t lda, const double* b, int64_t ldb, double beta,
                      const double* c, int64_t ldc, double* out, double* mag, int64_t ldo) {
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < n; ++j) {
    for (int64_t i = 0; i < m; ++i) {
      long double s = 0.0L, sa = 0.0L;
      for (int64_t l = 0; l < k; ++l) {
        const long double av = ta ? a[l + i * lda] : a[i + l * lda];
        const long double bv = tb ? b[j + l * ldb] : b[l + j * ldb];
        s += av * bv;
        sa += fabsl(av * bv);
      }
      long double r = (long double)alpha * s;
      long double ra = fabsl((long double)alpha) * sa;
      if (beta != 0.0) {
        r += (long double)beta * c[i + j * ldc];
        ra += fabsl((long double)beta * c[i + j * ldc]);
      }
      out[i + j * ldo] = (double)r;
      mag[i + j * ldo] = (double)ra;
    }
  }
  return ORC_OK;
}

/* The reference's element comparison (common/include/common/float_comparison.hpp:163-188):
 * returns the index of the first mismatch or -1.  abs_margin / rel_margin are
 * the per-type margins of :100-158. */
int64_t oracle_compare_f64(const double* vec, const double* ref, int64_t count, double abs_margin,
                           double rel_margin) {
  for (int64_t i = 0; i < count; ++i) {
    const double x = vec[i], y = ref[i];
    if (x == y) continue;
    if ((isnan(x) && isnan(y)) || (isinf(x) && isinf(y))) continue;
    const double d = fabs(x - y);
    if (x == 0.0 || y == 0.0 || d < abs_margin) {
      if (d < abs_margin) continue;
      return i;
    }
    if (d / (fabs(x) + fabs(y)) < rel_margin) continue;
    return i;
  }
  return -1;
}

/* ---------------------------------------------------------------- half -> half
 * The reference's arithmetic for the half -> half pair.  With Tin == Tout its mul_add is
 * sycl::mad(half, half, half) (src/operations/blas3/gemm_common.hpp:45-51): the accumulator
 * lives in binary16 and every k step rounds to binary16 -- unlike the half -> float pair,
 * which widens both operands first (:53-59).  alpha, beta and every epilogue operation are
 * binary16 too (src/interface/blas3/gemm.cpp.in:33-39).
 *
 * binary16 values are carried as bit patterns and the arithmetic runs in double: a product of
 * two binary16 values has 22 significant bits (exact in double), and the sum of that product
 * and a binary16 accumulator is rounded to 53 bits before it is rounded to 11 -- which cannot
 * change the result: the first rounding only happens when the two terms are more than 2^30
 * apart, and then the 53-bit value is never a binary16 tie that the exact sum is not.  So
 * fused(a, b, c) below is the correctly rounded binary16 fma.  sycl::mad leaves it to the
 * device whether the product is rounded on its own; `fused` == 0 restates that unfused variant. */
static double half_bits_to_double(uint16_t h) {
  const int sign = h >> 15, e = (h >> 10) & 31, man = h & 1023;
  double v;
  if (e == 31) v = man ? NAN : INFINITY;
  else if (e == 0) v = ldexp((double)man, -24);
  else v = ldexp((double)(man | 1024), e - 25);
  return sign ? -v : v;
}

/* round-to-nearest-even double -> binary16 (overflow to infinity, gradual underflow) */
static uint16_t double_to_half_bits(double d) {
  uint64_t u;
  memcpy(&u, &d, sizeof u);
  const uint16_t sign = (uint16_t)((u >> 48) & 0x8000u);
  const int e = (int)((u >> 52) & 0x7ff);
  uint64_t man = u & 0xfffffffffffffULL;
  if (e == 0x7ff) return (uint16_t)(sign | 0x7c00u | (man ? 0x200u : 0u));
  if (e == 0) return sign; /* zero or a double subnormal: far below half the smallest binary16 */
  const int he = e - 1023 + 15;
  if (he >= 31) return (uint16_t)(sign | 0x7c00u);
  man |= (uint64_t)1 << 52;
  int shift = 42; /* 53-bit significand -> 11 bits */
  if (he <= 0) shift += 1 - he;
  if (shift > 63) return sign;
  uint64_t q = man >> shift;
  const uint64_t rem = man & (((uint64_t)1 << shift) - 1), tie = (uint64_t)1 << (shift - 1);
  if (rem > tie || (rem == tie && (q & 1))) ++q;
  /* normal: q in [1024, 2048]; q - 1024 added to the exponent field carries by itself */
  const uint32_t out = he <= 0 ? (uint32_t)q : ((uint32_t)he << 10) + (uint32_t)(q - 1024);
  return (uint16_t)(sign | (out >= 0x7c00u ? 0x7c00u : out));
}

static double rh(double x) { return half_bits_to_double(double_to_half_bits(x)); }

void oracle_half_from_double(const double* in, uint16_t* out, int64_t count) {
  for (int64_t i = 0; i < count; ++i) out[i] = double_to_half_bits(in[i]);
}
void oracle_half_to_double(const uint16_t* in, double* out, int64_t count) {
  for (int64_t i = 0; i < count; ++i) out[i] = half_bits_to_double(in[i]);
}

/* One half -> half GEMM with every operation rounded to binary16 (bit patterns in and out).
 * form: ORC_FORM_REF   alpha*acc + beta*C, chain from zero (gemm_ref.hpp:234-251), or
 *       ORC_FORM_LOCAL alpha*acc with the chain started at (beta/alpha)*C, the tiled kernels'
 *                      form (gemm_local.hpp:182, :375-409, :557-581) -- the one the reference's
 *                      AMD and DEFAULT back ends run for half (backend/default.hpp:150-199). */
int oracle_gemm_half_chain(int form, int fused, int ta, int tb, int64_t m, int64_t n, int64_t k,
                           uint16_t alpha_h, const uint16_t* a, int64_t lda, const uint16_t* b,
                           int64_t ldb, uint16_t beta_h, uint16_t* c, int64_t ldc) {
  if (form != ORC_FORM_REF && form != ORC_FORM_LOCAL) return ORC_INVALID_VALUE;
  const double alpha = half_bits_to_double(alpha_h), beta = half_bits_to_double(beta_h);
  const int beta_zero = (beta == 0.0);
  const double ratio = rh(beta / alpha);
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < m; ++i) {
      uint16_t* cp = c + i + j * ldc;
      double acc = 0.0;
      if (form == ORC_FORM_LOCAL && !beta_zero) acc = rh(ratio * half_bits_to_double(*cp));
      for (int64_t l = 0; l < k; ++l) {
        const double av = half_bits_to_double(ta ? a[l + i * lda] : a[i + l * lda]);
        const double bv = half_bits_to_double(tb ? b[j + l * ldb] : b[l + j * ldb]);
        acc = fused ? rh(av * bv + acc) : rh(rh(av * bv) + acc);
      }
      double out = rh(alpha * acc);
      if (form == ORC_FORM_REF && !beta_zero) out = rh(out + rh(beta * half_bits_to_double(*cp)));
      *cp = double_to_half_bits(out);
    }
  return ORC_OK;
}
