// Callers of the GEMM path inside BLAS-3: _symm and _trsm.
//
// In the reference both are thin drivers over GEMM:
//   _symm  src/interface/symm_interface.hpp:34-73 -- one _gemm_backend call whose
//          tile loader mirrors the stored triangle (gemm_local.hpp:813-873);
//   _trsm  src/interface/trsm_interface.hpp:106-396 -- inverts the 16x16 diagonal
//          blocks of A (src/operations/blas3/trsm.hpp:64-154), then walks the block
//          rows/columns with two GEMM calls per block.
// Here _symm expands the stored triangle into a full K x K matrix in handle-owned
// scratch and calls the GEMM entry point; _trsm keeps the reference's algorithm
// call for call, with the block inverter as a small HIP kernel.
#include <algorithm>
#include <cctype>

#include "gemm_mfma.hpp"
#include "sbamd_internal.hpp"

namespace sbamd {
namespace {

// out (K x K, packed) <- the symmetric matrix whose `upper` or lower triangle is stored in a.
template <typename T>
__global__ void __launch_bounds__(256)
    symmetrize_kernel(const T* __restrict__ a, int64_t lda, int64_t k, int upper, T* __restrict__ out) {
  const int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= k * k) return;
  const int64_t i = idx % k, j = idx / k;
  const bool stored = upper ? (i <= j) : (i >= j);
  out[idx] = stored ? a[i + j * lda] : a[j + i * lda];
}

// Inverse of every 16x16 diagonal block of the triangular matrix A (n x n), written as
// consecutive column-major 16x16 blocks.  One 16-lane group per block, lane = row;
// the arithmetic (reciprocal of the diagonal, fma chains in k order, one final
// multiply) is that of src/operations/blas3/trsm.hpp:93-153 step for step.
// Elements outside the stored triangle or outside the matrix count as zero and are
// never read.
template <typename T, bool UNIT, bool UPPER>
__global__ void __launch_bounds__(64)
    trsm_invert_blocks_kernel(const T* __restrict__ a, int64_t lda, int64_t n, int64_t nblocks, T* __restrict__ inv) {
  constexpr int B = 16;
  __shared__ T lds[4][B * B];
  const int sub = threadIdx.x >> 4;
  const int i = threadIdx.x & 15;
  const int64_t blk = int64_t(blockIdx.x) * 4 + sub;
  const bool live = blk < nblocks;
  T* local = lds[sub];  // local[j + i*B] = element (row i, column j)
  const int64_t origin = blk * B;
  for (int j = 0; j < B; ++j) {
    bool in_range = false;
    if (live) in_range = UPPER ? (i <= j) && (origin + j < n) : (i >= j) && (origin + i < n);
    T v = T(0);
    if (in_range) v = (UNIT && i == j) ? T(1) : a[(origin + j) * lda + origin + i];
    local[j + i * B] = v;
  }
  __syncthreads();
  if (!UNIT) {
    local[i + i * B] = T(1) / local[i + i * B];
    __syncthreads();
  }
  if (UPPER) {
    for (int j = 1; j < B; ++j) {
      T sum = T(0);
      if (i < j)
        for (int k = 0; k < j; ++k) sum = fma_rn(local[k + i * B], local[j + k * B], sum);
      __syncthreads();
      if (i < j) local[j + i * B] = sum * (UNIT ? T(-1) : -local[j + j * B]);
      __syncthreads();
    }
  } else {
    for (int j = B - 2; j >= 0; --j) {
      T sum = T(0);
      if (i > j)
        for (int k = j + 1; k < B; ++k) sum = fma_rn(local[k + i * B], local[j + k * B], sum);
      __syncthreads();
      if (i > j) local[j + i * B] = sum * (UNIT ? T(-1) : -local[j + j * B]);
      __syncthreads();
    }
  }
  if (live)
    for (int j = 0; j < B; ++j) inv[blk * B * B + j * B + i] = local[j + i * B];
}

template <typename T>
struct Api;
template <>
struct Api<float> {
  static int gemm(sbamd_handle_t h, char ta, char tb, int64_t m, int64_t n, int64_t k, float al, const float* a,
                  int64_t lda, const float* b, int64_t ldb, float be, float* c, int64_t ldc) {
    return sbamd_sgemm(h, ta, tb, m, n, k, al, a, lda, b, ldb, be, c, ldc);
  }
};
template <>
struct Api<double> {
  static int gemm(sbamd_handle_t h, char ta, char tb, int64_t m, int64_t n, int64_t k, double al, const double* a,
                  int64_t lda, const double* b, int64_t ldb, double be, double* c, int64_t ldc) {
    return sbamd_dgemm(h, ta, tb, m, n, k, al, a, lda, b, ldb, be, c, ldc);
  }
};

int hip_fail(Handle* h, hipError_t e, const char* what) {
  h->last_error = std::string(what) + ": " + hipGetErrorString(e);
  return e == hipErrorOutOfMemory ? SBAMD_ALLOC_FAILED : SBAMD_HIP_ERROR;
}

template <typename T>
int symm(sbamd_handle_t hh, char side, char uplo, int64_t m, int64_t n, T alpha, const T* a, int64_t lda,
         const T* b, int64_t ldb, T beta, T* c, int64_t ldc) {
  if (!hh) return SBAMD_INVALID_HANDLE;
  Handle* h = &hh->impl;
  if (m < 0 || n < 0 || lda < 0 || ldb < 0 || ldc < 0) return SBAMD_INVALID_VALUE;
  side = char(std::tolower(static_cast<unsigned char>(side)));
  uplo = char(std::tolower(static_cast<unsigned char>(uplo)));
  if (uplo != 'u' && uplo != 'l') return SBAMD_INVALID_UPLO;  // symm_interface.hpp:51-53
  if (side != 'l' && side != 'r') return SBAMD_INVALID_SIDE;  // :70-72
  const int64_t k = side == 'l' ? m : n;
  // alpha == 0 (and empty problems) never look at A: the GEMM front end scales C.
  if (alpha == T(0) || m == 0 || n == 0) {
    return side == 'l' ? Api<T>::gemm(hh, 'n', 'n', m, n, k, alpha, a, lda, b, ldb, beta, c, ldc)
                       : Api<T>::gemm(hh, 'n', 'n', m, n, k, alpha, b, ldb, a, lda, beta, c, ldc);
  }
  if (!a) return SBAMD_INVALID_VALUE;
  hipError_t e = ensure_aux(h, size_t(k) * size_t(k) * sizeof(T));
  if (e != hipSuccess) return hip_fail(h, e, "symm scratch");
  T* full = static_cast<T*>(h->aux);
  const int64_t grid = (k * k + 255) / 256;
  hipLaunchKernelGGL((symmetrize_kernel<T>), dim3((unsigned)grid), dim3(256), 0, h->stream, a, lda, k,
                     uplo == 'u' ? 1 : 0, full);
  e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(h, e, "symmetrize launch");
  // left: C <- alpha * A * B + beta * C;  right: C <- alpha * B * A + beta * C
  return side == 'l' ? Api<T>::gemm(hh, 'n', 'n', m, n, k, alpha, full, k, b, ldb, beta, c, ldc)
                     : Api<T>::gemm(hh, 'n', 'n', m, n, k, alpha, b, ldb, full, k, beta, c, ldc);
}

template <typename T>
int trsm(sbamd_handle_t hh, char side, char uplo, char trans, char diag, int64_t M, int64_t N, T alpha, const T* A,
         int64_t lda, T* B, int64_t ldb) {
  if (!hh) return SBAMD_INVALID_HANDLE;
  Handle* h = &hh->impl;
  if (M < 0 || N < 0 || lda < 0 || ldb < 0) return SBAMD_INVALID_VALUE;
  if (M == 0 || N == 0 || lda == 0 || ldb == 0) return SBAMD_TRSM_INVALID_SIZE;  // trsm_interface.hpp:114-116
  side = char(std::tolower(static_cast<unsigned char>(side)));
  uplo = char(std::tolower(static_cast<unsigned char>(uplo)));
  trans = char(std::tolower(static_cast<unsigned char>(trans)));
  diag = char(std::tolower(static_cast<unsigned char>(diag)));
  if (side != 'l' && side != 'r') return SBAMD_TRSM_INVALID_SIDE;    // :123-131
  if (uplo != 'u' && uplo != 'l') return SBAMD_TRSM_INVALID_UPLO;
  if (trans != 'n' && trans != 't') return SBAMD_TRSM_INVALID_TRANS;
  if (diag != 'u' && diag != 'n') return SBAMD_TRSM_INVALID_DIAG;
  if (!A || !B) return SBAMD_INVALID_VALUE;

  const int64_t K = side == 'l' ? M : N;
  const bool unit = diag == 'u', upper = uplo == 'u', left = side == 'l', tr = trans == 't';
  constexpr int64_t bs = 16;
  const int64_t nblocks = (K + bs - 1) / bs;
  const int64_t inv_elems = nblocks * bs * bs;
  const int64_t b_elems = ldb * (N - 1) + M;

  hipError_t e = ensure_aux(h, size_t(inv_elems + b_elems) * sizeof(T));
  if (e != hipSuccess) return hip_fail(h, e, "trsm scratch");
  T* invA = static_cast<T*>(h->aux);
  T* X = invA + inv_elems;
  e = hipMemsetAsync(invA, 0, size_t(inv_elems) * sizeof(T), h->stream);
  if (e != hipSuccess) return hip_fail(h, e, "trsm memset");
  const unsigned grid = unsigned((nblocks + 3) / 4);
  if (unit && upper) hipLaunchKernelGGL((trsm_invert_blocks_kernel<T, true, true>), dim3(grid), dim3(64), 0, h->stream, A, lda, K, nblocks, invA);
  if (!unit && upper) hipLaunchKernelGGL((trsm_invert_blocks_kernel<T, false, true>), dim3(grid), dim3(64), 0, h->stream, A, lda, K, nblocks, invA);
  if (unit && !upper) hipLaunchKernelGGL((trsm_invert_blocks_kernel<T, true, false>), dim3(grid), dim3(64), 0, h->stream, A, lda, K, nblocks, invA);
  if (!unit && !upper) hipLaunchKernelGGL((trsm_invert_blocks_kernel<T, false, false>), dim3(grid), dim3(64), 0, h->stream, A, lda, K, nblocks, invA);
  e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(h, e, "trsm inverter launch");
  // X starts as a copy of B (the whole ldb-strided range), trsm_interface.hpp:199-208
  e = hipMemcpyAsync(X, B, size_t(b_elems) * sizeof(T), hipMemcpyDeviceToDevice, h->stream);
  if (e != hipSuccess) return hip_fail(h, e, "trsm copy");
  const int64_t ldx = ldb;
  const char tc = tr ? 't' : 'n';
  int st = SBAMD_OK;
#define GEMM(...)                                     \
  do {                                                \
    st = Api<T>::gemm(hh, __VA_ARGS__);               \
    if (st != SBAMD_OK) return st;                    \
  } while (0)

  if (left) {
    if ((upper && tr) || (!upper && !tr)) {  // forward over block rows, :227-251
      for (int64_t i = 0; i < M; i += bs) {
        const int64_t cb = std::min(M - i, bs);
        GEMM(tc, 'n', cb, N, cb, (i == 0) ? alpha : T(1), invA + i * bs, bs, B + i, ldb, T(0), X + i, ldx);
        if (i + bs >= M) break;
        const int64_t off = !tr ? ((i + bs) + i * lda) : (i + (bs + i) * lda);
        GEMM(tc, 'n', M - i - bs, N, bs, T(-1), A + off, lda, X + i, ldx, (i == 0) ? alpha : T(1), B + i + bs, ldb);
      }
    } else {  // backward, :267-292
      const int64_t special = (M % bs == 0) ? bs : (M % bs);
      const int64_t start = M - special;
      for (int64_t i = start; i >= 0; i -= bs) {
        const int64_t cb = (i == start) ? special : bs;
        GEMM(tc, 'n', cb, N, cb, (i == start) ? alpha : T(1), invA + i * bs, bs, B + i, ldb, T(0), X + i, ldx);
        if (i - bs < 0) break;
        GEMM(tc, 'n', i, N, cb, T(-1), A + (!tr ? (i * lda) : i), lda, X + i, ldx, (i == start) ? alpha : T(1), B, ldb);
      }
    }
  } else {
    if ((upper && tr) || (!upper && !tr)) {  // backward over block columns, :312-337
      const int64_t special = (N % bs == 0) ? bs : (N % bs);
      const int64_t start = N - special;
      for (int64_t i = start; i >= 0; i -= bs) {
        const int64_t cb = (i == start) ? special : bs;
        GEMM('n', tc, M, cb, cb, (i == start) ? alpha : T(1), B + i * ldb, ldb, invA + i * bs, bs, T(0), X + i * ldx, ldx);
        if (i - bs < 0) break;
        GEMM('n', tc, M, i, cb, T(-1), X + i * ldx, ldx, A + (!tr ? i : (i * lda)), lda, (i == start) ? alpha : T(1), B, ldb);
      }
    } else {  // forward, :354-380
      for (int64_t i = 0; i < N; i += bs) {
        const int64_t cb = std::min(N - i, bs);
        GEMM('n', tc, M, cb, cb, (i == 0) ? alpha : T(1), B + i * ldb, ldb, invA + i * bs, bs, T(0), X + i * ldx, ldx);
        if (i + bs > N) break;  // when i + bs == N the call below has no columns and returns at once
        const int64_t off = !tr ? (i + (bs + i) * lda) : ((i + bs) + i * lda);
        GEMM('n', tc, M, N - i - bs, bs, T(-1), X + i * ldx, ldx, A + off, lda, (i == 0) ? alpha : T(1), B + (i + bs) * ldb, ldb);
      }
    }
  }
#undef GEMM
  // the result travels back into B (:384-389)
  e = hipMemcpyAsync(B, X, size_t(b_elems) * sizeof(T), hipMemcpyDeviceToDevice, h->stream);
  if (e != hipSuccess) return hip_fail(h, e, "trsm copy back");
  return SBAMD_OK;
}

}  // namespace
}  // namespace sbamd

extern "C" {

int sbamd_ssymm(sbamd_handle_t h, char side, char uplo, int64_t m, int64_t n, float alpha, const float* a, int64_t lda,
                const float* b, int64_t ldb, float beta, float* c, int64_t ldc) {
  return sbamd::symm<float>(h, side, uplo, m, n, alpha, a, lda, b, ldb, beta, c, ldc);
}
int sbamd_dsymm(sbamd_handle_t h, char side, char uplo, int64_t m, int64_t n, double alpha, const double* a,
                int64_t lda, const double* b, int64_t ldb, double beta, double* c, int64_t ldc) {
  return sbamd::symm<double>(h, side, uplo, m, n, alpha, a, lda, b, ldb, beta, c, ldc);
}
int sbamd_strsm(sbamd_handle_t h, char side, char uplo, char trans, char diag, int64_t m, int64_t n, float alpha,
                const float* a, int64_t lda, float* b, int64_t ldb) {
  return sbamd::trsm<float>(h, side, uplo, trans, diag, m, n, alpha, a, lda, b, ldb);
}
int sbamd_dtrsm(sbamd_handle_t h, char side, char uplo, char trans, char diag, int64_t m, int64_t n, double alpha,
                const double* a, int64_t lda, double* b, int64_t ldb) {
  return sbamd::trsm<double>(h, side, uplo, trans, diag, m, n, alpha, a, lda, b, ldb);
}

}  // extern "C"
