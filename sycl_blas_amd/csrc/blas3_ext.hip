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
#include <type_traits>

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
// consecutive column-major 16x16 blocks.  A wavefront handles four blocks, 16 lanes each; a lane
// owns one ROW of its block in registers.  Column j of the inverse needs, in every row above (upper)
// or below (lower) the diagonal, the product of that row's finished part with the still original
// column j of the block -- whose element (k, j) sits in lane k's register j and reaches the other
// lanes of the group by a lane shuffle.  No LDS, no barrier (the reference's form, 30 barriers per
// block over a shared 16x16 array: src/operations/blas3/trsm.hpp:93-153).  The arithmetic is that
// form's, value for value: reciprocal of the diagonal, one fma chain per element over k in the same
// order with the same operands (zeros of the other triangle included), one final multiply by the
// negated reciprocal -- so the inverted blocks, and with them every _trsm result, keep their bits.
// Elements outside the stored triangle or outside the matrix count as zero and are never read.
template <typename T, bool UNIT, bool UPPER>
__global__ void __launch_bounds__(64)
    trsm_invert_blocks_kernel(const T* __restrict__ a, int64_t lda, int64_t n, int64_t nblocks, T* __restrict__ inv) {
  constexpr int B = 16;
  const int i = threadIdx.x & 15;  // the row this lane owns
  const int64_t blk = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 4);
  const bool live = blk < nblocks;
  const int64_t origin = blk * B;
  T row[B];  // row[c] = element (i, c)
#pragma unroll
  for (int c = 0; c < B; ++c) {
    bool in_range = false;
    if (live) in_range = UPPER ? (i <= c) && (origin + c < n) : (i >= c) && (origin + i < n);
    T v = T(0);
    if (in_range) v = (UNIT && i == c) ? T(1) : a[(origin + c) * lda + origin + i];
    row[c] = v;
  }
  if (!UNIT) {
#pragma unroll
    for (int c = 0; c < B; ++c)
      if (c == i) row[c] = T(1) / row[c];
  }
  // one column of the inverse per step; every lane runs the same shuffles, the rows the column
  // does not reach keep their value
  auto column = [&](auto j_tag) {
    constexpr int J = decltype(j_tag)::value;
    T sum = T(0);
    if (UPPER) {
#pragma unroll
      for (int k = 0; k < J; ++k) sum = fma_rn(row[k], __shfl(row[J], k, B), sum);
    } else {
#pragma unroll
      for (int k = J + 1; k < B; ++k) sum = fma_rn(row[k], __shfl(row[J], k, B), sum);
    }
    const T diag = __shfl(row[J], J, B);
    const T scaled = sum * (UNIT ? T(-1) : -diag);
    const bool mine = UPPER ? i < J : i > J;
    row[J] = mine ? scaled : row[J];
  };
  auto columns = [&](auto... js) { (column(js), ...); };
  using std::integral_constant;
  if (UPPER) {
    columns(integral_constant<int, 1>{}, integral_constant<int, 2>{}, integral_constant<int, 3>{}, integral_constant<int, 4>{},
            integral_constant<int, 5>{}, integral_constant<int, 6>{}, integral_constant<int, 7>{}, integral_constant<int, 8>{},
            integral_constant<int, 9>{}, integral_constant<int, 10>{}, integral_constant<int, 11>{},
            integral_constant<int, 12>{}, integral_constant<int, 13>{}, integral_constant<int, 14>{},
            integral_constant<int, 15>{});
  } else {
    columns(integral_constant<int, 14>{}, integral_constant<int, 13>{}, integral_constant<int, 12>{},
            integral_constant<int, 11>{}, integral_constant<int, 10>{}, integral_constant<int, 9>{}, integral_constant<int, 8>{},
            integral_constant<int, 7>{}, integral_constant<int, 6>{}, integral_constant<int, 5>{}, integral_constant<int, 4>{},
            integral_constant<int, 3>{}, integral_constant<int, 2>{}, integral_constant<int, 1>{}, integral_constant<int, 0>{});
  }
  if (live) {
#pragma unroll
    for (int c = 0; c < B; ++c) inv[blk * B * B + c * B + i] = row[c];
  }
}

// Solves a diagonal super-block of up to 128 rows (columns on the right side) of the
// triangular system in one launch: the 16-wide block sweep of the reference restricted
// to the super-block -- X_i = alpha_i * inv(A_ii) * B_i, then B_j <- -(A_ji * X_i) + beta_i * B_j
// for the blocks j still to come -- with every sum the k-ordered fma chain, and every
// product / sum rounding, of the GEMM calls it replaces (trsm_interface.hpp:227-380).
// A workgroup owns a slab of 32 columns (rows on the right side) of the panel, keeps it
// in LDS from the first block to the last and writes the solved slab into X.
//
// The panel is indexed (p, q): p runs along the solved dimension inside the super-block,
// q along the free one.  LEFT:  element (p, q) is B[(o + p) + q * ldb];  right: B[q + (o + p) * ldb].
// W(p, k), the coefficient of X[k] in row p, is A[(o + p) + (o + k) * lda], or with WT the
// transposed access A[(o + k) + (o + p) * lda]  (WT = trans on the left, !trans on the right);
// the inverted diagonal blocks are read the same way.
constexpr int TRSM_DIAG_NT = 512;

template <typename T, bool LEFT, bool WT, bool FORWARD>
__global__ void __launch_bounds__(TRSM_DIAG_NT)
    trsm_diag_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ inv, const T* __restrict__ B,
                     int64_t ldb, T* __restrict__ X, int64_t ldx, int64_t o, int s, int64_t nq, T alpha,
                     int first_blk) {
  constexpr int CW = 32, SM = 128, BS = 16, NT = TRSM_DIAG_NT;
  constexpr int RG = NT / CW;  // row groups: a thread owns column q of rows pr, pr + RG, ...
  using vec_t = typename Mfma<T>::vec_t;
  constexpr int VL = Mfma<T>::V;
  constexpr int WSTR = BS + VL;                  // keeps every row 16-byte aligned
  constexpr int NW = SM * BS / NT;               // coefficient slots one thread stages per block
  __shared__ T V[SM][CW + 1];
  __shared__ T Xi[BS][CW + 1];
  __shared__ T Wi[BS][BS + 1];
  __shared__ __attribute__((aligned(16))) T Ws[SM][WSTR];
  const int t = threadIdx.x;
  const int64_t q0 = int64_t(blockIdx.x) * CW;
  const int nblk = (s + BS - 1) / BS;

  // The coefficients of a block -- its inverted diagonal block and the rows of op(A) that
  // couple it to the blocks still to come -- are fetched into registers one block ahead
  // and dropped into LDS at the top of the block's step.
  T winv = T(0), wreg[NW];
  auto geometry = [&](int step, int& pb, int& cb, int& rem0, int& rem1) {
    const int b = FORWARD ? step : nblk - 1 - step;
    pb = b * BS;
    cb = s - pb < BS ? s - pb : BS;
    rem0 = FORWARD ? pb + BS : 0;  // rows still to come
    rem1 = FORWARD ? (pb + BS < s ? s : pb + BS) : pb;
  };
  auto fetch = [&](int step) {
    int pb, cb, rem0, rem1;
    geometry(step, pb, cb, rem0, rem1);
    if (t < BS * BS) winv = inv[(o / BS + pb / BS) * (BS * BS) + t];  // column-major block: [r + c * 16]
    // slot idx <-> (row p of the super-block, k): fixed, so no division by a run-time count;
    // rows outside [rem0, rem1) are simply skipped
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int idx = t + i * NT;
      const int k = WT ? idx % BS : idx / SM;
      const int p = WT ? idx / BS : idx % SM;
      const bool live = p >= rem0 && p < rem1 && k < cb;
      T v = T(0);
      if (live) v = WT ? A[(o + pb + k) + (o + p) * lda] : A[(o + p) + (o + pb + k) * lda];
      wreg[i] = v;
    }
  };
  auto deposit = [&](int step) {
    int pb, cb, rem0, rem1;
    geometry(step, pb, cb, rem0, rem1);
    const int r = t % BS, c = t / BS;
    if (t < BS * BS) {
      if (WT) Wi[c][r] = winv;
      else Wi[r][c] = winv;
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int idx = t + i * NT;
      const int k = WT ? idx % BS : idx / SM;
      const int p = WT ? idx / BS : idx % SM;
      if (p >= rem0 && p < rem1) Ws[p][k] = wreg[i];
    }
  };

  fetch(0);
#pragma unroll 4
  for (int idx = t; idx < SM * CW; idx += NT) {
    const int p = LEFT ? idx % SM : idx / CW;  // lanes along the contiguous index
    const int q = LEFT ? idx / SM : idx % CW;
    if (p < s) {
      T v = T(0);
      if (q0 + q < nq) v = LEFT ? B[(o + p) + (q0 + q) * ldb] : B[(q0 + q) + (o + p) * ldb];
      V[p][q] = v;
    }
  }
  const int q = t % CW, pr = t / CW;
  for (int step = 0; step < nblk; ++step) {
    int pb, cb, rem0, rem1;
    geometry(step, pb, cb, rem0, rem1);
    const bool is_first = (pb / BS) == first_blk;
    deposit(step);
    if (step + 1 < nblk) fetch(step + 1);
    __syncthreads();
    if (cb == BS) {  // one X element per thread; all operands fetched before the chain starts
      T wi[BS], vk[BS];
#pragma unroll
      for (int k = 0; k < BS; ++k) {
        wi[k] = Wi[pr][k];
        vk[k] = V[pb + k][q];
      }
      T acc = T(0);
#pragma unroll
      for (int k = 0; k < BS; ++k) acc = fma_rn(wi[k], vk[k], acc);
      Xi[pr][q] = is_first ? alpha * acc : acc;
    } else if (pr < cb) {
      T acc = T(0);
      for (int k = 0; k < cb; ++k) acc = fma_rn(Wi[pr][k], V[pb + k][q], acc);
      Xi[pr][q] = is_first ? alpha * acc : acc;
    }
    __syncthreads();
    T xk[BS];
#pragma unroll
    for (int k = 0; k < BS; ++k) xk[k] = k < cb ? Xi[k][q] : T(0);
    if (pr < cb) V[pb + pr][q] = Xi[pr][q];
    if (cb == BS) {
      // two rows per trip, so the LDS reads of one hide behind the chain of the other
      for (int p = rem0 + pr; p < rem1; p += 2 * RG) {
        const int p2 = p + RG < rem1 ? p + RG : p;
        T w[2][BS];
#pragma unroll
        for (int g = 0; g < BS / VL; ++g) {
          const vec_t w0 = *reinterpret_cast<const vec_t*>(&Ws[p][g * VL]);
          const vec_t w1 = *reinterpret_cast<const vec_t*>(&Ws[p2][g * VL]);
#pragma unroll
          for (int e = 0; e < VL; ++e) {
            w[0][g * VL + e] = w0[e];
            w[1][g * VL + e] = w1[e];
          }
        }
        T bc0 = V[p][q], bc1 = V[p2][q];
        T acc0 = T(0), acc1 = T(0);
#pragma unroll
        for (int k = 0; k < BS; ++k) {
          acc0 = fma_rn(w[0][k], xk[k], acc0);
          acc1 = fma_rn(w[1][k], xk[k], acc1);
        }
        if (is_first) {
          bc0 = alpha * bc0;
          bc1 = alpha * bc1;
        }
        const T neg0 = T(-1) * acc0, neg1 = T(-1) * acc1;
        V[p][q] = neg0 + bc0;
        if (p2 != p) V[p2][q] = neg1 + bc1;
      }
    } else {
      for (int p = rem0 + pr; p < rem1; p += RG) {
        T acc = T(0);
        for (int k = 0; k < cb; ++k) acc = fma_rn(Ws[p][k], Xi[k][q], acc);
        T bc = V[p][q];
        if (is_first) bc = alpha * bc;
        const T neg = T(-1) * acc;
        V[p][q] = neg + bc;
      }
    }
    __syncthreads();
  }
#pragma unroll 4
  for (int idx = t; idx < SM * CW; idx += NT) {
    const int p = LEFT ? idx % SM : idx / CW;
    const int qq = LEFT ? idx / SM : idx % CW;
    if (p < s && q0 + qq < nq) {
      if (LEFT) X[(o + p) + (q0 + qq) * ldx] = V[p][qq];
      else X[(q0 + qq) + (o + p) * ldx] = V[p][qq];
    }
  }
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

// ---- chunked-accumulation GEMM launch (see gemm_chunked_kernel) ----------------
constexpr int TRSM_BS = 16;  // the reference's diagonal block (trsm_interface.hpp:169)

template <typename T, int BM, int BN, int BK, int WN, int OCC, bool TA, bool TB, bool REVERSE>
hipError_t launch_chunked_one(GemmArgs<T> g, hipStream_t stream) {
  using K = GemmTile<T, BM, BN, BK, 2, WN, TA, TB>;
  auto kernel = gemm_chunked_kernel<T, BM, BN, BK, TRSM_BS, 2, WN, TA, TB, REVERSE, OCC>;
  static PerDeviceOnce lds_limit;
  hipError_t e = lds_limit.run([&] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               int(K::LDS_BYTES));
  });
  if (e != hipSuccess) return e;
  g.tiles_m = int32_t((g.m + BM - 1) / BM);
  g.tiles_n = int32_t((g.n + BN - 1) / BN);
  const int64_t grid = int64_t(g.tiles_m) * g.tiles_n;
  if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  // 16-byte loads along M / N may run up to the next multiple of a vector on an M/N-major
  // operand; the caller provides storage up to there.
  constexpr int64_t V = 16 / sizeof(T);
  const int64_t m_lim = TA ? g.m : (g.m + V - 1) / V * V;
  const int64_t n_lim = TB ? (g.n + V - 1) / V * V : g.n;
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(K::NT), K::LDS_BYTES, stream, g, m_lim, n_lim);
  return hipGetLastError();
}

template <typename T, int BM, int BN, int BK, int WN, int OCC>
hipError_t launch_chunked_tile(const GemmArgs<T>& g, bool ta, bool tb, bool reverse, hipStream_t stream) {
  // the solver multiplies op(A) * X or X * op(A): at most one operand is transposed
  if (ta && tb) return hipErrorInvalidValue;
  if (reverse) {
    if (ta) return launch_chunked_one<T, BM, BN, BK, WN, OCC, true, false, true>(g, stream);
    if (tb) return launch_chunked_one<T, BM, BN, BK, WN, OCC, false, true, true>(g, stream);
    return launch_chunked_one<T, BM, BN, BK, WN, OCC, false, false, true>(g, stream);
  }
  if (ta) return launch_chunked_one<T, BM, BN, BK, WN, OCC, true, false, false>(g, stream);
  if (tb) return launch_chunked_one<T, BM, BN, BK, WN, OCC, false, true, false>(g, stream);
  return launch_chunked_one<T, BM, BN, BK, WN, OCC, false, false, false>(g, stream);
}

// C <- the k / 16 successive updates  C <- -1 * op(A)[:, q] * op(B)[q, :] + beta_q * C  (beta_0 = beta,
// then 1), chunks ascending or, with `reverse`, descending.  The caller has checked the operands.
template <typename T>
hipError_t launch_chunked(Handle* h, bool ta, bool tb, bool reverse, int64_t m, int64_t n, int64_t k, T alpha,
                          const T* a, int64_t lda, const T* b, int64_t ldb, T beta, T* c, int64_t ldc) {
  if (!(alpha == T(-1))) return hipErrorInvalidValue;  // the kernel subtracts
  GemmArgs<T> g{};
  g.a = a;
  g.b = b;
  g.c = c;
  g.m = m;
  g.n = n;
  g.k = k;
  g.lda = lda;
  g.ldb = ldb;
  g.ldc = ldc;
  g.batch = 1;
  g.split_k = 1;
  g.k_chunk = k;
  g.vec_a = g.vec_b = 1;
  g.alpha = alpha;
  g.beta = beta;
  g.beta_zero = (beta == T(0)) ? 1 : 0;
  // The 128 x 128 tile needs about two workgroups per compute unit to pay off; below that
  // the 64 x 64 tile spreads the panel over the whole chip.  Its K step holds two chunks
  // when k allows (k is 16 times a power of two here), which halves the chain of dependent
  // global loads a workgroup walks through.
  // (fp32: four waves of 64 x 64; fp64: eight waves of 64 x 32, so that accumulators and running tile
  // leave room for two waves per SIMD.)
  const int64_t tiles128 = ((m + 127) / 128) * ((n + 127) / 128);
  if (tiles128 >= 2 * int64_t(h->compute_units))
    return launch_chunked_tile<T, 128, 128, 16, sizeof(T) == 4 ? 2 : 4, 2>(g, ta, tb, reverse, h->stream);
  if (k % 32 == 0) return launch_chunked_tile<T, 64, 64, 32, 2, 2>(g, ta, tb, reverse, h->stream);
  return launch_chunked_tile<T, 64, 64, 16, 2, 2>(g, ta, tb, reverse, h->stream);
}

int hip_fail(Handle* h, hipError_t e, const char* what) {
  h->last_error = std::string(what) + ": " + hipGetErrorString(e);
  return e == hipErrorOutOfMemory ? SBAMD_ALLOC_FAILED : SBAMD_HIP_ERROR;
}

template <typename T>
int symm(sbamd_handle_t hh, char side, char uplo, int64_t m, int64_t n, T alpha, const T* a, int64_t lda,
         const T* b, int64_t ldb, T beta, T* c, int64_t ldc) {
  if (!hh) return SBAMD_INVALID_HANDLE;
  Handle* h = &hh->impl;
  // scratch, the expansion kernel and the kernel attributes all belong to the handle's device
  DeviceGuard guard(h->device);
  if (!guard.ok) return SBAMD_HIP_ERROR;
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

// The block sweep of the triangular solve.  The reference walks the 16-wide diagonal
// blocks in order; at block i it forms X_i = inv(A_ii) * B_i and then subtracts
// A_(rest,i) * X_i from every block still to come, one GEMM call each
// (trsm_interface.hpp:227-251 left/forward, :267-292 left/backward, :312-337
// right/backward, :354-380 right/forward).  An element of a later block therefore sees
// the updates of the earlier blocks one after the other, in sweep order.
//
// Here the same per-element sequence is produced with far fewer passes over B: the block
// range is halved recursively, the first half is solved, and all of its updates to the
// second half go out as ONE chunked-accumulation launch (k = the size of the first half,
// folded into the running tile every 16), before the second half is solved.  Every
// element still receives the same products in the same order with the same roundings, so
// the result equals the block-by-block sweep bit for bit; the work on B drops from
// O(K/16) passes to O(log K).  The chunked kernel loads 16-byte vectors: X is laid out for
// it, and an A it could not read that way is copied once into aligned, padded scratch.
template <typename T>
struct TrsmSweep {
  sbamd_handle_t hh;
  Handle* h;
  bool left, tr, forward;
  int64_t M, N;
  T alpha;
  const T* A;
  int64_t lda;
  T* B;
  int64_t ldb;
  T* X;
  int64_t ldx;
  const T* invA;
  const T* Ap;  // A as the GEMM calls read it: A itself, or its 16-byte-aligned, padded copy
  int64_t ldap;

  static constexpr int64_t bs = TRSM_BS;
  static constexpr int64_t super = 128;  // rows one trsm_diag_kernel launch solves
  int64_t first = 0;   // the block whose calls carry alpha (-1: already spent)
  bool fused = false;  // diagonal super-blocks go to trsm_diag_kernel
  int64_t panel_launches = 0, diag_launches = 0;

  // All blocks of [lo, hi) (at most `super` rows) in one launch.
  int diag(int64_t lo, int64_t hi) {
    const int64_t nq = left ? N : M;
    const unsigned grid = unsigned((nq + 31) / 32);
    const int fb = (first >= lo && first < hi) ? int((first - lo) / bs) : -1;
    const bool wt = left ? tr : !tr;
    const int s = int(hi - lo);
#define SBAMD_DIAG(L, W, F)                                                                                       \
  hipLaunchKernelGGL((trsm_diag_kernel<T, L, W, F>), dim3(grid), dim3(TRSM_DIAG_NT), 0, h->stream, A, lda, invA, B, ldb, X, \
                     ldx, lo, s, nq, alpha, fb)
    if (left) {
      if (wt) { if (forward) SBAMD_DIAG(true, true, true); else SBAMD_DIAG(true, true, false); }
      else    { if (forward) SBAMD_DIAG(true, false, true); else SBAMD_DIAG(true, false, false); }
    } else {
      if (wt) { if (forward) SBAMD_DIAG(false, true, true); else SBAMD_DIAG(false, true, false); }
      else    { if (forward) SBAMD_DIAG(false, false, true); else SBAMD_DIAG(false, false, false); }
    }
#undef SBAMD_DIAG
    ++diag_launches;
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SBAMD_OK : hip_fail(h, e, "trsm diagonal launch");
  }

  int64_t extent() const { return left ? M : N; }

  // X_i = alpha_i * inv(A_ii) * B_i   (or B_i * inv(A_ii) on the right)
  int leaf(int64_t i, int64_t cb) {
    const T al = (i == first) ? alpha : T(1);
    const char tc = tr ? 't' : 'n';
    if (left) return Api<T>::gemm(hh, tc, 'n', cb, N, cb, al, invA + i * bs, bs, B + i, ldb, T(0), X + i, ldx);
    return Api<T>::gemm(hh, 'n', tc, M, cb, cb, al, B + i * ldb, ldb, invA + i * bs, bs, T(0), X + i * ldx, ldx);
  }

  // Blocks [d0, d1) of B  -=  (the A block between them and the solved blocks [s0, s1)) * X[s0, s1)
  int update(int64_t s0, int64_t s1, int64_t d0, int64_t d1) {
    if (d1 <= d0) return SBAMD_OK;
    const bool has_first = first >= s0 && first < s1;
    const T beta = has_first ? alpha : T(1);
    const char tc = tr ? 't' : 'n';
    const int64_t k = s1 - s0;
    const T* ablk = left ? Ap + (!tr ? d0 + s0 * ldap : s0 + d0 * ldap) : Ap + (!tr ? s0 + d0 * ldap : d0 + s0 * ldap);
    if (k <= bs) {
      if (left) return Api<T>::gemm(hh, tc, 'n', d1 - d0, N, k, T(-1), ablk, ldap, X + s0, ldx, beta, B + d0, ldb);
      return Api<T>::gemm(hh, 'n', tc, M, d1 - d0, k, T(-1), X + s0 * ldx, ldx, ablk, ldap, beta, B + d0 * ldb, ldb);
    }
    ++panel_launches;
    hipError_t e = left ? launch_chunked<T>(h, tr, false, !forward, d1 - d0, N, k, T(-1), ablk, ldap, X + s0, ldx, beta,
                                            B + d0, ldb)
                        : launch_chunked<T>(h, false, tr, !forward, M, d1 - d0, k, T(-1), X + s0 * ldx, ldx, ablk,
                                            ldap, beta, B + d0 * ldb, ldb);
    return e == hipSuccess ? SBAMD_OK : hip_fail(h, e, "trsm update launch");
  }

  static int64_t split_blocks(int64_t nb) {  // largest power of two below nb
    int64_t p = 1;
    while (p * 2 < nb) p *= 2;
    return p;
  }

  // [lo, hi): lo is a multiple of 16; hi is one too or the end of the matrix.
  int solve(int64_t lo, int64_t hi) {
    const int64_t nb = (hi - lo + bs - 1) / bs;
    if (fused && hi - lo <= super) return diag(lo, hi);
    if (nb <= 1) return leaf(lo, hi - lo);
    int st;
    if (forward) {
      const int64_t mid = lo + split_blocks(nb) * bs;
      if ((st = solve(lo, mid)) != SBAMD_OK) return st;
      if ((st = update(lo, mid, mid, hi)) != SBAMD_OK) return st;
      return solve(mid, hi);
    }
    const int64_t mid = hi - split_blocks(nb) * bs;  // backward ranges hold whole blocks
    if ((st = solve(mid, hi)) != SBAMD_OK) return st;
    if ((st = update(mid, hi, lo, mid)) != SBAMD_OK) return st;
    return solve(lo, mid);
  }

  int run() {
    const int64_t K = extent();
    // alpha == 0 turns the first block's GEMM calls into the reference's scaling path
    // (gemm_interface.hpp:112-139): leave that to the GEMM entry point.
    fused = !(alpha == T(0));
    if (forward) {
      first = 0;
      return solve(0, K);
    }
    // Backward sweeps start at the ragged block, if there is one: it is handled on its own
    // (its update has k < 16), the blocks below it are whole.
    const int64_t special = (K % bs == 0) ? bs : (K % bs);
    first = K - special;
    if (special == bs) return solve(0, K);
    int st = leaf(first, special);
    if (st != SBAMD_OK || first == 0) return st;
    if ((st = update(first, K, 0, first)) != SBAMD_OK) return st;
    const int64_t top = first;
    first = -1;
    return solve(0, top);
  }
};

template <typename T>
int trsm(sbamd_handle_t hh, char side, char uplo, char trans, char diag, int64_t M, int64_t N, T alpha, const T* A,
         int64_t lda, T* B, int64_t ldb) {
  if (!hh) return SBAMD_INVALID_HANDLE;
  Handle* h = &hh->impl;
  DeviceGuard guard(h->device);
  if (!guard.ok) return SBAMD_HIP_ERROR;
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
  // X: the M x N window of B in a layout the vector loads can use (columns start 16-byte
  // aligned, rows exist up to a whole vector).  The reference copies the whole ldb-strided
  // buffer back and forth (trsm_interface.hpp:199-208, :384-389); the rows between M and ldb
  // make the round trip unchanged there and are simply left alone here.
  constexpr int64_t V = 16 / sizeof(T);
  const int64_t ldx = (M + V - 1) / V * V;
  const int64_t x_elems = ldx * N;
  // A copy of A when 16-byte loads cannot read it in place: unaligned, or read along its
  // columns (M/N-major) with an extent that ends inside a vector.
  const bool a_mn_major = left ? !tr : tr;
  const bool a_in_place = reinterpret_cast<uintptr_t>(A) % 16 == 0 && lda % V == 0 && !(a_mn_major && K % V != 0);
  const int64_t ldac = (K + V - 1) / V * V;
  const int64_t a_elems = a_in_place ? 0 : ldac * K;

  hipError_t e = ensure_aux(h, size_t(inv_elems + x_elems + a_elems) * sizeof(T));
  if (e != hipSuccess) return hip_fail(h, e, "trsm scratch");
  T* invA = static_cast<T*>(h->aux);
  T* X = invA + inv_elems;
  const T* Ap = A;
  int64_t ldap = lda;
  if (!a_in_place) {
    T* acopy = X + x_elems;
    e = hipMemcpy2DAsync(acopy, size_t(ldac) * sizeof(T), A, size_t(lda) * sizeof(T), size_t(K) * sizeof(T), size_t(K),
                         hipMemcpyDeviceToDevice, h->stream);
    if (e != hipSuccess) return hip_fail(h, e, "trsm copy of A");
    Ap = acopy;
    ldap = ldac;
  }
  e = hipMemsetAsync(invA, 0, size_t(inv_elems) * sizeof(T), h->stream);
  if (e != hipSuccess) return hip_fail(h, e, "trsm memset");
  const unsigned grid = unsigned((nblocks + 3) / 4);
  if (unit && upper) hipLaunchKernelGGL((trsm_invert_blocks_kernel<T, true, true>), dim3(grid), dim3(64), 0, h->stream, A, lda, K, nblocks, invA);
  if (!unit && upper) hipLaunchKernelGGL((trsm_invert_blocks_kernel<T, false, true>), dim3(grid), dim3(64), 0, h->stream, A, lda, K, nblocks, invA);
  if (unit && !upper) hipLaunchKernelGGL((trsm_invert_blocks_kernel<T, true, false>), dim3(grid), dim3(64), 0, h->stream, A, lda, K, nblocks, invA);
  if (!unit && !upper) hipLaunchKernelGGL((trsm_invert_blocks_kernel<T, false, false>), dim3(grid), dim3(64), 0, h->stream, A, lda, K, nblocks, invA);
  e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(h, e, "trsm inverter launch");
  // X starts as a copy of B
  e = hipMemcpy2DAsync(X, size_t(ldx) * sizeof(T), B, size_t(ldb) * sizeof(T), size_t(M) * sizeof(T), size_t(N),
                       hipMemcpyDeviceToDevice, h->stream);
  if (e != hipSuccess) return hip_fail(h, e, "trsm copy");
  const bool forward = left ? ((upper && tr) || (!upper && !tr)) : !((upper && tr) || (!upper && !tr));
  TrsmSweep<T> sw{hh, h, left, tr, forward, M, N, alpha, A, lda, B, ldb, X, ldx, invA, Ap, ldap};
  const int st = sw.run();
  h->last_trsm_diag = sw.diag_launches;
  h->last_trsm_panel = sw.panel_launches;
  if (st != SBAMD_OK) return st;
  // the result travels back into B (:384-389)
  e = hipMemcpy2DAsync(B, size_t(ldb) * sizeof(T), X, size_t(ldx) * sizeof(T), size_t(M) * sizeof(T), size_t(N),
                       hipMemcpyDeviceToDevice, h->stream);
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
