// Host-side kernel selection and launch for the fp32 / fp64 MFMA kernels.
//
// Stands in for the reference's per-target dispatch header
// (src/interface/blas3/backend/amd_gpu.hpp:33-290), its launcher
// (src/interface/gemm_launcher.hpp:37-69) and the tall-skinny driver
// (src/sb_handle/portblas_handle.hpp:296-403).  The shape classes are the
// reference's (tall-skinny / small-or-batched / large); the tile shapes are
// chosen for CDNA4 MFMA instead of the MI210 auto-tuner table.
#pragma once

#include <algorithm>

#include "gemm_mfma.hpp"
#include "sbamd_internal.hpp"

namespace sbamd {

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

template <typename T>
inline bool vec_ok(const T* p, int64_t ld, int64_t stride, int64_t batch) {
  constexpr int V = 16 / sizeof(T);
  return (reinterpret_cast<uintptr_t>(p) % 16 == 0) && (ld % V == 0) &&
         (batch <= 1 || stride % V == 0);
}

template <typename T, int BM, int BN, int BK, int WM, int WN, int OCC, bool TA, bool TB>
hipError_t launch_one(const GemmArgs<T>& g, int64_t nz, hipStream_t stream) {
  using K = GemmTile<T, BM, BN, BK, WM, WN, TA, TB>;
  auto kernel = gemm_mfma_kernel<T, BM, BN, BK, WM, WN, TA, TB, OCC>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize,
                                       int(K::LDS_BYTES));
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int64_t grid = int64_t(g.tiles_m) * g.tiles_n * nz;
  if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  // A K range that fits one K step never touches the second LDS buffer: ask for
  // one stage only, which doubles the workgroups a compute unit can hold.
  const int64_t k_span = g.split_k > 1 ? g.k_chunk : g.k;
  const size_t lds = k_span <= BK ? K::LDS_BYTES / 2 : K::LDS_BYTES;
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(K::NT), lds, stream, g);
  return hipGetLastError();
}

template <typename T, int BM, int BN, int BK, int WM, int WN, int OCC = 1>
hipError_t launch_tile(GemmArgs<T> g, bool ta, bool tb, hipStream_t stream) {
  g.tiles_m = int32_t(ceil_div(g.m, BM));
  g.tiles_n = int32_t(ceil_div(g.n, BN));
  const int64_t nz = g.split_k > 1 ? g.split_k : g.batch;
  if (ta) {
    return tb ? launch_one<T, BM, BN, BK, WM, WN, OCC, true, true>(g, nz, stream)
              : launch_one<T, BM, BN, BK, WM, WN, OCC, true, false>(g, nz, stream);
  }
  return tb ? launch_one<T, BM, BN, BK, WM, WN, OCC, false, true>(g, nz, stream)
            : launch_one<T, BM, BN, BK, WM, WN, OCC, false, false>(g, nz, stream);
}

template <typename T>
hipError_t launch_interleaved(const GemmArgs<T>& g, bool ta, bool tb, hipStream_t stream) {
  constexpr int IR = 4, IC = 4;
  const int64_t total = ceil_div(g.m, IR) * ceil_div(g.n, IC) * g.batch;
  const int64_t grid = ceil_div(total, 256);
  if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
  if (ta) {
    if (tb) hipLaunchKernelGGL((gemm_interleaved_kernel<T, true, true, IR, IC>), dim3((unsigned)grid), dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((gemm_interleaved_kernel<T, true, false, IR, IC>), dim3((unsigned)grid), dim3(256), 0, stream, g);
  } else {
    if (tb) hipLaunchKernelGGL((gemm_interleaved_kernel<T, false, true, IR, IC>), dim3((unsigned)grid), dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((gemm_interleaved_kernel<T, false, false, IR, IC>), dim3((unsigned)grid), dim3(256), 0, stream, g);
  }
  return hipGetLastError();
}

template <typename T>
hipError_t launch_naive(const GemmArgs<T>& g, bool ta, bool tb, hipStream_t stream) {
  const int64_t total = g.m * g.n * g.batch;
  const int64_t grid = ceil_div(total, 256);
  if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
  if (ta) {
    if (tb) hipLaunchKernelGGL((gemm_naive_kernel<T, true, true>), dim3((unsigned)grid), dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((gemm_naive_kernel<T, true, false>), dim3((unsigned)grid), dim3(256), 0, stream, g);
  } else {
    if (tb) hipLaunchKernelGGL((gemm_naive_kernel<T, false, true>), dim3((unsigned)grid), dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((gemm_naive_kernel<T, false, false>), dim3((unsigned)grid), dim3(256), 0, stream, g);
  }
  return hipGetLastError();
}

// Tile families and their K steps, picked by measurement at 8192^3 on MI355X
// (TFLOP/s NN / NT / TN):
//   fp32 256x128x32: 136.6 / 134.3 / 137.1    128x128x16: 132.9 / 134.9 / 135.0
//   fp64 256x128x16:  73.4 /  73.4 /  73.6    128x128x8 :  71.6 /  71.9 /  71.9
// The small family stages a 64 x 64 x 64 batched item in four steps.
template <typename T>
struct Tiles;
template <>
struct Tiles<float> {
  static constexpr int LARGE_BK = 16, SMALL_BK = 16, HUGE_BK = 32, ONESHOT_BK = 64;
};
template <>
struct Tiles<double> {
  static constexpr int LARGE_BK = 8, SMALL_BK = 8, HUGE_BK = 16, ONESHOT_BK = 64;
};

// The reference's tall-skinny predicate (amd_gpu.hpp:77-81).
inline bool is_tall_skinny(int64_t m, int64_t n, int64_t k, int64_t batch) {
  return batch == 1 && ((k > 8192 && m <= 1024 && n <= 1024) || (k > 1024 && m <= 256 && n <= 256));
}

template <typename T>
hipError_t run_gemm_real(Handle* h, const Problem<T, T>& p) {
  GemmArgs<T> g;
  g.a = p.a;
  g.b = p.b;
  g.c = p.c;
  g.m = p.m;
  g.n = p.n;
  g.k = p.k;
  g.lda = p.lda;
  g.ldb = p.ldb;
  g.ldc = p.ldc;
  g.stride_a = p.stride_a;
  g.stride_b = p.stride_b;
  g.stride_c = p.stride_c;
  g.batch = int32_t(p.batch);
  g.split_k = 1;
  g.k_chunk = p.k;
  g.slice_stride = 0;
  g.tiles_m = g.tiles_n = 0;
  g.vec_a = vec_ok(p.a, p.lda, p.stride_a, p.batch) ? 1 : 0;
  g.vec_b = vec_ok(p.b, p.ldb, p.stride_b, p.batch) ? 1 : 0;
  g.alpha = p.alpha;
  g.beta = p.beta;
  g.beta_zero = p.beta_zero ? 1 : 0;

  if (p.interleaved) {
    // First branch of the reference's dispatch (amd_gpu.hpp:58-74).
    h->last_kernel = SBAMD_KERNEL_INTERLEAVED;
    return launch_interleaved<T>(g, p.ta, p.tb, h->stream);
  }

  const int cus = h->compute_units > 0 ? h->compute_units : 256;

  int family = h->kernel_override;
  if (family == SBAMD_KERNEL_AUTO) {
    if (is_tall_skinny(p.m, p.n, p.k, p.batch)) {
      family = SBAMD_KERNEL_SPLITK;
    } else {
      // The biggest tile that (a) is mostly filled by the matrix and (b) still
      // gives every compute unit a workgroup; otherwise the 64x64 tile, whose
      // many small workgroups keep the chip occupied on small and batched shapes
      // (the role of the reference's low-arithmetic-ratio configurations,
      // amd_gpu.hpp:242-271).
      auto fits = [&](int bm, int bn) {
        const int64_t tm = ceil_div(p.m, bm), tn = ceil_div(p.n, bn);
        const double fill = double(p.m) * double(p.n) / (double(tm * bm) * double(tn * bn));
        return fill >= 0.75 && tm * tn * p.batch >= cus;
      };
      // (fp32 with both operands transposed measured 112 TFLOP/s on the 256x128
      // tile against 136 on the 128x128 one at 8192^3, so it skips the big tile.)
      const bool skip_huge = sizeof(T) == 4 && p.ta && p.tb;
      if (!skip_huge && fits(256, 128)) family = SBAMD_KERNEL_HUGE;
      else if (fits(128, 128)) family = SBAMD_KERNEL_LARGE;
      else family = SBAMD_KERNEL_SMALL;
    }
  }

  if (family == SBAMD_KERNEL_SPLITK) {
    // Tile for the partial products: the large tile once M and N fill it.
    const bool big = p.m > 64 && p.n > 64;
    // 256x128 once M fills it: A is then streamed by half as many tile columns.
    const bool huge = p.m > 192 && p.n > 64 && !(sizeof(T) == 4 && p.ta && p.tb);
    const int bm = huge ? 256 : big ? 128 : 64;
    const int bn = big ? 128 : 64;
    const int bk = huge ? Tiles<T>::HUGE_BK : big ? Tiles<T>::LARGE_BK : Tiles<T>::SMALL_BK;
    const int64_t tiles_mn = ceil_div(p.m, bm) * ceil_div(p.n, bn);
    // Cube depth: four workgroups per compute unit as in the reference
    // (src/operations/blas3/gemm_partial_local.hpp:191-199); one per compute unit
    // for the eight-wave 256x128 workgroups, which measured best at 256x256x4M
    // (124 TFLOP/s at one per CU, 121 at four) -- unless pinned.
    const int64_t per_cu = huge ? 1 : 4;
    int64_t depth = h->kernel_param > 0 ? h->kernel_param : ceil_div(per_cu * int64_t(cus), tiles_mn);
    // The reference falls back to one final partial kernel for shallow K
    // (portblas_handle.hpp:323-331).
    if (h->kernel_param <= 0 && (depth == 1 || p.k <= 2048)) depth = 1;
    if (p.batch != 1) depth = 1;
    // K slab of one slice: whole K tiles (gemm_partial_local.hpp:174, :253).
    const int64_t k_chunk = std::max<int64_t>(bk, bk * ceil_div(p.k, int64_t(bk) * depth));
    depth = std::max<int64_t>(1, ceil_div(p.k, k_chunk));
    if (depth > 1) {
      const int64_t slice = p.m * p.n;
      hipError_t e = ensure_scratch(h, size_t(slice) * depth * sizeof(T));
      if (e != hipSuccess) return e;
      GemmArgs<T> gp = g;
      gp.c = static_cast<T*>(h->scratch);
      gp.ldc = p.m;
      gp.split_k = int32_t(depth);
      gp.k_chunk = k_chunk;
      gp.slice_stride = slice;
      gp.beta_zero = 1;
      e = huge  ? launch_tile<T, 256, 128, Tiles<T>::HUGE_BK, 4, 2>(gp, p.ta, p.tb, h->stream)
          : big ? launch_tile<T, 128, 128, Tiles<T>::LARGE_BK, 2, 2, 2>(gp, p.ta, p.tb, h->stream)
                : launch_tile<T, 64, 64, Tiles<T>::SMALL_BK, 2, 2>(gp, p.ta, p.tb, h->stream);
      if (e != hipSuccess) return e;
      const int64_t grid = ceil_div(slice, 256);
      hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)grid), dim3(256), 0, h->stream,
                         static_cast<const T*>(h->scratch), slice, int(depth), p.c, p.m, p.n,
                         p.ldc, p.beta, g.beta_zero);
      h->last_kernel = SBAMD_KERNEL_SPLITK;
      h->last_depth = depth;
      h->last_k_chunk = k_chunk;
      return hipGetLastError();
    }
    family = big ? SBAMD_KERNEL_LARGE : SBAMD_KERNEL_SMALL;
  }

  h->last_kernel = family;
  switch (family) {
    case SBAMD_KERNEL_LARGE:
      return launch_tile<T, 128, 128, Tiles<T>::LARGE_BK, 2, 2, 2>(g, p.ta, p.tb, h->stream);
    case SBAMD_KERNEL_HUGE:
      return launch_tile<T, 256, 128, Tiles<T>::HUGE_BK, 4, 2>(g, p.ta, p.tb, h->stream);
    case SBAMD_KERNEL_SMALL:
      // Short K (the batched 64x64x64 class): stage the whole K range at once --
      // one load burst, one barrier, no K loop -- so a CU overlaps the loads of
      // some items with the matrix work of others.
      if (p.k <= Tiles<T>::ONESHOT_BK && p.k > Tiles<T>::SMALL_BK)
        return launch_tile<T, 64, 64, Tiles<T>::ONESHOT_BK, 2, 2, sizeof(T) == 4 ? 4 : 2>(g, p.ta, p.tb, h->stream);
      return launch_tile<T, 64, 64, Tiles<T>::SMALL_BK, 2, 2>(g, p.ta, p.tb, h->stream);
    case SBAMD_KERNEL_NAIVE:
      return launch_naive<T>(g, p.ta, p.tb, h->stream);
    default:
      return hipErrorInvalidValue;
  }
}

template <typename T>
hipError_t run_scale_real(Handle* h, T* c, int64_t rows, int64_t cols, int64_t ld, int64_t stride,
                          int64_t batch, T beta, bool assign_zero) {
  const int64_t total = rows * cols * batch;
  if (total <= 0) return hipSuccess;
  const int64_t grid = ceil_div(total, 256);
  if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL((scale_matrix_kernel<T>), dim3((unsigned)grid), dim3(256), 0, h->stream, c,
                     rows, cols, ld, stride, batch, beta, assign_zero ? 1 : 0);
  return hipGetLastError();
}

}  // namespace sbamd
