// 16-bit storage back end: half -> half, half -> float, bfloat16 -> float.
#include <cmath>

#include "gemm_dispatch.hpp"
#include "gemm_half.hpp"
#include "gemm_half_dma.hpp"

namespace sbamd {
namespace {

template <typename Tag, typename TOut, int BM, int BN, int BK, int WM, int WN, bool TA, bool TB>
hipError_t launch16_one(const GemmArgs16<TOut>& g, hipStream_t stream) {
  constexpr int NT = WM * WN * 64;
  constexpr size_t lds = 2 * size_t(TileLoader16<BM, BK, NT, TA>::ELEMS + TileLoader16<BN, BK, NT, !TB>::ELEMS) * 2;
  auto kernel = gemm_mfma16_kernel<Tag, TOut, BM, BN, BK, WM, WN, TA, TB>;
  static PerDeviceOnce lds_limit;
  hipError_t e = lds_limit.run([&] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               int(lds));
  });
  if (e != hipSuccess) return e;
  const int64_t grid = int64_t(g.tiles_m) * g.tiles_n * (g.split_k > 1 ? g.split_k : g.batch);
  if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(NT), lds, stream, g);
  return hipGetLastError();
}

template <typename Tag, typename TOut, int BM, int BN, int BK, int WM, int WN>
hipError_t launch16(GemmArgs16<TOut> g, bool ta, bool tb, hipStream_t stream) {
  g.tiles_m = int32_t(ceil_div(g.m, BM));
  g.tiles_n = int32_t(ceil_div(g.n, BN));
  if (ta) {
    return tb ? launch16_one<Tag, TOut, BM, BN, BK, WM, WN, true, true>(g, stream)
              : launch16_one<Tag, TOut, BM, BN, BK, WM, WN, true, false>(g, stream);
  }
  return tb ? launch16_one<Tag, TOut, BM, BN, BK, WM, WN, false, true>(g, stream)
            : launch16_one<Tag, TOut, BM, BN, BK, WM, WN, false, false>(g, stream);
}

template <typename Tag, typename TOut, int BM, int BN, int WM, int WN, bool TA, bool TB>
hipError_t launch16_dma_one(const GemmArgs16<TOut>& g, hipStream_t stream) {
  auto kernel = gemm_mfma16_ring_kernel<Tag, TOut, TA, TB, 0, BM, BN, WM, WN>;
  constexpr int lds = ring_lds_bytes(BM, BN);
  static_assert(lds <= 160 * 1024, "the ring must fit the LDS");
  static PerDeviceOnce lds_limit;
  hipError_t e = lds_limit.run([&] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  });
  if (e != hipSuccess) return e;
  const int64_t grid = int64_t(g.tiles_m) * g.tiles_n * (g.split_k > 1 ? g.split_k : g.batch);
  if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(WM * WN * 64), lds, stream, g);
  return hipGetLastError();
}

// The direct-to-LDS ring kernel: 256 x 256 tiles on 2 x 4 wavefronts or 128 x 128 on 2 x 2.
template <typename Tag, typename TOut, int BM, int BN, int WM, int WN>
hipError_t launch16_dma(GemmArgs16<TOut> g, bool ta, bool tb, hipStream_t stream) {
  g.tiles_m = int32_t(ceil_div(g.m, BM));
  g.tiles_n = int32_t(ceil_div(g.n, BN));
  if (ta) {
    return tb ? launch16_dma_one<Tag, TOut, BM, BN, WM, WN, true, true>(g, stream)
              : launch16_dma_one<Tag, TOut, BM, BN, WM, WN, true, false>(g, stream);
  }
  return tb ? launch16_dma_one<Tag, TOut, BM, BN, WM, WN, false, true>(g, stream)
            : launch16_dma_one<Tag, TOut, BM, BN, WM, WN, false, false>(g, stream);
}

// Which tile family for a 16-bit GEMM: estimated microseconds per family from the costs per unit of K that
// a compute unit showed on MI355X (bf16 NN, tools/sweep16.py), cheapest wins.
//   256 x 256 ring kernel, one workgroup per unit: 0.0254 us with every unit busy, 0.018 with few (the clock
//     gives back under load), so a round of r * cus tiles costs 0.018 + 0.009 r^2; a unit holds one such
//     workgroup, so nothing runs beside a tile's 256 KiB of C going out: 13.4 us per round on top for K <= 256
//     (8192 x 8192 x 256: 87 us, 54 of them the four epilogues), 9 us for longer K, where the units drift apart
//     and one's stores meet another's matrix work (8192 x 8192 x 512: 94 us, x 1024: 146, 8192 x 4096 x 1024: 75.5);
//   128 x 128 ring kernel, two workgroups per unit: 0.0086 us per unit of K and 0.7 us per tile a unit has to work
//     through (8192 x 8192 x 1024: 160 us, 8192^3: 1132); with at most one tile per unit 0.0049 + 0.0039 * (share of
//     the units that have one) -- 1024^3 10.0 us, 1536^3 16.2, 2048^3 23.1;
//   128 x 128 register-staged (operands the ring kernel cannot take): 0.0193 us for a resident pair, 0.0141
//     for a workgroup alone;
//   64 x 64: 0.0079 us alone, 0.0043 per workgroup from four per unit on;
//   the kernels with several workgroups per unit overlap one tile's stores with another's matrix work, but cannot
//     be faster than C goes out (5 TB/s + 8 us: 8192 x 8192 x 256 on 128 x 128 tiles 65-67 us, x 64: 46 us).
// Against the measured best of the three families over 24 shapes (768^3 ... 8192^3, 512 x 4096 x 4096, 8192 x 8192 x
// 64 ... 1024, 1024 x 16384 x 1024, 3000^3) this picks the best one everywhere (within 1 %).
inline int pick_family16(int64_t m, int64_t n, int64_t k, int64_t batch, int cus, bool ring_ok, double* t_best = nullptr) {
  auto tiles = [&](int64_t b) { return double(ceil_div(m, b) * ceil_div(n, b) * batch); };
  const double kk = double(k);
  const double q256 = tiles(256) / cus, q128 = std::ceil(tiles(128) / cus), q64 = std::ceil(tiles(64) / cus);
  const double whole = std::floor(q256), part = q256 - whole;
  const double c_out = 8.0 + double(m) * double(n) * double(batch) * 4.0 / 5.0e6;
  const double t_huge = 5.0 + kk * (whole * 0.0254 + (part > 0.0 ? 0.018 + 0.009 * part * part : 0.0)) +
                        (k <= 256 ? 13.4 : 9.0) * std::ceil(q256);
  const double t_ring128 = q128 <= 1.0 ? 5.0 + kk * (0.0049 + 0.0039 * tiles(128) / cus) : 5.0 + q128 * (kk * 0.0086 + 0.7);
  const double t_large = std::max(
      c_out, ring_ok ? t_ring128 : 5.0 + kk * (std::floor(q128 / 2.0) * 0.0193 + (std::fmod(q128, 2.0) != 0.0 ? 0.0141 : 0.0)));
  const double per64 = q64 <= 1.0 ? 0.0079 : q64 <= 2.0 ? 0.0105 : q64 <= 3.0 ? 0.0147 : 0.0043 * q64;
  const double t_small = std::max(c_out, 5.0 + kk * per64);
  if (t_best) *t_best = std::min(ring_ok ? t_huge : t_large, std::min(t_large, t_small));
  if (ring_ok && t_huge <= t_large && t_huge <= t_small) return SBAMD_KERNEL_HUGE;
  return t_large <= t_small ? SBAMD_KERNEL_LARGE : SBAMD_KERNEL_SMALL;
}

// How many K slabs on the 128 x 128 ring kernel (1 = no cut), from the same kind of estimate (tools/sweep16_split.py:
// within 10 % of the measured time over depths 8 .. 512 on thin shapes): two launches (6.5 us; 256 x 256 x 2048 in eight
// slabs: 9.8 us); a slab's K range at
// 0.0062 us per unit of K while every workgroup has a compute unit to itself, 0.0088 times the workgroups per unit
// beyond that; never faster than the operands stream (4.7 TB/s with one workgroup per unit, 3.5 with more, and
// no more than about 23 GB/s per workgroup);
// the reduction walks the slabs one after the other: 0.2 us per slab plus the slab's bytes written and read back.
// A cut has to beat the best single launch by 10 %.
inline int64_t pick_depth16(int64_t m, int64_t n, int64_t k, int cus, double t_single) {
  const double tiles = double(ceil_div(m, 128) * ceil_div(n, 128));
  const double bytes_ab = 2.0 * (double(m) * double(k) + double(k) * double(n));
  const double per_slab = 0.2 + double(m) * double(n) * 8.0 / 5e6;
  int64_t best = 1;
  double best_t = 0.9 * t_single;
  const int64_t d_max = std::min<int64_t>(512, k / 128);
  for (int64_t d = 2; d <= d_max; ++d) {
    const double wgs = tiles * double(d);
    const double q = std::ceil(wgs / cus);
    const double k_part = double(k) / double(d) * (q <= 1.0 ? 0.0062 : 0.0088 * q);
    const double stream = bytes_ab / std::min(q <= 1.0 ? 4.7e6 : 3.5e6, wgs * 23e3);
    const double t = 6.5 + std::max(k_part, stream) + double(d) * per_slab;
    if (t < best_t) best_t = t, best = d;
  }
  return best;
}

inline bool vec16_ok(const uint16_t* p, int64_t ld, int64_t stride, int64_t batch) {
  return (reinterpret_cast<uintptr_t>(p) % 16 == 0) && (ld % 8 == 0) && (batch <= 1 || stride % 8 == 0);
}

template <typename Tag, typename TOut, int IR, int IC>
hipError_t launch16_interleaved_tile(const GemmArgs16<TOut>& g, bool ta, bool tb, hipStream_t stream) {
  const int64_t total = ceil_div(g.m, IR) * ceil_div(g.n, IC) * g.batch;
  const int64_t grid = ceil_div(total, 256);
  if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
  if (ta) {
    if (tb) hipLaunchKernelGGL((gemm_interleaved16_kernel<Tag, TOut, true, true, IR, IC>), dim3((unsigned)grid), dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((gemm_interleaved16_kernel<Tag, TOut, true, false, IR, IC>), dim3((unsigned)grid), dim3(256), 0, stream, g);
  } else {
    if (tb) hipLaunchKernelGGL((gemm_interleaved16_kernel<Tag, TOut, false, true, IR, IC>), dim3((unsigned)grid), dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((gemm_interleaved16_kernel<Tag, TOut, false, false, IR, IC>), dim3((unsigned)grid), dim3(256), 0, stream, g);
  }
  return hipGetLastError();
}

// 4 x 4 or 8 x 8 block of C per lane, chosen as for the fp32 / fp64 kernel (interleaved_big_block).
template <typename Tag, typename TOut>
hipError_t launch16_interleaved(const GemmArgs16<TOut>& g, bool ta, bool tb, hipStream_t stream, int cus = 256) {
  if (interleaved_big_block(g.m, g.n, g.batch, cus)) return launch16_interleaved_tile<Tag, TOut, 8, 8>(g, ta, tb, stream);
  return launch16_interleaved_tile<Tag, TOut, 4, 4>(g, ta, tb, stream);
}

template <typename Tag, typename TOut, int IR, int IC>
hipError_t launch16_block(const GemmArgs16<TOut>& g, bool ta, bool tb, hipStream_t stream) {
  const int64_t total = ceil_div(g.m, IR) * ceil_div(g.n, IC) * g.batch;
  const int64_t grid = ceil_div(total, 256);
  if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
  if (ta) {
    if (tb) hipLaunchKernelGGL((gemm_block16_kernel<Tag, TOut, true, true, IR, IC>), dim3((unsigned)grid), dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((gemm_block16_kernel<Tag, TOut, true, false, IR, IC>), dim3((unsigned)grid), dim3(256), 0, stream, g);
  } else {
    if (tb) hipLaunchKernelGGL((gemm_block16_kernel<Tag, TOut, false, true, IR, IC>), dim3((unsigned)grid), dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((gemm_block16_kernel<Tag, TOut, false, false, IR, IC>), dim3((unsigned)grid), dim3(256), 0, stream, g);
  }
  return hipGetLastError();
}

template <typename Tag, typename TOut>
hipError_t run16(Handle* h, const Problem<uint16_t, TOut>& p, float alpha, float beta);

// Interleaved batches (first branch of the reference's dispatch, amd_gpu.hpp:58-74; for half inputs
// default.hpp:150-166).  Small items run with the batch index on the lanes; items with enough
// arithmetic per repacked byte are turned into one column-major matrix per item in the handle's
// second scratch area, run on the strided matrix-instruction kernels and written back.
template <typename Tag, typename TOut>
hipError_t run16_interleaved(Handle* h, const Problem<uint16_t, TOut>& p, const GemmArgs16<TOut>& g, float alpha,
                             float beta) {
  const double flops = 2.0 * double(p.m) * double(p.n) * double(p.k);
  const double moved = 2.0 * (2.0 * (double(p.m) * double(p.k) + double(p.k) * double(p.n)) +
                              sizeof(TOut) * (p.beta_zero ? 1.0 : 2.0) * double(p.m) * double(p.n));
  const bool repack = p.m >= 32 && p.n >= 32 && flops >= 10.0 * moved;
  if (!repack) {
    h->last_kernel = SBAMD_KERNEL_INTERLEAVED;
    return launch16_interleaved<Tag, TOut>(g, p.ta, p.tb, h->stream, h->compute_units > 0 ? h->compute_units : 256);
  }
  const int64_t a_rows = p.ta ? p.k : p.m, a_cols = p.ta ? p.m : p.k;
  const int64_t b_rows = p.tb ? p.n : p.k, b_cols = p.tb ? p.k : p.n;
  // packed leading dimensions padded to whole 16-byte vectors, so the strided kernels take their fast path
  const int64_t lda = ceil_div(a_rows, 8) * 8, ldb = ceil_div(b_rows, 8) * 8, ldc = ceil_div(p.m, 8) * 8;
  const int64_t na = lda * a_cols, nb = ldb * b_cols, nc = ldc * p.n;
  const size_t bytes_ab = size_t(na + nb) * size_t(p.batch) * 2;
  const size_t off_c = (bytes_ab + 255) & ~size_t(255);
  hipError_t e = ensure_aux(h, off_c + size_t(nc) * size_t(p.batch) * sizeof(TOut));
  if (e != hipSuccess) return e;
  uint16_t* pa = static_cast<uint16_t*>(h->aux);
  uint16_t* pb = pa + na * p.batch;
  TOut* pc = reinterpret_cast<TOut*>(static_cast<unsigned char*>(h->aux) + off_c);
  // the padding rows of the packed operands feed nothing (rows past m / k are never read as operands)
  auto to_packed = [&](auto* inter, auto* packed, int64_t rows, int64_t cols, int64_t ld_inter, int64_t ld_packed) {
    using E = std::remove_cv_t<std::remove_pointer_t<decltype(packed)>>;
    const int64_t grid = ceil_div(rows, 32) * ceil_div(p.batch, 32) * cols;
    if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((repack_interleaved_ld_kernel<E, true>), dim3((unsigned)grid), dim3(256), 0, h->stream,
                       const_cast<E*>(inter), packed, rows, cols, ld_inter, ld_packed, p.batch);
    return hipGetLastError();
  };
  if ((e = to_packed(p.a, pa, a_rows, a_cols, p.lda, lda)) != hipSuccess) return e;
  if ((e = to_packed(p.b, pb, b_rows, b_cols, p.ldb, ldb)) != hipSuccess) return e;
  if (!p.beta_zero && (e = to_packed(p.c, pc, p.m, p.n, p.ldc, ldc)) != hipSuccess) return e;
  Problem<uint16_t, TOut> q = p;
  q.interleaved = false;
  q.a = pa, q.lda = lda, q.stride_a = na;
  q.b = pb, q.ldb = ldb, q.stride_b = nb;
  q.c = pc, q.ldc = ldc, q.stride_c = nc;
  if ((e = run16<Tag, TOut>(h, q, alpha, beta)) != hipSuccess) return e;
  const int64_t grid = ceil_div(p.m, 32) * ceil_div(p.batch, 32) * p.n;
  if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL((repack_interleaved_ld_kernel<TOut, false>), dim3((unsigned)grid), dim3(256), 0, h->stream, p.c, pc,
                     p.m, p.n, p.ldc, ldc, p.batch);
  return hipGetLastError();
}

template <typename Tag, typename TOut>
hipError_t run16(Handle* h, const Problem<uint16_t, TOut>& p, float alpha, float beta) {
  GemmArgs16<TOut> g;
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
  g.tiles_m = g.tiles_n = 0;
  g.vec_a = vec16_ok(p.a, p.lda, p.stride_a, p.batch) ? 1 : 0;
  g.vec_b = vec16_ok(p.b, p.ldb, p.stride_b, p.batch) ? 1 : 0;
  g.alpha = alpha;
  g.beta = beta;
  g.beta_zero = p.beta_zero ? 1 : 0;
  g.split_k = 1;
  g.k_chunk = p.k;
  g.slice_stride = 0;
  if (p.interleaved) return run16_interleaved<Tag, TOut>(h, p, g, alpha, beta);

  const int cus = h->compute_units > 0 ? h->compute_units : 256;
  // Batches of very small items: lane-level kernel, one element of C per lane or -- from 256 elements per item on --
  // a 4 x 4 block (reported as the NAIVE family, like the fp32 / fp64 route for such batches).
  if (p.batch > 1 && p.m <= 16 && p.n <= 16 && h->kernel_override == SBAMD_KERNEL_AUTO) {
    h->last_kernel = SBAMD_KERNEL_NAIVE;
    return p.m * p.n >= 256 ? launch16_block<Tag, TOut, 4, 4>(g, p.ta, p.tb, h->stream)
                            : launch16_block<Tag, TOut, 1, 1>(g, p.ta, p.tb, h->stream);
  }
  // The direct-to-LDS 256x256 kernel needs at least one whole K step, 16-byte vectors and no
  // vector straddling the matrix edge.
  bool dma_ok = g.vec_a && g.vec_b && p.k >= RING_BK && (p.ta || p.m % 8 == 0) && (!p.tb || p.n % 8 == 0);
  // A big product whose operands that kernel cannot take as they lie (odd leading dimension or offset, M / N not a
  // multiple of 8 along a contiguous operand): copy the offending operand(s) into the handle's second scratch area
  // with columns padded to whole vectors, when the copies cost less than the register-staged kernels lose
  // (bf16 8188^3: 1800 -> 995 us).
  // (not for operands that already live in that scratch area: the repacked items of an interleaved batch)
  auto in_aux = [&](const void* q) {
    const unsigned char* lo = static_cast<const unsigned char*>(h->aux);
    return lo && static_cast<const unsigned char*>(q) >= lo && static_cast<const unsigned char*>(q) < lo + h->aux_bytes;
  };
  if (!dma_ok && p.batch == 1 && p.k >= RING_BK && h->kernel_override == SBAMD_KERNEL_AUTO && !in_aux(p.a) && !in_aux(p.b) &&
      !in_aux(p.c)) {
    const bool fix_a = !g.vec_a || (!p.ta && p.m % 8 != 0), fix_b = !g.vec_b || (p.tb && p.n % 8 != 0);
    const int64_t rows_a = p.ta ? p.k : p.m, cols_a = p.ta ? p.m : p.k, rows_b = p.tb ? p.n : p.k, cols_b = p.tb ? p.k : p.n;
    const int64_t lda2 = ceil_div(rows_a, 8) * 8, ldb2 = ceil_div(rows_b, 8) * 8;
    const double moved = (fix_a ? 4.0 * double(lda2) * double(cols_a) : 0.0) + (fix_b ? 4.0 * double(ldb2) * double(cols_b) : 0.0);
    const double t_copy = 3.0 * (int(fix_a) + int(fix_b)) + moved / 3.5e6;
    double t_ring = 0.0, t_staged = 0.0;
    pick_family16(p.m, p.n, p.k, p.batch, cus, true, &t_ring);
    pick_family16(p.m, p.n, p.k, p.batch, cus, false, &t_staged);
    // misaligned 16-byte loads cost the register-staged kernels about 1.7x (bf16 4092^3: 287 us against 163 at 4096^3)
    if (!g.vec_a || !g.vec_b) t_staged *= 1.7;
    const bool forced = (h->kernel_param & 1) != 0;  // SBAMD_KERNEL_AUTO with bit 0 of the parameter: always copy (tests)
    if (forced || t_ring + t_copy < 0.9 * t_staged) {
      const size_t bytes_a = fix_a ? size_t(lda2) * size_t(cols_a) * 2 : 0, bytes_b = fix_b ? size_t(ldb2) * size_t(cols_b) * 2 : 0;
      const size_t off_b = (bytes_a + 255) & ~size_t(255);
      hipError_t e = ensure_aux(h, off_b + bytes_b);
      if (e != hipSuccess) return e;
      auto copy = [&](const uint16_t* src, int64_t ld_src, uint16_t* dst, int64_t ld_dst, int64_t rows, int64_t cols) {
        const int64_t grid = ceil_div(ld_dst / 8 * cols, 256);
        if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
        hipLaunchKernelGGL(copy_pad16_kernel, dim3((unsigned)grid), dim3(256), 0, h->stream, src, ld_src, dst, ld_dst, rows, cols);
        return hipGetLastError();
      };
      if (fix_a) {
        uint16_t* dst = static_cast<uint16_t*>(h->aux);
        if ((e = copy(p.a, p.lda, dst, lda2, rows_a, cols_a)) != hipSuccess) return e;
        g.a = dst, g.lda = lda2, g.vec_a = 1;
        if (!p.ta) g.lim_a = lda2;
      }
      if (fix_b) {
        uint16_t* dst = reinterpret_cast<uint16_t*>(static_cast<unsigned char*>(h->aux) + off_b);
        if ((e = copy(p.b, p.ldb, dst, ldb2, rows_b, cols_b)) != hipSuccess) return e;
        g.b = dst, g.ldb = ldb2, g.vec_b = 1;
        if (p.tb) g.lim_b = ldb2;
      }
      dma_ok = true;
    }
  }
  int family = h->kernel_override;
  double t_single = 0.0;  // estimated microseconds of the best single launch (automatic choice only)
  if (family == SBAMD_KERNEL_HUGE && !dma_ok) family = SBAMD_KERNEL_AUTO;
  if (family != SBAMD_KERNEL_LARGE && family != SBAMD_KERNEL_SMALL && family != SBAMD_KERNEL_HUGE) {
    family = pick_family16(p.m, p.n, p.k, p.batch, cus, dma_ok, &t_single);
  }
  // Few tiles and a long K (thin transformer shapes: 48 x 4096 x 4096, 224 x 1024 x 32768; weight-gradient
  // shapes: 128 x 576 x 401408): cut K over extra workgroups, fp32 partial sums in the handle's cube, one
  // reduction that also applies beta*C and rounds to the output type.  The slabs run on the 128 x 128 ring
  // kernel where it can take the operands (128 x 576 x 401408: 318 -> 160 us, the operands then stream at
  // 3.6 TB/s), else on the register-staged 64 x 64 tile.  Pinned: the low 16 bits of the parameter give the
  // depth, bits 16-23 == 1 keep the register-staged tile.
  if (h->kernel_override == SBAMD_KERNEL_AUTO || h->kernel_override == SBAMD_KERNEL_SPLITK) {
    const bool pinned = h->kernel_override == SBAMD_KERNEL_SPLITK;
    const int64_t tiles64 = ceil_div(p.m, 64) * ceil_div(p.n, 64);
    const bool ring_slabs = dma_ok && !(pinned && ((h->kernel_param >> 16) & 0xff) == 1);
    int64_t depth = 1;
    if (pinned && (h->kernel_param & 0xffff) > 0) {
      depth = h->kernel_param & 0xffff;
    } else if (p.batch == 1 && ring_slabs) {
      depth = pick_depth16(p.m, p.n, p.k, cus, pinned ? 1e30 : t_single);
    } else if (p.batch == 1 && (pinned || (tiles64 * 2 <= cus && p.k >= 1024) || (tiles64 <= 2 * cus && p.k >= 2048))) {
      depth = std::min<int64_t>(ceil_div(4 * int64_t(cus), tiles64), p.k / 128);
    }
    if (p.batch == 1 && depth > 1) {
      const int64_t k_chunk = 32 * ceil_div(p.k, 32 * depth);
      depth = ceil_div(p.k, k_chunk);
      if (depth > 1) {
        const int64_t slice = p.m * p.n;
        hipError_t e = ensure_scratch(h, size_t(slice) * depth * sizeof(float));
        if (e != hipSuccess) return e;
        GemmArgs16<float> gp;
        gp.a = g.a, gp.b = g.b, gp.c = static_cast<float*>(h->scratch);
        gp.m = g.m, gp.n = g.n, gp.k = g.k;
        gp.lda = g.lda, gp.ldb = g.ldb, gp.ldc = p.m;
        gp.stride_a = gp.stride_b = gp.stride_c = 0;
        gp.batch = 1;
        gp.tiles_m = gp.tiles_n = 0;
        gp.vec_a = g.vec_a, gp.vec_b = g.vec_b;
        gp.lim_a = g.lim_a, gp.lim_b = g.lim_b;
        gp.alpha = alpha, gp.beta = 0.0f, gp.beta_zero = 1;
        gp.split_k = int32_t(depth), gp.k_chunk = k_chunk, gp.slice_stride = slice;
        e = ring_slabs ? launch16_dma<Tag, float, 128, 128, 2, 2>(gp, p.ta, p.tb, h->stream)
                       : launch16<Tag, float, 64, 64, 32, 2, 2>(gp, p.ta, p.tb, h->stream);
        if (e != hipSuccess) return e;
        const int64_t grid = ceil_div(slice, 256);
        hipLaunchKernelGGL((splitk_reduce16_kernel<TOut>), dim3((unsigned)grid), dim3(256), 0, h->stream,
                           static_cast<const float*>(h->scratch), slice, int(depth), p.c, p.m, p.n, p.ldc, beta,
                           g.beta_zero);
        h->last_kernel = SBAMD_KERNEL_SPLITK;
        h->last_depth = depth;
        h->last_k_chunk = k_chunk;
        return hipGetLastError();
      }
    }
    if (h->kernel_override == SBAMD_KERNEL_SPLITK) family = SBAMD_KERNEL_SMALL;
  }
  h->last_kernel = family;
  if (family == SBAMD_KERNEL_HUGE) {
    // Four wavefronts of 128 x 128 (one per SIMD, 256 accumulator registers each, a third fewer fragment bytes per
    // flop) measure 0.5-2.8 % ahead of eight of 128 x 64 at 4096^3 and 8192^3 in every layout (same box, interleaved
    // runs: NN 1362 against 1326 TFLOP/s, NT 1405 / 1371, TT 1360 / 1335, TN 1270 / 1264).  Their epilogue is twice as
    // long per wavefront, so eight wavefronts stay for short K (8192 x 8192 x 64: 46.9 against 54.2 us) and for tiles
    // at a ragged edge, which store element by element (3000^3: 68 against 83 us); pinned with bit 1 of the parameter
    // set they always run.  Same bits either way.
    const bool four = p.m % DMA_BM == 0 && p.n % DMA_BN == 0 && p.k >= 2048;
    if (!four || (h->kernel_override == SBAMD_KERNEL_HUGE && (h->kernel_param & 2)))
      return launch16_dma<Tag, TOut, DMA_BM, DMA_BN, 2, 4>(g, p.ta, p.tb, h->stream);
    return launch16_dma<Tag, TOut, DMA_BM, DMA_BN, 2, 2>(g, p.ta, p.tb, h->stream);
  }
  if (family == SBAMD_KERNEL_LARGE) {
    // pinned with bit 0 of the parameter set: keep the register-staged kernel where the ring kernel would run
    const bool keep_staged = h->kernel_override == SBAMD_KERNEL_LARGE && (h->kernel_param & 1);
    if (dma_ok && !keep_staged) return launch16_dma<Tag, TOut, 128, 128, 2, 2>(g, p.ta, p.tb, h->stream);
    return launch16<Tag, TOut, 128, 128, 64, 2, 2>(g, p.ta, p.tb, h->stream);
  }
  return launch16<Tag, TOut, 64, 64, 32, 2, 2>(g, p.ta, p.tb, h->stream);
}

}  // namespace

hipError_t run_gemm_h(Handle* h, const Problem<uint16_t, uint16_t>& p, float alpha, float beta) {
  return run16<HalfTag, uint16_t>(h, p, alpha, beta);
}
hipError_t run_gemm_hs(Handle* h, const Problem<uint16_t, float>& p, float alpha, float beta) {
  return run16<HalfTag, float>(h, p, alpha, beta);
}
hipError_t run_gemm_bs(Handle* h, const Problem<uint16_t, float>& p, float alpha, float beta) {
  return run16<Bf16Tag, float>(h, p, alpha, beta);
}

hipError_t run_scale_f16(Handle* h, uint16_t* c, int64_t rows, int64_t cols, int64_t ld, int64_t stride,
                         int64_t batch, float beta, bool assign_zero) {
  const int64_t total = rows * cols * batch;
  if (total <= 0) return hipSuccess;
  const int64_t grid = ceil_div(total, 256);
  if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(scale_matrix16_kernel, dim3((unsigned)grid), dim3(256), 0, h->stream, c, rows, cols, ld, stride,
                     batch, beta, assign_zero ? 1 : 0);
  return hipGetLastError();
}

}  // namespace sbamd
