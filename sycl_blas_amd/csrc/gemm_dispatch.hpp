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
#include <cstdlib>
#include <type_traits>

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

// SBAMD_GROUP_M=<n> overrides the band height of the workgroup -> tile map for every launch (a tuning
// switch for counter sweeps; the result does not depend on it).
inline int group_m_override() {
  static const int value = [] {
    const char* e = std::getenv("SBAMD_GROUP_M");
    const int v = e ? std::atoi(e) : 0;
    return v > 0 && v <= 64 ? v : 0;
  }();
  return value;
}

template <typename T, int BM, int BN, int BK, int WM, int WN, int OCC, bool TA, bool TB, typename MS = Mfma<T>>
hipError_t launch_one(const GemmArgs<T>& g, int64_t nz, hipStream_t stream) {
  using K = GemmTile<T, BM, BN, BK, WM, WN, TA, TB, MS>;
  auto kernel = gemm_mfma_kernel<T, BM, BN, BK, WM, WN, TA, TB, OCC, MS>;
  static PerDeviceOnce lds_limit;
  hipError_t e = lds_limit.run([&] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               int(K::LDS_BYTES));
  });
  if (e != hipSuccess) return e;
  const int64_t grid = int64_t(g.tiles_m) * g.tiles_n * nz;
  if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  // A K range that fits one K step never touches the second LDS buffer: ask for
  // one stage only, which doubles the workgroups a compute unit can hold.
  const int64_t k_span = g.split_k > 1 ? g.k_chunk : g.k;
  const size_t lds = k_span <= BK ? K::LDS_BYTES / 2 : K::LDS_BYTES;
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(K::NT), lds, stream, g);
  return hipGetLastError();
}

template <typename T, int BM, int BN, int BK, int WM, int WN, int OCC = 1, typename MS = Mfma<T>>
hipError_t launch_tile(GemmArgs<T> g, bool ta, bool tb, hipStream_t stream) {
  g.tiles_m = int32_t(ceil_div(g.m, BM));
  g.tiles_n = int32_t(ceil_div(g.n, BN));
  const int64_t nz = g.split_k > 1 ? g.split_k : g.batch;
  if (ta) {
    return tb ? launch_one<T, BM, BN, BK, WM, WN, OCC, true, true, MS>(g, nz, stream)
              : launch_one<T, BM, BN, BK, WM, WN, OCC, true, false, MS>(g, nz, stream);
  }
  return tb ? launch_one<T, BM, BN, BK, WM, WN, OCC, false, true, MS>(g, nz, stream)
            : launch_one<T, BM, BN, BK, WM, WN, OCC, false, false, MS>(g, nz, stream);
}

// fp32 tiles staged global -> LDS directly (gemm_mfma_dma_kernel): K in whole K steps, both operands
// 16-byte loadable, no vector straddling an edge.
template <int BM, int BN, int BK, int WM, int WN, int OCC, int STAGES, bool TA, bool TB>
hipError_t launch_dma_one(const GemmArgs<float>& g, int64_t nz, hipStream_t stream) {
  auto kernel = gemm_mfma_dma_kernel<BM, BN, BK, WM, WN, TA, TB, OCC, STAGES>;
  constexpr size_t lds = size_t(STAGES) * (BM + BN) * BK * sizeof(float);
  static_assert(lds <= 160 * 1024, "the stages must fit the LDS");
  static PerDeviceOnce lds_limit;
  hipError_t e = lds_limit.run([&] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               int(lds));
  });
  if (e != hipSuccess) return e;
  const int64_t grid = int64_t(g.tiles_m) * g.tiles_n * nz;
  if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(WM * WN * 64), lds, stream, g);
  return hipGetLastError();
}

template <int BM, int BN, int BK, int WM, int WN, int OCC, int STAGES>
hipError_t launch_dma_tile(GemmArgs<float> g, hipStream_t stream) {
  g.tiles_m = int32_t(ceil_div(g.m, BM));
  g.tiles_n = int32_t(ceil_div(g.n, BN));
  const int64_t nz = g.split_k > 1 ? g.split_k : g.batch;
  return launch_dma_one<BM, BN, BK, WM, WN, OCC, STAGES, false, true>(g, nz, stream);  // A 'n', B 't' (dma_ok)
}

// Should the direct-to-LDS kernel (gemm_mfma_dma_kernel) take this launch?  Not any more by itself: it was ahead of
// the compiled register-staged loop on the 64 x 64 tile with both operands M/N-contiguous (1024^3 NT 29.7 -> 27.1 us),
// but the register-staged tiles now run their K loop as a hand-scheduled assembly block (gemm_kloop_asm.hpp) and
// are ahead of it everywhere measured (NT 1024^3: 20.8 us against 20.6, 2048^3: 124.7 us against 141.3).  The kernel
// stays reachable for A/B timing: SBAMD_DMA_TILES=1 in the environment restores the old rule (same bits either way).
template <typename T>
inline bool dma_ok(const GemmArgs<T>& g, bool ta, bool tb, int64_t k_span, int bk) {
  if constexpr (sizeof(T) != 4) return false;
  static const bool enabled = [] {
    const char* e = std::getenv("SBAMD_DMA_TILES");
    return e && e[0] == '1';
  }();
  return enabled && !ta && tb && g.vec_a && g.vec_b && k_span >= 2 * bk && g.m % 4 == 0 && g.n % 4 == 0;
}

// Batches of one-tile, one-step items on a fixed set of workgroups (gemm_batch_stream_kernel).
template <typename T, int BK, int OCC, bool TA, bool TB, bool READ_C, bool NT_LOADS>
hipError_t launch_stream_nt(const GemmArgs<T>& g, int cus, hipStream_t stream) {
  using K = GemmTile<T, 64, 64, BK, 2, 2, TA, TB>;
  auto kernel = gemm_batch_stream_kernel<T, 64, BK, TA, TB, READ_C, NT_LOADS, OCC>;
  constexpr size_t lds = K::LDS_BYTES / 2;  // one stage
  static_assert(lds * OCC <= 160 * 1024, "the resident workgroups must fit the LDS");
  static PerDeviceOnce lds_limit;
  hipError_t e = lds_limit.run([&] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               int(lds));
  });
  if (e != hipSuccess) return e;
  const int64_t items = g.tiles_m > 0 ? int64_t(g.tiles_m) * int64_t(g.tiles_n) : int64_t(g.batch);
  const int64_t grid = std::min<int64_t>(items, int64_t(cus) * OCC);
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(256), lds, stream, g);
  return hipGetLastError();
}

template <typename T, int BK, int OCC, bool TA, bool TB, bool READ_C>
hipError_t launch_stream_one(const GemmArgs<T>& g, int cus, hipStream_t stream) {
  // the tile walk re-reads its operand panels out of the L2: never non-temporal
  if (g.tiles_m > 0) return launch_stream_nt<T, BK, OCC, TA, TB, READ_C, false>(g, cus, stream);
  // operands that will not stay in the Infinity Cache anyway are loaded non-temporally
  const double bytes = double(g.batch) * sizeof(T) *
                       (double(g.m) * double(g.k) + double(g.k) * double(g.n) + (READ_C ? 2.0 : 1.0) * double(g.m) * double(g.n));
  return bytes > 320e6 ? launch_stream_nt<T, BK, OCC, TA, TB, READ_C, true>(g, cus, stream)
                       : launch_stream_nt<T, BK, OCC, TA, TB, READ_C, false>(g, cus, stream);
}

template <typename T, int BK, int OCC, bool READ_C>
hipError_t launch_stream_rc(const GemmArgs<T>& g, bool ta, bool tb, int cus, hipStream_t stream) {
  // fp32 with both operands K-major needs a few registers more than four workgroups per unit leave
  constexpr int OCC_TN = (sizeof(T) == 4 && OCC > 3) ? 3 : OCC;
  if (ta) {
    return tb ? launch_stream_one<T, BK, OCC, true, true, READ_C>(g, cus, stream)
              : launch_stream_one<T, BK, OCC_TN, true, false, READ_C>(g, cus, stream);
  }
  return tb ? launch_stream_one<T, BK, OCC, false, true, READ_C>(g, cus, stream)
            : launch_stream_one<T, BK, OCC, false, false, READ_C>(g, cus, stream);
}

template <typename T, int BK, int OCC>
hipError_t launch_stream(const GemmArgs<T>& g, bool ta, bool tb, int cus, hipStream_t stream) {
  // the C operands of an item wait in registers during its matrix work: one workgroup fewer per unit
  return g.beta_zero ? launch_stream_rc<T, BK, OCC, false>(g, ta, tb, cus, stream)
                     : launch_stream_rc<T, BK, (OCC > 2 ? OCC - 1 : OCC), true>(g, ta, tb, cus, stream);
}

// Both halves of K in one workgroup (gemm_mfma_halves_kernel); g.k_chunk = k / 2.
template <typename T, int BM, int BN, int BK, int WM, int WN, int OCC, bool TA, bool TB, typename MS = Mfma<T>, int SETS = 2>
hipError_t launch_halves_one(const GemmArgs<T>& g, hipStream_t stream) {
  using K = GemmTile<T, BM, BN, BK, WM, WN, TA, TB, MS>;
  auto kernel = gemm_mfma_halves_kernel<T, BM, BN, BK, WM, WN, TA, TB, OCC, MS, SETS>;
  // the sets' staging buffers, reused for the hand-over of SETS - 1 sets of accumulators
  constexpr size_t handover = size_t(SETS - 1) * BM * BN * sizeof(T);
  constexpr size_t lds = SETS * K::LDS_BYTES > handover ? SETS * K::LDS_BYTES : handover;
  static_assert(lds <= 160 * 1024, "the sets' buffers must fit the LDS");
  static PerDeviceOnce lds_limit;
  hipError_t e = lds_limit.run([&] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               int(lds));
  });
  if (e != hipSuccess) return e;
  const int64_t grid = int64_t(g.tiles_m) * g.tiles_n * g.batch;
  if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(SETS * K::NT), lds, stream, g);
  return hipGetLastError();
}

template <typename T, int BM, int BN, int BK, int WM, int WN, int OCC, typename MS = Mfma<T>, int SETS = 2>
hipError_t launch_halves(GemmArgs<T> g, bool ta, bool tb, hipStream_t stream) {
  g.tiles_m = int32_t(ceil_div(g.m, BM));
  g.tiles_n = int32_t(ceil_div(g.n, BN));
  if (ta) {
    return tb ? launch_halves_one<T, BM, BN, BK, WM, WN, OCC, true, true, MS, SETS>(g, stream)
              : launch_halves_one<T, BM, BN, BK, WM, WN, OCC, true, false, MS, SETS>(g, stream);
  }
  return tb ? launch_halves_one<T, BM, BN, BK, WM, WN, OCC, false, true, MS, SETS>(g, stream)
            : launch_halves_one<T, BM, BN, BK, WM, WN, OCC, false, false, MS, SETS>(g, stream);
}

// The biggest tile: 256 x 256 (float) / 256 x 128 (double), eight waves as 4 (M) x 2 (N).
// With both operands transposed the fp32 kernel mirrors the wave grid to 2 x 4: A is then
// the K-major operand and B the N-major one, and the fragment traffic per wave is that of
// the untransposed case with the roles of M and N swapped.  NN uses 2 x 4 as well (measured).
template <typename T>
hipError_t launch_huge(GemmArgs<T> g, bool ta, bool tb, hipStream_t stream);

template <typename T, int IR, int IC>
hipError_t launch_interleaved_tile(const GemmArgs<T>& g, bool ta, bool tb, hipStream_t stream) {
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

// Register block of C per lane: 4 x 4 (eight loads feed sixteen multiply-adds per k) or 8 x 8 (sixteen loads feed
// sixty-four: the loop is bound by load issue, not by memory -- fp32 8^3 x 262144: 4.2 -> 5.7 TB/s, 16 x 16 x 64 x
// 32768: 2.9 -> 4.9, 64^3 x 4096: 130 -> 75 us).  The big block needs items it tiles without much padding and a
// batch that still fills the chip with a quarter of the lanes.
inline bool interleaved_big_block(int64_t m, int64_t n, int64_t batch, int cus) {
  const int64_t pad8 = ceil_div(m, 8) * 8 * (ceil_div(n, 8) * 8), pad4 = ceil_div(m, 4) * 4 * (ceil_div(n, 4) * 4);
  return 10 * pad8 <= 13 * pad4 && ceil_div(m, 8) * ceil_div(n, 8) * batch >= int64_t(cus) * 512;
}
template <typename T>
hipError_t launch_interleaved(const GemmArgs<T>& g, bool ta, bool tb, hipStream_t stream, int cus = 256) {
  if (interleaved_big_block(g.m, g.n, g.batch, cus)) return launch_interleaved_tile<T, 8, 8>(g, ta, tb, stream);
  return launch_interleaved_tile<T, 4, 4>(g, ta, tb, stream);
}

// Batches of very small items: a 4 x 4 block of C per lane (gemm_block_kernel).
template <typename T>
hipError_t launch_block(const GemmArgs<T>& g, bool ta, bool tb, hipStream_t stream) {
  constexpr int IR = 4, IC = 4;
  const int64_t total = ceil_div(g.m, IR) * ceil_div(g.n, IC) * g.batch;
  const int64_t grid = ceil_div(total, 256);
  if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
  if (ta) {
    if (tb) hipLaunchKernelGGL((gemm_block_kernel<T, true, true, IR, IC>), dim3((unsigned)grid), dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((gemm_block_kernel<T, true, false, IR, IC>), dim3((unsigned)grid), dim3(256), 0, stream, g);
  } else {
    if (tb) hipLaunchKernelGGL((gemm_block_kernel<T, false, true, IR, IC>), dim3((unsigned)grid), dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((gemm_block_kernel<T, false, false, IR, IC>), dim3((unsigned)grid), dim3(256), 0, stream, g);
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

// Tile families and their K steps, picked by measurement on MI355X.  The biggest tile
// at 8192^3: fp32 256x256 (each of the 8 waves owns 64x128; BK 16, the deepest that
// keeps the main loop free of spills): 143 TFLOP/s against 138 for 256x128x32 -- a
// third fewer global loads and LDS stores per flop; fp64 256x128 BK 16: 74 (256x256
// needs 256 accumulator registers per lane).  The 128x128 and 64x64 tiles were tuned
// over a sweep of mid-size shapes (doubling their K step from 16 / 8 gained 3 % and
// 10-15 % there: fewer barriers per matrix instruction for workgroups that run one
// or two to a compute unit).  K <= 64 on the 64x64 tile is staged in one shot.
template <typename T>
struct Tiles;
// SMALL_OCC: register bound of the 64 x 64 tile in waves per SIMD.  With a bound of one (512 registers) hipcc keeps
// the accumulators of the double kernel in accumulation registers and copies all 32 of them to vector registers and
// back around the matrix instructions of every K step (64 moves for 16 matrix instructions); bounded to two waves
// (256 registers, of which it needs 114) the moves are gone.  The float kernel had such moves only in its
// ragged-edge and clamped loops; it takes the same bound.
template <>
struct Tiles<float> {
  static constexpr int LARGE_BK = 32, SMALL_BK = 32, HUGE_BK = 16, ONESHOT_BK = 64, HUGE_BN = 256, HUGE_OCC = 2, TINY_BK = 16, STREAM_OCC = 4;
  static constexpr int LARGE_WM = 4;  // 128 x 128 as eight waves of 32 x 64: 4-6 % faster than four of 64 x 64 at 2048^3..4096^3
  static constexpr int SMALL_OCC = 2;
  static constexpr int MEDIUM_BK = 32;  // the 96 x 96 and 192 x 192 tiles, on the 16x16x4 instruction
  using MediumShape = MfmaF32Small;
};
template <>
struct Tiles<double> {
  static constexpr int LARGE_BK = 16, SMALL_BK = 16, HUGE_BK = 16, ONESHOT_BK = 64, HUGE_BN = 128, HUGE_OCC = 1, TINY_BK = 8, STREAM_OCC = 2;
  static constexpr int LARGE_WM = 2;  // four waves of 64 x 64 (eight waves gain nothing in fp64)
  static constexpr int SMALL_OCC = 2;
  static constexpr int MEDIUM_BK = 16;
  using MediumShape = Mfma<double>;
};

// What to launch: a tile shape and how many K slabs (1 = a single kernel).
enum { TILE_64 = 1, TILE_128 = 2, TILE_256 = 3, TILE_32 = 4, TILE_NAIVE = 5, TILE_96 = 6, TILE_192 = 7 };
// the tiles a K cut may run on (the cube route; two halves in one workgroup: 64, 128 and 96)
inline bool tile_cuts_k(int tile) { return (tile >= TILE_64 && tile <= TILE_256) || tile == TILE_96 || tile == TILE_192; }
struct Plan {
  int tile;
  int64_t depth;
  bool fused = false;  // depth == 2 with both halves of K inside one workgroup (no cube)
};

// Cost model of one launch, fitted to a sweep of every (tile, depth) pair over
// M, N in {64..2048}, K in {256..16384} on MI355X (benchmark/tune_gemm --csv).
// per_k[r-1] = microseconds per unit of K that a compute unit needs for r
// resident workgroups of the tile (latency-bound at r = 1 for the small tiles,
// matrix-pipe-bound when the unit is full).
template <typename T>
struct CostModel;
// The constants are a least-squares fit (tools/fit_cost_model.py) to ~1200
// measured candidates per type; against the best measured candidate of each
// shape the fitted model's choice is within 3 % at the 90th percentile.
struct CostOverheads {
  double t_single;  // one launch, us
  double t_split;   // partial-product launch + reduction launch, us
  double bw;        // cube write + read, bytes per us
  double c_depth;   // reduction: serial walk over the slabs, us per slab
};
template <>
struct CostModel<float> {
  // Fitted (tools/fit_cost_model.py) to a sweep of every candidate, each timed after 15 ms of its own launches,
  // taken with the tiles' K loops running as assembly blocks (profiles/r03/suites/tune_sweep_f32.csv; the compiled
  // loops' constants were small_k {0.0180, 0.0355, 0.0530, 0.0690}, large_k {0.0620, 0.1165}, huge_k 0.2300).  The
  // fit said small_k {0.0134, 0.0287, 0.0421, 0.0579}, large_k {0.0556, 0.1107}, huge_k 0.2272; entries it gets
  // mostly from short K slabs are set from long single launches instead: 64 x 64 alone on a unit 0.0145
  // (1024 x 1024 x 4096), four per unit 0.0595-0.0603 (4096^3, 8192^3); 128 x 128 alone 0.0562 (2048 x 2048 x 16384),
  // two per unit 0.1146 (4096^3) and 0.1183 (8192^3); 256 x 256 0.2197 (8192^3) and 0.2229 (4096^3) -- it has to
  // stay ahead of 128 x 128 on both, as measured.
  static constexpr double small_k[4] = {0.0140, 0.0290, 0.0435, 0.0597};
  static constexpr double large_k[2] = {0.0565, 0.1160};
  static constexpr double huge_k[1] = {0.2200};  // 256 x 256 tile
  // 96 x 96 tile on the 16x16x4 instruction, one / two workgroups per unit (assembly loop; 1536^3, 256 tiles:
  // 53.0 us -> 0.0314; 3072^3, 1024 tiles: 390 us -> 0.0627; 6144^3: 3054 us -> 0.0621)
  static constexpr double medium_k[2] = {0.0314, 0.0625};
  // 192 x 192 tile on the same instruction, eight wavefronts of 96 x 48, one workgroup per unit (3072^3, 256 tiles:
  // 388 us -> 0.1247; 6144^3, 1024 tiles: 3047 us -> 0.1240)
  static constexpr double wide_k[1] = {0.1245};
  static constexpr double medium_margin = 1.03;  // what the two tiles above must win by (short K with a partly filled last round reads up to 30 % over their model: 1024 x 6272 x 512)
  // two K halves in one workgroup, relative to two resident workgroups of the tile {64x64, 128x128}: 64 x 64 the
  // fit's median over the sweep; 128 x 128 from long launches (8192^3 0.960, 6144^3 0.966, 4096^3 0.976,
  // 2048 x 2048 x 16384 0.952)
  static constexpr double halves_factor[3] = {0.981, 0.965, 1.05};  // third: the 96 x 96 tile (1536^3: 55.3 us against 53.0 in one set)
  // four K quarters in one workgroup of four one-wave sets on the 32 x 32 tile: microseconds per unit of a
  // quarter's K and per round of workgroups over the units, and the launch's fixed part (512^3: 7.3 us,
  // 512 x 512 x 2048: 16.5; 640^3, two rounds: 11.2; 1024 x 1024 x 512, four rounds: 16.0)
  static constexpr double quarter_k = 0.0241, quarter_over = 4.2;
  // (the fit's 7.4 us for the two launches of a cut comes mostly from long slabs; cuts of launches under 15 us
  // measure 5.5-6.6 us, and the reference suites' small shapes choose better with 6.2)
  static constexpr CostOverheads over = {4.71, 6.2, 5.63e6, 0.0497};
};
template <>
struct CostModel<double> {
  // Same procedure (profiles/r03/suites/tune_sweep_f64.csv).  The fit: small_k {0.0335, 0.0693, 0.0815, 0.1149},
  // large_k {0.1128, 0.2205}, huge_k 0.2247, halves {0.789, 0.998}; from long single launches 256 x 128 runs
  // 0.2184 (8192^3) and 0.2225 (4096^3), 128 x 128 two per unit 0.2196 / 0.2235, alone 0.1113.
  static constexpr double small_k[4] = {0.0335, 0.0693, 0.0815, 0.1149};
  static constexpr double large_k[2] = {0.1120, 0.2215};
  static constexpr double huge_k[1] = {0.2200};
  // 96 x 96 (assembly loop; 1536^3: 100.1 us -> 0.0611 alone; 3072^3: 767 us -> 0.1238 for two per unit, 6144^3 0.128)
  // and 192 x 192 (3072^3: 777 us -> 0.2508; 6144^3: 6079 us -> 0.2473)
  static constexpr double medium_k[2] = {0.0611, 0.1245};
  static constexpr double wide_k[1] = {0.2490};
  static constexpr double medium_margin = 1.03;
  static constexpr double halves_factor[3] = {0.800, 0.998, 1.01};  // (96 x 96 at 1536^3: 102.6 us against 100.1 in one set)
  // (512^3: 9.8 us, 512 x 512 x 2048: 26.2; 640^3: 16.6; 768^3, three rounds: 25.5)
  static constexpr double quarter_k = 0.0428, quarter_over = 4.3;
  static constexpr CostOverheads over = {6.16, 6.81, 6.25e6, 0.0438};
};

template <typename T>
hipError_t launch_huge(GemmArgs<T> g, bool ta, bool tb, hipStream_t stream) {
  if constexpr (sizeof(T) == 4) {
    // K slabs streamed from HBM with A read along K: a K step of 32 fetches whole 128-byte lines of
    // every row (16 fetches half a line per step, and rows a large power of two apart crowd the
    // same memory channels).  256 x 256 x 4194304: TN 114.8 -> 133.5 TFLOP/s, TT 127.5 -> 131.5;
    // NN (B read along K) measured no gain on the mirrored 128 x 256 x 32 tile and keeps 256 x 256 x 16.
    if (g.split_k > 1 && ta) return launch_tile<T, 256, 128, 32, 4, 2, 1>(g, ta, tb, stream);
    // 2 x 4 waves (128 x 64 each): TT, where it mirrors NN's fragment traffic, and NN itself, where it
    // measured 0.7 % faster than 4 x 2 in alternating runs (8192^3: 143.6 against 142.6 TFLOP/s) and
    // needs no spills; NT measured 1.4 % slower that way and keeps 4 x 2, as does TN.
    if (ta == tb) {
      g.tiles_m = int32_t(ceil_div(g.m, 256));
      g.tiles_n = int32_t(ceil_div(g.n, 256));
      const int64_t nz = g.split_k > 1 ? g.split_k : g.batch;
      return ta ? launch_one<T, 256, 256, Tiles<T>::HUGE_BK, 2, 4, Tiles<T>::HUGE_OCC, true, true>(g, nz, stream)
                : launch_one<T, 256, 256, Tiles<T>::HUGE_BK, 2, 4, Tiles<T>::HUGE_OCC, false, false>(g, nz, stream);
    }
  }
  // 256 x 128 tiles (double): bands of 4 tile rows make the 32 tiles resident on an XCD a 4 x 8 patch, which
  // shares more operand rows in that XCD's L2 than 8 x 4 does (4*256 + 8*128 against 8*256 + 4*128 per K step)
  if (g.group_m == 0 && Tiles<T>::HUGE_BN == 128) g.group_m = 4;
  return launch_tile<T, 256, Tiles<T>::HUGE_BN, Tiles<T>::HUGE_BK, 4, 2, Tiles<T>::HUGE_OCC>(g, ta, tb, stream);
}

// Estimated microseconds for `tile` with the K range cut into `depth` slabs.
template <typename T>
inline double estimate_us(int tile, int64_t depth, int64_t m, int64_t n, int64_t k, int64_t batch, int cus,
                          bool fused = false) {
  using C = CostModel<T>;
  const int bm = tile == TILE_256 ? 256 : tile == TILE_128 ? 128 : tile == TILE_96 ? 96 : tile == TILE_192 ? 192 : 64;
  const int bn = tile == TILE_64 ? 64 : tile == TILE_128 ? 128 : tile == TILE_96 ? 96 : tile == TILE_192 ? 192 : Tiles<T>::HUGE_BN;
  const int bk = tile == TILE_256 ? Tiles<T>::HUGE_BK : tile == TILE_128 ? Tiles<T>::LARGE_BK : (tile == TILE_96 || tile == TILE_192) ? Tiles<T>::MEDIUM_BK : Tiles<T>::SMALL_BK;
  const int occ = (tile == TILE_256 || tile == TILE_192) ? 1 : (tile == TILE_128 || tile == TILE_96) ? 2 : 4;
  const double* per_k = tile == TILE_256 ? C::huge_k : tile == TILE_128 ? C::large_k : tile == TILE_96 ? C::medium_k : tile == TILE_192 ? C::wide_k : C::small_k;
  // slabs are whole K steps, so a cut into `depth` may come out as fewer slabs (K = 1024 into 12: eleven of 96)
  const int64_t k_slab = bk * ceil_div(k, int64_t(bk) * depth);
  if (!fused) depth = std::max<int64_t>(1, ceil_div(k, k_slab));
  const int64_t wgs = ceil_div(m, bm) * ceil_div(n, bn) * batch * depth;
  // workgroups the fullest compute unit runs; the sets of a fused workgroup always share a unit
  const int64_t busiest = fused ? ceil_div(wgs / depth, cus) * depth : ceil_div(wgs, cus);
  const int64_t full = busiest / occ, rem = busiest % occ;
  double us_per_k = double(full) * per_k[occ - 1] + (rem ? per_k[rem - 1] : 0.0);
  // A lone 64 x 64 workgroup on a long K slab of its own streams from HBM (no other workgroup reads the same
  // addresses at the same time, as the tiles of a single launch do): its K step cannot be shorter than the memory
  // latency its one step of lookahead has to cover (256 x 256 x 262144 in 16 slabs: 0.018 us per k, not 0.014;
  // the 128-wide tile with 64 slabs runs 16 % faster there).
  if (sizeof(T) == 4 && tile == TILE_64 && depth > 1 && !fused && busiest == 1 && k_slab >= 4096) us_per_k = std::max(us_per_k, 0.018);
  const int64_t k_chunk = k_slab;
  // a matrix narrower than one 16-byte vector along M or N (vector-shaped products): the tiles load element by
  // element, about three times the time per K step (1 x 4096 x 4096: the best cut is four times as deep),
  // and the rounds of resident workgroups wait for memory side by side: a round costs what one workgroup does
  if (std::min(m, n) < Mfma<T>::V) us_per_k = 3.0 * per_k[0] * double(full + (rem ? 1 : 0));
  if (fused) us_per_k *= C::halves_factor[tile == TILE_128 ? 1 : tile == TILE_96 ? 2 : 0];
  double us = us_per_k * double(k_chunk);
  if (depth > 1 && !fused) {
    us += C::over.t_split + 2.0 * double(m) * double(n) * double(depth) * sizeof(T) / C::over.bw +
          C::over.c_depth * double(depth);
  } else {
    us += C::over.t_single;
  }
  return us;
}

template <typename T>
inline Plan plan_gemm(int64_t m, int64_t n, int64_t k, int64_t batch, int cus, double cube_limit = double(1u << 30)) {
  // Batches of tiny matrices (the shapes the reference's interleaved layout exists
  // for): a 64x64 tile would be mostly padding.  Up to 16x16 one lane per C element
  // wins (4^3 x 65536: 5.6 us against 244; 16^3 x 8192: 14 against 36), up to 32x32 a
  // one-wave 32x32 tile.  All of them run the same k-ordered chain per element.
  if (batch > 1 && m <= 16 && n <= 16) return {TILE_NAIVE, 1};
  if (batch > 1 && m <= 32 && n <= 32) return {TILE_32, 1};
  // Short K (the batched 64x64x64 class) has a kernel of its own on the small tile.
  if (k <= Tiles<T>::ONESHOT_BK && m <= 64 && n <= 64) return {TILE_64, 1};
  // One short product with at most one 64 x 64 tile per unit: the one-wave 32 x 32 tile spreads it over four times
  // as many wavefront slots (512 x 512 x 16: 4.2 -> 3.1 us; 256 x 256 x 49: 5.5 -> 4.3; 2048 x 512 x 49: 5.9 -> 5.0;
  // fp64 100 x 700 x 30: 6.7 -> 4.6).
  if (batch == 1 && k <= 64 && ceil_div(m, 64) * ceil_div(n, 64) <= cus && ceil_div(m, 32) * ceil_div(n, 32) >= 4)
    return {TILE_32, 1};
  Plan best = {TILE_64, 1};
  double best_us = 1e300;
  for (int tile = TILE_64; tile <= TILE_256; ++tile) {
    const double us = estimate_us<T>(tile, 1, m, n, k, batch, cus);
    if (us < best_us) best_us = us, best = {tile, 1};
  }
  // The 96 x 96 and 192 x 192 tiles: where the other tile counts quantise badly over the units (fp32 1536^3: 576
  // tiles of 64 x 64 are three on the busiest unit, 256 tiles of 96 x 96 one each: 84 -> 65 us; 3072^3: 256 tiles
  // of 192 x 192, 489 -> 388 us).  They have to win by CostModel::medium_margin.
  for (int tile : {TILE_96, TILE_192}) {
    const double us = CostModel<T>::medium_margin * estimate_us<T>(tile, 1, m, n, k, batch, cus);
    if (us < best_us) best_us = us, best = {tile, 1};
  }
  if (batch != 1) return best;
  // Cutting K (the reference does it for its tall-skinny class only,
  // amd_gpu.hpp:77-81 / portblas_handle.hpp:296-403) pays whenever M x N alone
  // cannot fill the chip.  It changes the rounding (partial sums are added slab
  // by slab), so it has to beat the single launch by a clear margin.
  static const int depths[] = {2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256};
  Plan split = best;
  double split_us = 1e300;
  for (int tile : {TILE_64, TILE_128, TILE_256, TILE_96, TILE_192})
    for (int depth : depths) {
      if (k / depth < 64) break;                                                   // slabs of at least 64
      if (double(m) * double(n) * depth * sizeof(T) > cube_limit) break;           // cube of at most 1 GiB (or what the caller provides)
      const double us = (tile > TILE_256 ? CostModel<T>::medium_margin : 1.0) * estimate_us<T>(tile, depth, m, n, k, 1, cus);
      if (us < split_us) split_us = us, split = {tile, depth};
    }
  Plan chosen = split_us < 0.9 * best_us ? split : best;
  double chosen_us = split_us < 0.9 * best_us ? split_us : best_us;
  // Two slabs can also live in one workgroup of the 64x64 / 128x128 tiles: no cube, one
  // launch, twice the wavefronts per SIMD.  Only where the tiles alone leave at least
  // half of a compute unit's workgroup slots empty (1024^3 fp64 56.7 -> 41.2 us; fp32
  // 28.9 -> 26.6 and 2048^3 fp32 156 -> 138, which the 10 % rule for K cuts turns down);
  // with more tiles it loses to the plain launch.
  for (int tile : {TILE_64, TILE_128, TILE_96}) {
    const int edge = tile == TILE_128 ? 128 : tile == TILE_96 ? 96 : 64, occ = tile == TILE_64 ? 4 : 2;
    // equal, vector-aligned halves of at least eight K steps each (short K: the hand-over is not free)
    if (k % (2 * Mfma<T>::V) != 0 || k < 16 * Tiles<T>::SMALL_BK) continue;
    // 64 x 64: only where the tiles alone leave at least half of a unit's workgroup slots empty (the measured
    // factor is the gain of doubling the wavefronts of an under-filled unit; on a full unit there is none).
    // 128 x 128: any tile count -- a workgroup of two sets has a unit to itself, which also evens out tile counts
    // that leave the last round of two-per-unit workgroups half empty (fp32 6144^3: 2304 tiles are nine per unit,
    // 148 TFLOP/s this way against 132).
    if (tile != TILE_128 && ceil_div(m, edge) * ceil_div(n, edge) * 2 > int64_t(cus) * occ) continue;
    const double us = (tile == TILE_96 ? CostModel<T>::medium_margin : 1.0) * estimate_us<T>(tile, 2, m, n, k, 1, cus, true);
    // 4 % suffices here, 2 % for the 128 x 128 tile, whose model is built from long launches (a cut through the
    // cube has to win 10 %): the cut is always in two equal halves, whatever the device, and costs neither a
    // second launch nor a trip through memory
    if (us < (tile == TILE_128 ? 0.98 : 0.96) * best_us && us < chosen_us) chosen_us = us, chosen = {tile, 2, true};
  }
  // Four quarters of K inside one workgroup of four one-wave sets on the 32 x 32 tile: a product too small to fill
  // the chip with 64 x 64 tiles and too short for a cut through the cube to pay for its second launch, spread over
  // four times as many wavefronts in ONE launch (512^3: fp64 13.8 -> 9.8 us, fp32 11.1 -> 7.3).
  // (not for a sliver of C: its 32 x 32 tiles would be mostly padding)
  if (k % (4 * Mfma<T>::V) == 0 && k >= 128 && m >= 16 && n >= 16) {
    const int64_t rounds = ceil_div(ceil_div(m, 32) * ceil_div(n, 32), int64_t(cus));
    // (a further round of workgroups overlaps the one before it: it adds 0.85 of a round -- fp64 768^3, three
    // rounds: 25.5 us; 640^3, two: 16.6)
    const double us = CostModel<T>::quarter_over + (1.0 + 0.85 * double(rounds - 1)) * CostModel<T>::quarter_k * double(k / 4);
    if (us < 0.96 * chosen_us) chosen_us = us, chosen = {TILE_32, 4, true};
  }
  return chosen;
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
  g.group_m = group_m_override();  // 0 = each tile family's default (launch_huge sets double's)

  if (p.interleaved) {
    // First branch of the reference's dispatch (amd_gpu.hpp:58-74).  Small items run
    // with the batch index on the lanes; items big enough for the matrix pipes are
    // repacked and take the strided path (256^3 x 32: 81 -> 36 us).  Either way
    // every C element is the same k-ordered chain.
    // The repacking reads and writes every operand once: it pays from about 10 flops per
    // repacked byte (128^3: 1.8x faster, 256^3: 2.3-3.4x; 64 x 5000 x 64 at 8 would lose).
    const double flops = 2.0 * double(p.m) * double(p.n) * double(p.k);
    const double moved = 2.0 * sizeof(T) *
                         (double(p.m) * double(p.k) + double(p.k) * double(p.n) +
                          (p.beta_zero ? 1.0 : 2.0) * double(p.m) * double(p.n));
    const bool repack = p.m >= 32 && p.n >= 32 && flops >= 10.0 * moved * (4.0 / sizeof(T));
    if (!repack) {
      h->last_kernel = SBAMD_KERNEL_INTERLEAVED;
      return launch_interleaved<T>(g, p.ta, p.tb, h->stream, h->compute_units > 0 ? h->compute_units : 256);
    }
    const int64_t a_rows = p.ta ? p.k : p.m, a_cols = p.ta ? p.m : p.k;
    const int64_t b_rows = p.tb ? p.n : p.k, b_cols = p.tb ? p.k : p.n;
    const int64_t na = a_rows * a_cols, nb = b_rows * b_cols, nc = p.m * p.n;
    hipError_t e = ensure_aux(h, size_t(na + nb + nc) * size_t(p.batch) * sizeof(T));
    if (e != hipSuccess) return e;
    T* pa = static_cast<T*>(h->aux);
    T* pb = pa + na * p.batch;
    T* pc = pb + nb * p.batch;
    auto repack_launch = [&](auto to_packed, const T* inter, T* packed, int64_t rows, int64_t cols, int64_t ld) {
      constexpr bool TO = decltype(to_packed)::value;
      const int64_t grid = ceil_div(rows, 32) * ceil_div(p.batch, 32) * cols;
      if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
      hipLaunchKernelGGL((repack_interleaved_kernel<T, TO>), dim3((unsigned)grid), dim3(256), 0, h->stream,
                         const_cast<T*>(inter), packed, rows, cols, ld, p.batch);
      return hipGetLastError();
    };
    using to_packed_t = std::integral_constant<bool, true>;
    using to_inter_t = std::integral_constant<bool, false>;
    if ((e = repack_launch(to_packed_t{}, p.a, pa, a_rows, a_cols, p.lda)) != hipSuccess) return e;
    if ((e = repack_launch(to_packed_t{}, p.b, pb, b_rows, b_cols, p.ldb)) != hipSuccess) return e;
    if (!p.beta_zero && (e = repack_launch(to_packed_t{}, p.c, pc, p.m, p.n, p.ldc)) != hipSuccess) return e;
    Problem<T, T> q = p;
    q.interleaved = false;
    q.a = pa, q.lda = a_rows, q.stride_a = na;
    q.b = pb, q.ldb = b_rows, q.stride_b = nb;
    q.c = pc, q.ldc = p.m, q.stride_c = nc;
    if ((e = run_gemm_real<T>(h, q)) != hipSuccess) return e;
    return repack_launch(to_inter_t{}, p.c, pc, p.m, p.n, p.ldc);
  }

  const int cus = h->compute_units > 0 ? h->compute_units : 256;

  // One product with K equal to the one-shot K step and many 64 x 64 tiles (rank-64 updates, 8192 x 8192 x 64):
  // bound by writing C.  A fixed set of workgroups walks the tiles with the next tile's operand panels in
  // flight during the current tile's matrix work and C stores (gemm_batch_stream_kernel's tile walk).  Measured
  // at 8192 x 8192 x 64 against the 256-wide tile: beta != 0 (C read and written) fp32 321 -> 136 us, fp64
  // 466 -> 294 us; beta == 0 the same 113 / 220 us (both then run at the clock the chip holds while it writes
  // 2.4 TB/s).  Same k-ordered chain per element, same bits.
  if (h->kernel_override == SBAMD_KERNEL_AUTO && p.batch == 1 && p.k == Tiles<T>::ONESHOT_BK && p.m % 64 == 0 &&
      p.n % 64 == 0 && g.vec_a && g.vec_b && std::max({p.lda, p.ldb, p.ldc}) < (int64_t(1) << 22) &&
      (p.m / 64) * (p.n / 64) >= 2 * int64_t(cus) * Tiles<T>::STREAM_OCC && p.m / 64 < (int64_t(1) << 30) &&
      p.n / 64 < (int64_t(1) << 30)) {
    g.tiles_m = int32_t(p.m / 64);
    g.tiles_n = int32_t(p.n / 64);
    h->last_tile_m = h->last_tile_n = 64;
    h->last_kernel = SBAMD_KERNEL_STREAM;
    h->last_depth = 1;
    h->last_k_chunk = p.k;
    return launch_stream<T, Tiles<T>::ONESHOT_BK, Tiles<T>::STREAM_OCC>(g, p.ta, p.tb, cus, h->stream);
  }

  Plan plan = p.cube ? plan_gemm<T>(p.m, p.n, p.k, p.batch, cus, double(p.cube_bytes)) : plan_gemm<T>(p.m, p.n, p.k, p.batch, cus);
  // A pinned family overrides the planner (testing / tuning hook).
  const int pin_depth = h->kernel_param & 0xffff, pin_tile = (h->kernel_param >> 16) & 0xff;
  const bool pin_fused = (h->kernel_param >> 24) & 1;
  switch (h->kernel_override) {
    case SBAMD_KERNEL_AUTO: break;
    case SBAMD_KERNEL_LARGE: plan = {TILE_128, 1}; break;
    case SBAMD_KERNEL_HUGE: plan = {TILE_256, 1}; break;
    case SBAMD_KERNEL_SMALL: plan = {TILE_64, 1}; break;
    case SBAMD_KERNEL_TINY: plan = {TILE_32, 1}; break;
    case SBAMD_KERNEL_NAIVE: plan = {TILE_NAIVE, 1}; break;
    case SBAMD_KERNEL_MEDIUM:  // parameter bit 0: the 192 x 192 tile
      plan = {(h->kernel_param & 1) ? TILE_192 : TILE_96, 1};
      break;
    case SBAMD_KERNEL_SPLITK:
      if (tile_cuts_k(pin_tile) || (pin_tile == TILE_32 && pin_fused)) plan.tile = pin_tile;
      else if (!tile_cuts_k(plan.tile)) plan.tile = TILE_64;
      if (pin_depth > 0) plan.depth = pin_depth;
      plan.fused = pin_fused;
      if (pin_fused) plan.depth = plan.tile == TILE_32 ? 4 : 2;  // the 32 x 32 tile: four quarters of K in one workgroup
      break;
    default: return hipErrorInvalidValue;
  }
  // four quarters of K in a workgroup of four one-wave sets on the 32 x 32 tile: equal, vector-aligned quarters of
  // at least one K step; anything else pinned that way takes the cube route on the 64 x 64 tile
  const bool quarters = plan.tile == TILE_32 && plan.fused && plan.depth == 4 && p.batch == 1 &&
                        p.k % (4 * Mfma<T>::V) == 0 && p.k / 4 >= Tiles<T>::TINY_BK;
  if (plan.tile == TILE_32 && plan.fused && !quarters) plan = {TILE_64, p.batch == 1 ? 4 : 1, false};
  if (p.batch != 1 || (!tile_cuts_k(plan.tile) && !quarters)) plan.depth = 1;
  if (plan.tile == TILE_NAIVE) {
    h->last_kernel = SBAMD_KERNEL_NAIVE;
    // pinned: one element per lane.  Chosen for a batch of very small items: its 4 x 4 register-block form once
    // an item's C holds 512 bytes (fp32 16^3 x 65536: 162 -> 89 us, fp64 8^3 x 262144: 190 -> 140); below that
    // the block form leaves too few lanes per item to keep their loads on shared cache lines (fp32 4^3: 64 -> 128 us)
    if (h->kernel_override == SBAMD_KERNEL_NAIVE || p.m * p.n * int64_t(sizeof(T)) < 512) return launch_naive<T>(g, p.ta, p.tb, h->stream);
    return launch_block<T>(g, p.ta, p.tb, h->stream);
  }
  if (quarters) {
    GemmArgs<T> gp = g;
    gp.k_chunk = p.k / 4;
    h->last_kernel = SBAMD_KERNEL_SPLITK;
    h->last_depth = 4;
    h->last_k_chunk = gp.k_chunk;
    h->last_tile_m = h->last_tile_n = 32;
    h->last_fused = true;
    return launch_halves<T, 32, 32, Tiles<T>::TINY_BK, 1, 1, 2, Mfma<T>, 4>(gp, p.ta, p.tb, h->stream);
  }
  if (plan.tile == TILE_32) {
    h->last_kernel = SBAMD_KERNEL_TINY;
    h->last_tile_m = h->last_tile_n = 32;
    return launch_tile<T, 32, 32, Tiles<T>::TINY_BK, 1, 1, 4>(g, p.ta, p.tb, h->stream);
  }
  using MS = typename Tiles<T>::MediumShape;
  const bool medium = plan.tile == TILE_96 || plan.tile == TILE_192;
  h->last_tile_m = plan.tile == TILE_256 ? 256 : plan.tile == TILE_128 ? 128 : plan.tile == TILE_96 ? 96 : plan.tile == TILE_192 ? 192 : 64;
  h->last_tile_n = plan.tile == TILE_256 ? Tiles<T>::HUGE_BN : h->last_tile_m;
  auto launch_medium = [&](const GemmArgs<T>& ga) {
    if (plan.tile == TILE_192)
      return launch_tile<T, 192, 192, Tiles<T>::MEDIUM_BK, 2, 4, sizeof(T) == 4 ? 2 : 1, MS>(ga, p.ta, p.tb, h->stream);
    return launch_tile<T, 96, 96, Tiles<T>::MEDIUM_BK, 2, 2, 2, MS>(ga, p.ta, p.tb, h->stream);
  };

  const int bk = plan.tile == TILE_256 ? Tiles<T>::HUGE_BK : plan.tile == TILE_128 ? Tiles<T>::LARGE_BK : medium ? Tiles<T>::MEDIUM_BK : Tiles<T>::SMALL_BK;
  // K slab of one slice: whole K tiles (gemm_partial_local.hpp:174, :253).
  int64_t depth = std::max<int64_t>(1, plan.depth);
  const int64_t k_chunk = std::max<int64_t>(bk, bk * ceil_div(p.k, int64_t(bk) * depth));
  depth = std::max<int64_t>(1, ceil_div(p.k, k_chunk));

  // equal halves whose start stays 16-byte aligned: both sets then run the same steps (and the same K tail)
  if (depth == 2 && plan.fused && plan.tile != TILE_256 && plan.tile != TILE_192 && p.k % (2 * Mfma<T>::V) == 0 && p.k / 2 >= bk) {
    GemmArgs<T> gp = g;
    gp.k_chunk = p.k / 2;
    hipError_t e = plan.tile == TILE_128
                       ? launch_halves<T, 128, 128, Tiles<T>::LARGE_BK, Tiles<T>::LARGE_WM, 2, sizeof(T) == 4 ? 4 : 2>(
                             gp, p.ta, p.tb, h->stream)
                   : plan.tile == TILE_96 ? launch_halves<T, 96, 96, Tiles<T>::MEDIUM_BK, 2, 2, 1, MS>(gp, p.ta, p.tb, h->stream)
                                          : launch_halves<T, 64, 64, Tiles<T>::SMALL_BK, 2, 2, 2>(gp, p.ta, p.tb, h->stream);
    h->last_kernel = SBAMD_KERNEL_SPLITK;
    h->last_depth = 2;
    h->last_k_chunk = gp.k_chunk;
    h->last_fused = true;
    return e;
  }
  if (depth > 1) {
    const int64_t slice = p.m * p.n;
    hipError_t e = hipSuccess;
    T* cube = static_cast<T*>(p.cube);
    if (cube) {
      if (size_t(slice) * depth * sizeof(T) > p.cube_bytes) return hipErrorOutOfMemory;  // only a pinned cut gets here
    } else {
      if ((e = ensure_scratch(h, size_t(slice) * depth * sizeof(T))) != hipSuccess) return e;
      cube = static_cast<T*>(h->scratch);
    }
    GemmArgs<T> gp = g;
    gp.c = cube;
    gp.ldc = p.m;
    gp.split_k = int32_t(depth);
    gp.k_chunk = k_chunk;
    gp.slice_stride = slice;
    gp.beta_zero = 1;
    bool slabs_dma = false;
    if constexpr (sizeof(T) == 4) slabs_dma = plan.tile == TILE_64 && p.k - (depth - 1) * k_chunk >= 64 && dma_ok(gp, p.ta, p.tb, k_chunk, 32);
    if constexpr (sizeof(T) == 4) {
      if (slabs_dma) e = launch_dma_tile<64, 64, 32, 2, 2, 3, 3>(gp, h->stream);
    }
    if (!slabs_dma)
      e = medium                  ? launch_medium(gp)
          : plan.tile == TILE_256 ? launch_huge<T>(gp, p.ta, p.tb, h->stream)
          : plan.tile == TILE_128 ? launch_tile<T, 128, 128, Tiles<T>::LARGE_BK, Tiles<T>::LARGE_WM, 2, 2>(gp, p.ta, p.tb, h->stream)
                                  : launch_tile<T, 64, 64, Tiles<T>::SMALL_BK, 2, 2, Tiles<T>::SMALL_OCC>(gp, p.ta, p.tb, h->stream);
    if (e != hipSuccess) return e;
    const int64_t grid = ceil_div(slice, 256);
    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)grid), dim3(256), 0, h->stream,
                       static_cast<const T*>(cube), slice, int(depth), p.c, p.m, p.n, p.ldc, p.beta,
                       g.beta_zero);
    h->last_kernel = SBAMD_KERNEL_SPLITK;
    h->last_depth = depth;
    h->last_k_chunk = k_chunk;
    return hipGetLastError();
  }

  if (medium) {
    h->last_kernel = SBAMD_KERNEL_MEDIUM;
    return launch_medium(g);
  }
  switch (plan.tile) {
    case TILE_128:
      h->last_kernel = SBAMD_KERNEL_LARGE;
      return launch_tile<T, 128, 128, Tiles<T>::LARGE_BK, Tiles<T>::LARGE_WM, 2, 2>(g, p.ta, p.tb, h->stream);
    case TILE_256:
      h->last_kernel = SBAMD_KERNEL_HUGE;
      return launch_huge<T>(g, p.ta, p.tb, h->stream);
    default:
      h->last_kernel = SBAMD_KERNEL_SMALL;
      // Short K (the batched 64x64x64 class): stage the whole K range at once --
      // one load burst, one barrier, no K loop -- so a CU overlaps the loads of
      // some items with the matrix work of others.
      // More items than resident workgroups: a fixed set of workgroups walks the batch
      // with the next item's operands requested ahead of the current item's matrix work.
      if (p.k == Tiles<T>::ONESHOT_BK && p.m <= 64 && p.n <= 64 && g.vec_a && g.vec_b &&
          (p.ta || p.m % Mfma<T>::V == 0) && (!p.tb || p.n % Mfma<T>::V == 0) &&
          std::max({p.lda, p.ldb, p.ldc}) < (int64_t(1) << 22) &&  // byte offsets inside an item fit 32 bits
          p.batch > int64_t(cus) * Tiles<T>::STREAM_OCC && h->kernel_override == SBAMD_KERNEL_AUTO) {
        h->last_kernel = SBAMD_KERNEL_STREAM;
        return launch_stream<T, Tiles<T>::ONESHOT_BK, Tiles<T>::STREAM_OCC>(g, p.ta, p.tb, cus, h->stream);
      }
      if (p.k <= Tiles<T>::ONESHOT_BK && p.k > Tiles<T>::SMALL_BK)
        return launch_tile<T, 64, 64, Tiles<T>::ONESHOT_BK, 2, 2, sizeof(T) == 4 ? 4 : 2>(g, p.ta, p.tb, h->stream);
      // Operand tiles straight into LDS where the shape allows (fp32; parameter bit 0 of a pinned SMALL
      // family keeps the register-staged kernel, for A/B runs).
      if constexpr (sizeof(T) == 4) {
        if (dma_ok(g, p.ta, p.tb, p.k, 32) && p.k >= 128 && !(h->kernel_override == SBAMD_KERNEL_SMALL && (h->kernel_param & 1)))
          return launch_dma_tile<64, 64, 32, 2, 2, 3, 3>(g, h->stream);
      }
      // At most two workgroups per compute unit (what the deeper stages leave room for): the K step of 64
      // halves the barriers and load waits per matrix instruction, which a unit with so few wavefronts
      // cannot hide (alternating runs: fp64 1024^3 57.7 -> 52.8 us, 768^3 43.0 -> 38.8; fp32 1-2 %; same chain, same bits).
      if (p.k >= 4 * Tiles<T>::ONESHOT_BK && ceil_div(p.m, 64) * ceil_div(p.n, 64) * p.batch <= 2 * int64_t(cus))
        return launch_tile<T, 64, 64, Tiles<T>::ONESHOT_BK, 2, 2, sizeof(T) == 4 ? 4 : 2>(g, p.ta, p.tb, h->stream);
      return launch_tile<T, 64, 64, Tiles<T>::SMALL_BK, 2, 2, Tiles<T>::SMALL_OCC>(g, p.ta, p.tb, h->stream);
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
