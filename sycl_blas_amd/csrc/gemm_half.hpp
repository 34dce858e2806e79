// LDS-tiled MFMA GEMM for 16-bit storage (IEEE half and bfloat16 inputs,
// fp32 accumulation) on gfx950.
//
// The reference instantiates half -> half and half -> float
// (cmake/CmakeFunctionHelper.cmake:148-159) and multiplies mixed types by
// casting both operands to float first (src/operations/blas3/gemm_common.hpp:53-59);
// its only bfloat16 path rounds float data to bf16 while staging and accumulates
// in float (src/operations/blas3/gemm_load_store_joint_matrix.hpp:76-81).  Here
// every 16-bit variant feeds v_mfma_f32_32x32x16_{f16,bf16}: products of two
// 16-bit values are exact in fp32 and the sum is kept in fp32, so the result is
// at least as accurate as the reference's; the output is rounded once at the end.
//
// LDS images keep the operand's own memory order, so staging never transposes:
//   K-contiguous operand (A transposed / B not): [mn][BK + 8], k fastest.  A
//     lane's MFMA fragment (8 consecutive k of one row) is one 16-byte read; the
//     16-byte row padding makes the stride an odd number of 16-byte slots, so the
//     16 lanes of a ds_read_b128 group hit 16 different slots.
//   M/N-contiguous operand: [k][BMN + 32], mn fastest.  The fragment is fetched
//     with two ds_read_b64_tr_b16 (the transposing LDS read of gfx950): each
//     16-lane group reads a 4 (k) x 16 (mn) block and lane i receives the four
//     k values of column i.  The 64-byte row padding places the four rows of a
//     32-lane access on four disjoint quarters of the 64 banks.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_mfma.hpp"

namespace sbamd {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct HalfTag {};  // IEEE binary16 storage
struct Bf16Tag {};  // bfloat16 storage

template <typename Tag>
struct Mfma16;
template <>
struct Mfma16<HalfTag> {
  __device__ static inline f32x16 mma(u16x8 op0, u16x8 op1, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, op0), __builtin_bit_cast(f16x8, op1), c,
                                                  0, 0, 0);
  }
};
template <>
struct Mfma16<Bf16Tag> {
  __device__ static inline f32x16 mma(u16x8 op0, u16x8 op1, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, op0), __builtin_bit_cast(bf16x8, op1),
                                                   c, 0, 0, 0);
  }
};

// Output conversion (one rounding, round-to-nearest-even; NaN stays NaN).
__device__ inline float load_out(const float* p) { return *p; }
__device__ inline float load_out(const uint16_t* p) { return float(__builtin_bit_cast(_Float16, *p)); }
__device__ inline void store_out(float* p, float v) { *p = v; }
__device__ inline void store_out(uint16_t* p, float v) { *p = __builtin_bit_cast(uint16_t, _Float16(v)); }

template <typename TOut>
struct GemmArgs16 {
  const uint16_t* a;
  const uint16_t* b;
  TOut* c;
  int64_t m, n, k;
  int64_t lda, ldb, ldc;
  int64_t stride_a, stride_b, stride_c;
  int32_t batch;
  int32_t tiles_m, tiles_n;
  int32_t vec_a, vec_b;  // 16-byte aligned operand, leading dimension and batch stride (the direct-to-LDS kernel needs it)
  float alpha, beta;
  int32_t beta_zero;
  // K cut (register-staged kernels only): slice z covers k in [z*k_chunk, (z+1)*k_chunk)
  // and writes alpha * (its partial sum) to c + z*slice_stride; 1 = no cut.
  int32_t split_k;
  int64_t k_chunk, slice_stride;
  // Direct-to-LDS kernel: how many indices along M (N) of A (B) may be loaded, when that is more than m (n) --
  // an operand copied into a buffer whose columns are padded to whole 16-byte vectors.  0 = m (n).
  int64_t lim_a = 0, lim_b = 0;
};

// Copy a column-major rows x cols matrix of 16-bit elements (leading dimension ld_src, any alignment) into a
// 16-byte aligned buffer with leading dimension ld_dst (a multiple of 8, >= rows); rows .. ld_dst - 1 are zeros.
__global__ void __launch_bounds__(256)
    copy_pad16_kernel(const uint16_t* __restrict__ src, int64_t ld_src, uint16_t* __restrict__ dst, int64_t ld_dst,
                      int64_t rows, int64_t cols) {
  typedef u16x8 u16x8_any __attribute__((aligned(2)));
  const int64_t per_col = ld_dst / 8;
  const int64_t v = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (v >= per_col * cols) return;
  const int64_t col = v / per_col, r0 = (v - col * per_col) * 8;
  const uint16_t* s = src + col * ld_src + r0;
  u16x8 out = {0, 0, 0, 0, 0, 0, 0, 0};
  if (r0 + 8 <= rows) {
    out = *reinterpret_cast<const u16x8_any*>(s);
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (r0 + e < rows) out[e] = s[e];
  }
  *reinterpret_cast<u16x8*>(dst + col * ld_dst + r0) = out;
}

// Global -> registers -> LDS for one operand tile (BMN x BK, 16-bit elements).
// KMAJOR: k contiguous in global memory (A transposed / B not transposed).
template <int BMN, int BK, int NT, bool KMAJOR>
struct TileLoader16 {
  static constexpr int V = 8;                              // elements per 16-byte vector
  static constexpr int S = KMAJOR ? (BK + 8) : (BMN + 32);  // LDS row stride (elements)
  static constexpr int ELEMS = KMAJOR ? BMN * S : BK * S;
  static constexpr int NVEC = BMN * BK / V;
  static_assert(NVEC % NT == 0, "tile must split evenly over the workgroup");
  static constexpr int NV = NVEC / NT;
  static constexpr int PER_ROW = KMAJOR ? (BK / V) : (BMN / V);

  u16x8 reg[NV];

  // 16-byte loads of any alignment (global memory takes a misaligned vector; the type below tells the
  // compiler so).  EDGE: the tile hangs over the matrix edge.  K-contiguous operand: a row past the edge is
  // pointed at the last valid row.  M/N-contiguous operand: a vector that ends past the edge fetches the
  // last valid vector instead (the matrix must hold one: mn_lim >= 8); store<true> puts the elements of the
  // one vector that straddles the edge where they belong.  What the other clamped positions hold only feeds
  // C elements that are never stored.  The caller guarantees a whole K step (k0 + BK <= k_end).
  typedef u16x8 u16x8_any __attribute__((aligned(2)));
  template <bool EDGE>
  __device__ inline void load_fast(const uint16_t* __restrict__ x, int64_t ld, int64_t mn0, int64_t mn_lim, int64_t k0,
                                   int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + i * NT;
      const int row = v / PER_ROW;
      const int col = (v % PER_ROW) * V;
      if (KMAJOR) {
        int64_t mn = mn0 + row;
        if (EDGE) mn = mn < mn_lim ? mn : mn_lim - 1;
        reg[i] = *reinterpret_cast<const u16x8_any*>(x + (k0 + col) + mn * ld);
      } else {
        int64_t mn = mn0 + col;
        if (EDGE) mn = mn + V <= mn_lim ? mn : mn_lim - V;
        reg[i] = *reinterpret_cast<const u16x8_any*>(x + mn + (k0 + row) * ld);
      }
    }
  }

  // Element loads: rows/columns past the edge re-read the last valid one (they
  // feed only unstored C elements); k past the end contributes exact zeros.
  __device__ inline void load_edge(const uint16_t* __restrict__ x, int64_t ld, int64_t mn0, int64_t mn_lim,
                                   int64_t k0, int64_t k_end, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + i * NT;
      const int row = v / PER_ROW;
      const int col = (v % PER_ROW) * V;
#pragma unroll
      for (int e = 0; e < V; ++e) {
        int64_t mn = mn0 + (KMAJOR ? row : col + e);
        const int64_t kk = k0 + (KMAJOR ? col + e : row);
        mn = mn < mn_lim ? mn : (mn_lim - 1);
        const bool live = kk < k_end;
        const int64_t ks = live ? kk : (k_end - 1);
        const uint16_t val = KMAJOR ? x[ks + mn * ld] : x[mn + ks * ld];
        reg[i][e] = live ? val : uint16_t(0);
      }
    }
  }

  // EDGE: the registers come from load_fast<true>.  The vector of an M/N-contiguous operand that straddles
  // the matrix edge holds the last 8 valid elements, i.e. it starts 8 - (valid elements of the vector)
  // positions early: its elements are stored one by one at their own columns (the leading ones land on what
  // the neighbouring vector stores there too -- the same values).  tile_valid = mn_lim - mn0.
  template <bool EDGE = false>
  __device__ inline void store(SBAMD_LDS uint16_t* lds, int tid, int tile_valid = BMN) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + i * NT;
      const int row = v / PER_ROW;
      const int col = (v % PER_ROW) * V;
      if (EDGE && !KMAJOR && col < tile_valid && col + V > tile_valid) {
        const int first = tile_valid - V;  // tile-relative column of reg[i][0]; negative in a tile narrower than a vector
#pragma unroll
        for (int e = 0; e < V; ++e)
          if (first + e >= 0) lds[row * S + first + e] = reg[i][e];
      } else {
        *reinterpret_cast<SBAMD_LDS u16x8*>(lds + row * S + col) = reg[i];
      }
    }
  }

  // The 8 k-consecutive elements lane `lane` feeds to the MFMA for k-step `s`
  // (16 k per step) and the 32-row block starting at tile-relative index mn0.
  __device__ static inline u16x8 fragment(const SBAMD_LDS uint16_t* lds, int s, int mn0, int lane) {
    const int h = lane >> 5;
    if (KMAJOR) {
      return *reinterpret_cast<const SBAMD_LDS u16x8*>(lds + (mn0 + (lane & 31)) * S + 16 * s + 8 * h);
    } else {
      const int g = (lane >> 4) & 1, i = lane & 15, q = i >> 2, p = i & 3;
      const SBAMD_LDS uint16_t* base = lds + (16 * s + 8 * h + q) * S + mn0 + 16 * g + 4 * p;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((SBAMD_LDS s16x4*)base);
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((SBAMD_LDS s16x4*)(base + 4 * S));
      u16x8 out;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        out[e] = uint16_t(lo[e]);
        out[4 + e] = uint16_t(hi[e]);
      }
      return out;
    }
  }
};

template <typename Tag, typename TOut, int BM, int BN, int BK, int WM, int WN, bool TA, bool TB>
// Two wavefronts per SIMD (256 registers): two 128 x 128 workgroups share a compute unit and cover each other's
// barriers; the gather at ragged edges may spill for it, the interior path does not.
__global__ void __launch_bounds__(WM* WN * 64, 2) gemm_mfma16_kernel(const GemmArgs16<TOut> g) {
  constexpr int NT = WM * WN * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  static_assert(MI >= 1 && NI >= 1 && BK % 16 == 0, "tile shape");
  using LoadA = TileLoader16<BM, BK, NT, TA>;
  using LoadB = TileLoader16<BN, BK, NT, !TB>;
  constexpr int A_ELEMS = LoadA::ELEMS, B_ELEMS = LoadB::ELEMS;
  constexpr int STAGE = A_ELEMS + B_ELEMS;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  SBAMD_LDS uint16_t* smem = (SBAMD_LDS uint16_t*)smem_raw;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;

  const int ntiles = g.tiles_m * g.tiles_n;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int z = logical / ntiles;
  int tm, tn;
  tile_coords(logical - z * ntiles, g.tiles_m, g.tiles_n, tm, tn);
  const int64_t m0 = int64_t(tm) * BM, n0 = int64_t(tn) * BN;
  const uint16_t* a = g.a;
  const uint16_t* b = g.b;
  TOut* c = g.c;
  int64_t k_begin = 0, k_end = g.k;
  if (g.split_k > 1) {
    k_begin = int64_t(z) * g.k_chunk;
    k_end = k_begin + g.k_chunk < g.k ? k_begin + g.k_chunk : g.k;
    c += int64_t(z) * g.slice_stride;
  } else {
    a += int64_t(z) * g.stride_a;
    b += int64_t(z) * g.stride_b;
    c += int64_t(z) * g.stride_c;
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  LoadA la;
  LoadB lb;
  const bool interior = (m0 + BM <= g.m) && (n0 + BN <= g.n);
  const int nk = k_end > k_begin ? int((k_end - k_begin + BK - 1) / BK) : 0;
  const int frag_mn = lane & 31, frag_h = lane >> 5;
  const int a_row = wm * WTM + frag_mn;

  // How a tile's operands travel: whole vectors (interior tile), whole vectors clamped at the matrix edge, or
  // element loads (K tail; a matrix narrower than one vector along a contiguous M / N).
  enum { VEC = 0, VEC_EDGE = 1, ELEMENTS = 2 };
  const bool vectors_fit = (TA || g.m >= 8) && (!TB || g.n >= 8);
  const int valid_m = int(g.m - m0 < BM ? g.m - m0 : BM), valid_n = int(g.n - n0 < BN ? g.n - n0 : BN);
  auto load_tile = [&](int kt) -> int {
    const int64_t k0 = k_begin + int64_t(kt) * BK;
    if (k0 + BK <= k_end && interior) {
      la.template load_fast<false>(a, g.lda, m0, g.m, k0, tid);
      lb.template load_fast<false>(b, g.ldb, n0, g.n, k0, tid);
      return VEC;
    }
    if (k0 + BK <= k_end && vectors_fit) {
      la.template load_fast<true>(a, g.lda, m0, g.m, k0, tid);
      lb.template load_fast<true>(b, g.ldb, n0, g.n, k0, tid);
      return VEC_EDGE;
    }
    la.load_edge(a, g.lda, m0, g.m, k0, k_end, tid);
    lb.load_edge(b, g.ldb, n0, g.n, k0, k_end, tid);
    return ELEMENTS;
  };
  auto store_tile = [&](SBAMD_LDS uint16_t* st, int how) {
    if (how == VEC_EDGE) {
      la.template store<true>(st, tid, valid_m);
      lb.template store<true>(st + A_ELEMS, tid, valid_n);
    } else {
      la.store(st, tid);
      lb.store(st + A_ELEMS, tid);
    }
  };

  if (nk > 0) store_tile(smem, load_tile(0));
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const SBAMD_LDS uint16_t* as = smem + (kt & 1) * STAGE;
    const SBAMD_LDS uint16_t* bs = as + A_ELEMS;
    const bool more = kt + 1 < nk;
    int how = VEC;
    if (more) how = load_tile(kt + 1);
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      u16x8 fa[MI], fb[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[i] = LoadA::fragment(as, s, wm * WTM + i * 32, lane);
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[j] = LoadB::fragment(bs, s, wn * WTN + j * 32, lane);
      // op0 carries the N index, op1 the M index: a lane ends up with one M index
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = Mfma16<Tag>::mma(fb[j], fa[i], acc[i][j]);
    }
    if (more) store_tile(smem + ((kt + 1) & 1) * STAGE, how);
    __syncthreads();
  }

  // epilogue in fp32: out = alpha*acc (+ beta*C), rounded once to the output type
  store_tiles<MI, NI, 16, float>(
      !g.beta_zero, g.beta,
      [&](int i, int j, int r) -> TOut* {
        const int64_t mm = m0 + a_row + i * 32;
        const int64_t nn = n0 + wn * WTN + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * frag_h;
        return (mm < g.m && nn < g.n) ? c + mm + nn * g.ldc : nullptr;
      },
      [&](int i, int j, int r) -> float { return g.alpha * acc[i][j][r]; },
      [](const TOut* p) -> float { return load_out(p); }, [](TOut* p, float v) { store_out(p, v); });
}

// Sum of the K slabs (fp32 cube, slabs added in index order) + beta * C, rounded once to
// the output type: the 16-bit twin of splitk_reduce_kernel.
template <typename TOut>
__global__ void __launch_bounds__(256)
    splitk_reduce16_kernel(const float* __restrict__ cube, int64_t slice_stride, int depth, TOut* c, int64_t m, int64_t n,
                           int64_t ldc, float beta, int beta_zero) {
  const int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= m * n) return;
  const float* src = cube + idx;
  float sum = *src;
  for (int d = 1; d < depth; ++d) sum = sum + src[int64_t(d) * slice_stride];
  TOut* p = c + (idx % m) + (idx / m) * ldc;
  if (!beta_zero) {
    const float bc = beta * load_out(p);
    sum = sum + bc;
  }
  store_out(p, sum);
}

// Storage bits -> float (exact for both formats).
__device__ inline float widen(HalfTag, uint16_t v) { return float(__builtin_bit_cast(_Float16, v)); }
__device__ inline float widen(Bf16Tag, uint16_t v) { return __uint_as_float(uint32_t(v) << 16); }

// Interleaved batches of 16-bit items: element (r, c) of item b at (r + c*ld)*batch + b
// (test/unittest/blas3/blas3_gemm_common.hpp:54-69).  Same structure as the fp32 / fp64
// gemm_interleaved_kernel -- consecutive lanes take consecutive items, an IR x IC register
// block of C per lane, no LDS (src/operations/blas3/gemm_interleaved.hpp:104-537) -- with
// both operands widened to float before the fused multiply-add, the mixed-precision rule of
// src/operations/blas3/gemm_common.hpp:53-59, and k walked in increasing order: every C
// element is the fp32 chain of the widened inputs, rounded once to the output type.
template <typename Tag, typename TOut, bool TA, bool TB, int IR, int IC>
__global__ void __launch_bounds__(256) gemm_interleaved16_kernel(const GemmArgs16<TOut> g) {
  const int64_t nb = g.batch;
  const int64_t tiles_r = (g.m + IR - 1) / IR;
  const int64_t tiles_c = (g.n + IC - 1) / IC;
  const int64_t gid = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (gid >= nb * tiles_r * tiles_c) return;
  const int64_t b = gid % nb;
  const int64_t t = gid / nb;
  const int64_t r0 = (t % tiles_r) * IR;
  const int64_t c0 = (t / tiles_r) * IC;

  float acc[IR][IC];
#pragma unroll
  for (int i = 0; i < IR; ++i)
#pragma unroll
    for (int j = 0; j < IC; ++j) acc[i][j] = 0.0f;

  // clamped rows / columns only feed C elements that are never stored
  const uint16_t* pa[IR];
  const uint16_t* pb[IC];
#pragma unroll
  for (int i = 0; i < IR; ++i) {
    const int64_t r = r0 + i < g.m ? r0 + i : g.m - 1;
    pa[i] = g.a + (TA ? r * g.lda : r) * nb + b;
  }
#pragma unroll
  for (int j = 0; j < IC; ++j) {
    const int64_t c = c0 + j < g.n ? c0 + j : g.n - 1;
    pb[j] = g.b + (TB ? c : c * g.ldb) * nb + b;
  }
  const int64_t step_a = (TA ? int64_t(1) : g.lda) * nb;
  const int64_t step_b = (TB ? g.ldb : int64_t(1)) * nb;
  for (int64_t k = 0; k < g.k; ++k) {
    float av[IR], bv[IC];
#pragma unroll
    for (int i = 0; i < IR; ++i) {
      av[i] = widen(Tag{}, *pa[i]);
      pa[i] += step_a;
    }
#pragma unroll
    for (int j = 0; j < IC; ++j) {
      bv[j] = widen(Tag{}, *pb[j]);
      pb[j] += step_b;
    }
#pragma unroll
    for (int i = 0; i < IR; ++i)
#pragma unroll
      for (int j = 0; j < IC; ++j) acc[i][j] = __builtin_fmaf(av[i], bv[j], acc[i][j]);
  }

#pragma unroll
  for (int j = 0; j < IC; ++j) {
#pragma unroll
    for (int i = 0; i < IR; ++i) {
      if (r0 + i < g.m && c0 + j < g.n) {
        TOut* p = g.c + ((r0 + i) + (c0 + j) * g.ldc) * nb + b;
        float out = g.alpha * acc[i][j];
        if (!g.beta_zero) {
          const float bc = g.beta * load_out(p);
          out = out + bc;
        }
        store_out(p, out);
      }
    }
  }
}

// Strided batches of very small 16-bit items (at most 16 x 16): the lanes of an item are neighbours, each owns an
// IR x IC block of its C (1 x 1 or 4 x 4), operands widened to float, k in increasing order with fused multiply-adds:
// the fp32 chain of the widened inputs rounded once, as in gemm_interleaved16_kernel.  A 64 x 64 matrix-instruction
// tile per item would be almost all padding (half 16^3 x 65536: 218 us on it).
template <typename Tag, typename TOut, bool TA, bool TB, int IR, int IC>
__global__ void __launch_bounds__(256) gemm_block16_kernel(const GemmArgs16<TOut> g) {
  const int64_t tiles_r = (g.m + IR - 1) / IR;
  const int64_t tiles_c = (g.n + IC - 1) / IC;
  const int64_t per = tiles_r * tiles_c;
  const int64_t gid = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (gid >= per * g.batch) return;
  const int64_t z = gid / per;
  const int64_t t = gid - z * per;
  const int64_t r0 = (t % tiles_r) * IR;
  const int64_t c0 = (t / tiles_r) * IC;

  float acc[IR][IC];
#pragma unroll
  for (int i = 0; i < IR; ++i)
#pragma unroll
    for (int j = 0; j < IC; ++j) acc[i][j] = 0.0f;

  // clamped rows / columns only feed C elements that are never stored
  const uint16_t* pa[IR];
  const uint16_t* pb[IC];
#pragma unroll
  for (int i = 0; i < IR; ++i) {
    const int64_t r = r0 + i < g.m ? r0 + i : g.m - 1;
    pa[i] = g.a + z * g.stride_a + (TA ? r * g.lda : r);
  }
#pragma unroll
  for (int j = 0; j < IC; ++j) {
    const int64_t c = c0 + j < g.n ? c0 + j : g.n - 1;
    pb[j] = g.b + z * g.stride_b + (TB ? c : c * g.ldb);
  }
  const int64_t step_a = TA ? int64_t(1) : g.lda;
  const int64_t step_b = TB ? g.ldb : int64_t(1);
  for (int64_t k = 0; k < g.k; ++k) {
    float av[IR], bv[IC];
#pragma unroll
    for (int i = 0; i < IR; ++i) {
      av[i] = widen(Tag{}, *pa[i]);
      pa[i] += step_a;
    }
#pragma unroll
    for (int j = 0; j < IC; ++j) {
      bv[j] = widen(Tag{}, *pb[j]);
      pb[j] += step_b;
    }
#pragma unroll
    for (int i = 0; i < IR; ++i)
#pragma unroll
      for (int j = 0; j < IC; ++j) acc[i][j] = __builtin_fmaf(av[i], bv[j], acc[i][j]);
  }

#pragma unroll
  for (int j = 0; j < IC; ++j) {
#pragma unroll
    for (int i = 0; i < IR; ++i) {
      if (r0 + i < g.m && c0 + j < g.n) {
        TOut* p = g.c + z * g.stride_c + (r0 + i) + (c0 + j) * g.ldc;
        float out = g.alpha * acc[i][j];
        if (!g.beta_zero) {
          const float bc = g.beta * load_out(p);
          out = out + bc;
        }
        store_out(p, out);
      }
    }
  }
}

// C = beta * C for 16-bit output (alpha == 0 path): multiply in fp32, round once.
__global__ void __launch_bounds__(256)
    scale_matrix16_kernel(uint16_t* c, int64_t rows, int64_t cols, int64_t ld, int64_t stride, int64_t batch,
                          float beta, int assign_zero) {
  const int64_t per = rows * cols;
  const int64_t gid = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (gid >= per * batch) return;
  const int64_t z = gid / per;
  const int64_t idx = gid - z * per;
  uint16_t* p = c + z * stride + (idx % rows) + (idx / rows) * ld;
  if (assign_zero) *p = 0;
  else store_out(p, beta * load_out(p));
}

}  // namespace sbamd
