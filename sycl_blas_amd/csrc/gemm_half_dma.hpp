// MFMA GEMM for 16-bit storage whose operand tiles go from global memory straight into
// LDS (global_load_lds_dwordx4, no register staging) through a ring of five LDS stages:
// 256 x 256 tiles on eight wavefronts (one workgroup per compute unit) and 128 x 128 tiles
// on four (two workgroups per unit, for shapes with too few big tiles).
//
// Same arithmetic as gemm_mfma16_kernel (gemm_half.hpp): v_mfma_f32_32x32x16, fp32
// accumulation in k order, one rounding at the end -- the two kernels agree bit for
// bit.  What changes is the data movement, sized for one 8-wave workgroup per
// compute unit:
//   * in the big tile each wave owns a 128 (M) x 64 (N) block of C: 6 fragment reads feed 8
//     matrix instructions per 16 k, a third of the LDS read traffic per flop of the 64 x 64
//     wave tile (which runs the LDS at its limit; the 128 x 128 tile lives with that);
//   * a load instruction writes 64 lanes x 16 bytes = 1 KiB of LDS contiguously, so
//     the LDS image is linear and the bank-conflict-free arrangement is produced by
//     permuting WHICH 16-byte piece of global memory each lane fetches; fragment
//     reads apply the same permutation (an involution) to their address.
// Requirements (checked by the launcher): K >= 32 (a remainder of K modulo 32 is one
// zero-padded, register-staged tile behind the ring), 16-byte aligned operands with
// leading dimensions that are multiples of 8, and M (N) a multiple of 8 when it is the
// contiguous index of A (B).  Tiles that hang over the matrix
// edge fetch clamped rows/columns, which only feed C elements that are never stored.
#pragma once

#include <type_traits>
#include <utility>

#include "gemm_half.hpp"

namespace sbamd {

constexpr int DMA_BM = 256, DMA_BN = 256, DMA_NT = 512;  // the big tile

// f(integral_constant<int, 0>) .. f(integral_constant<int, N - 1>), in order
template <int... Is, typename F>
__device__ inline void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ inline void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

typedef __attribute__((address_space(1))) const void* dma_gptr_t;
typedef __attribute__((address_space(3))) void* dma_lptr_t;

// K steps of 32 through a ring of five LDS stages (all 160 KiB), four tiles in flight.  The loads of tile t + 4 are issued during step t, two pieces
// behind the matrix instructions of each half step, and are only waited for -- with a
// counted s_waitcnt, never a full drain -- in the middle of step t + 3, so a load has
// about three K steps to land.  One raw s_barrier per K step, placed before the
// matrix instructions of the second half step: by then every wave has finished its
// fragment reads of tile t (the stage may be refilled right after the barrier) and
// has seen its own pieces of tile t + 1 land (the first fragments of tile t + 1 are
// fetched behind the remaining matrix instructions).
constexpr int RING_BK = 32, RING_STAGES = 5, RING_AHEAD = 4;
constexpr int ring_operand_bytes(int rows) { return rows * RING_BK * 2; }  // 256 rows: 16 KiB
constexpr int ring_lds_bytes(int bm, int bn) { return RING_STAGES * (ring_operand_bytes(bm) + ring_operand_bytes(bn)); }
constexpr int RING_LDS_BYTES = ring_lds_bytes(DMA_BM, DMA_BN);

// Images per stage and operand (ROWS = 256 or 128 indices along M / N; 16 or 8 KiB):
//   K-contiguous:   [ROWS mn][32 k], 64-byte rows of four 16-byte pieces; piece c of row
//                   r sits at position c ^ ((r >> 2) & 3): the 16 rows of a ds_read_b128
//                   lane group cover the 16 slots of the bank row.
//   M/N-contiguous: [32 k][ROWS mn], rows of ROWS / 8 pieces (512 or 256 bytes, whole
//                   turns of the 64 banks); piece c of row k sits at position
//                   c ^ ((k & 3) << 2): the four k rows that one ds_read_b64_tr_b16 lane
//                   group touches land on four different 64-byte bank segments.
// ds_read_b64_tr_b16 as inline assembly: behind the builtin the compiler drains every
// outstanding LDS-writing load (s_waitcnt vmcnt(0)) before each transposing read,
// which would serialise the ring.  The caller waits (lgkmcnt) before using the result.
template <int OFF>
__device__ inline s16x4 ds_read_tr16(uint32_t addr) {
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

template <int SHAPE>
__device__ inline int ring_swizzle_k(int row) {
  return (row >> 2) & 3;
}
template <int SHAPE>
__device__ inline int ring_swizzle_mn(int krow) {
  return (krow & 3) << 2;
}

template <bool KMAJOR, int ROWS = 256, int NT = DMA_NT, int SHAPE = 32>
struct RingOperand {
  static constexpr int NP = ROWS * 4 / NT;  // 16-byte pieces per thread and operand tile
  static_assert(ROWS * 4 == NP * NT && (NP == 2 || NP == 4), "two or four 16-byte pieces per thread and operand tile");
  static_assert(ROWS % 128 == 0, "the M/N-contiguous swizzle moves a piece by up to 12 positions inside its row");
  static constexpr int PER_KROW = ROWS / 8;  // pieces per k row of the M/N-contiguous image
  static constexpr int BYTES = ring_operand_bytes(ROWS);
  const uint16_t* src[NP];

  __device__ inline void init(const uint16_t* __restrict__ x, int64_t ld, int64_t mn0, int64_t mn_lim, int tid) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int L = i * NT + tid;  // ROWS * 4 pieces per operand tile
      if (KMAJOR) {
        const int row = L >> 2;
        const int c = (L & 3) ^ ring_swizzle_k<SHAPE>(row);
        int64_t mn = mn0 + row;
        mn = mn < mn_lim ? mn : mn_lim - 1;
        src[i] = x + mn * ld + c * 8;
      } else {
        const int krow = L / PER_KROW;
        const int c = (L % PER_KROW) ^ ring_swizzle_mn<SHAPE>(krow);
        int64_t mn = mn0 + c * 8;
        mn = mn < mn_lim ? mn : mn_lim - 8;
        src[i] = x + mn + int64_t(krow) * ld;
      }
    }
  }

  // Piece i of this thread for the tile the pointers address.
  template <int I>
  __device__ inline void issue_piece(SBAMD_LDS unsigned char* lds, int wave) {
    SBAMD_LDS unsigned char* dst = lds + (I * NT + wave * 64) * 16;
    __builtin_amdgcn_global_load_lds((dma_gptr_t)src[I], (dma_lptr_t)dst, 16, 0, 0);
  }
  __device__ inline void advance(int64_t ld) {
    const int64_t step = KMAJOR ? int64_t(RING_BK) : int64_t(RING_BK) * ld;
#pragma unroll
    for (int i = 0; i < NP; ++i) src[i] += step;
  }

  // Piece I of the K tail (k_rem = 1 .. 31 valid k behind the last whole K step, which is where the
  // pointers stand after the main loop): through registers, zeros past the end of K.
  template <int I>
  __device__ inline u16x8 load_tail(int k_rem, int tid) const {
    const int L = I * NT + tid;
    u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (KMAJOR) {
      const int row = L >> 2;
      const int kl = ((L & 3) ^ ring_swizzle_k<SHAPE>(row)) * 8;
      if (kl + 8 <= k_rem) {
        v = *reinterpret_cast<const u16x8*>(src[I]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (kl + e < k_rem) v[e] = src[I][e];
      }
    } else {
      if (L / PER_KROW < k_rem) v = *reinterpret_cast<const u16x8*>(src[I]);
    }
    return v;
  }
  template <int I>
  __device__ static inline void store_tail(SBAMD_LDS unsigned char* lds, int tid, u16x8 v) {
    *reinterpret_cast<SBAMD_LDS u16x8*>(lds + (I * NT + tid) * 16) = v;
  }

  // Fragment of k-step S (16 k) for the 32 rows starting at tile-relative mn0.
  template <int S>
  __device__ static inline u16x8 fragment(const SBAMD_LDS unsigned char* lds, int mn0, int lane) {
    const int h = lane >> 5;
    if (KMAJOR) {
      const int row = mn0 + (lane & 31);
      const int c = (2 * S + h) ^ ((row >> 2) & 3);
      return *reinterpret_cast<const SBAMD_LDS u16x8*>(lds + row * 64 + c * 16);
    } else {
      const int g = (lane >> 4) & 1, i = lane & 15, q = i >> 2, p = i & 3;
      const int col = mn0 + 16 * g + 4 * p;
      const int c = (col >> 3) ^ (q << 2);
      constexpr int ROW_BYTES = ROWS * 2;
      const uint32_t addr = uint32_t(uintptr_t(lds)) + uint32_t((8 * h + q) * ROW_BYTES + c * 16 + (col & 7) * 2);
      const s16x4 lo = ds_read_tr16<S * 16 * ROW_BYTES>(addr);
      const s16x4 hi = ds_read_tr16<S * 16 * ROW_BYTES + 4 * ROW_BYTES>(addr);
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

// ABLATE is a timing-only diagnostic (bit 1: no loads inside the K loop, bit 2: no matrix
// instructions, bit 4: no fragment reads); the library instantiates 0.
// BM x BN tile on WM x WN wavefronts: 256 x 256 on 2 x 4 (wave tile 128 x 64) or on 2 x 2 (wave tile 128 x 128: one
// wavefront per SIMD with 256 accumulator registers, a third fewer fragment bytes per flop), 128 x 128 on 2 x 2 (64 x 64).
template <typename Tag, typename TOut, bool TA, bool TB, int ABLATE = 0, int BM = DMA_BM, int BN = DMA_BN, int WM = 2, int WN = 4>
__global__ void __launch_bounds__(WM* WN * 64, ((BM / WM) * (BN / WN) > 128 * 64 ? 1 : 2)) gemm_mfma16_ring_kernel(const GemmArgs16<TOut> g) {
  constexpr int NT = WM * WN * 64;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  constexpr int NM = MI * NI;  // matrix instructions per half step
  using OpA = RingOperand<TA, BM, NT>;
  using OpB = RingOperand<!TB, BN, NT>;
  constexpr int NP = OpA::NP;   // load instructions per thread, operand and tile
  constexpr int LPH = NP / 2;   // of them per half step
  static_assert(OpA::NP == OpB::NP, "square tiles");
  constexpr int RING_OPERAND_BYTES = OpA::BYTES;  // offset of the B image inside a stage
  constexpr int RING_STAGE_BYTES = OpA::BYTES + OpB::BYTES;

  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
  SBAMD_LDS unsigned char* smem = (SBAMD_LDS unsigned char*)smem_raw;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // the wavefront's index is uniform: kept in a scalar register, so is everything derived from it
  // (the LDS destination of each load piece reaches M0 without a detour through vector registers:
  // 69 -> 24 vector instructions beside the 80 matrix instructions of one trip of the main loop)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  int64_t k_len = g.k;
  if (g.split_k > 1) {
    // K slab z (k_chunk is a multiple of the K step, so only the last slab can end in a tail): alpha * (its
    // partial sum) goes to slice z of the cube
    const int64_t k_begin = int64_t(z) * g.k_chunk;
    k_len = g.k - k_begin < g.k_chunk ? g.k - k_begin : g.k_chunk;
    a += TA ? k_begin : k_begin * g.lda;
    b += TB ? k_begin * g.ldb : k_begin;
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

  OpA oa;
  OpB ob;
  oa.init(a, g.lda, m0, g.lim_a > 0 ? g.lim_a : g.m, tid);
  ob.init(b, g.ldb, n0, g.lim_b > 0 ? g.lim_b : g.n, tid);
  const int nk = int(k_len / RING_BK);

  auto stage_ptr = [&](int t) { return smem + (t % RING_STAGES) * RING_STAGE_BYTES; };

  // prologue: tiles 0 .. RING_AHEAD-1 (those that exist) go in flight
#pragma unroll
  for (int t = 0; t < RING_AHEAD; ++t) {
    if (t < nk) {
      SBAMD_LDS unsigned char* st = smem + t * RING_STAGE_BYTES;
      static_for<NP>([&](auto i) { oa.template issue_piece<decltype(i)::value>(st, wave); });
      static_for<NP>([&](auto i) { ob.template issue_piece<decltype(i)::value>(st + RING_OPERAND_BYTES, wave); });
      oa.advance(g.lda);
      ob.advance(g.ldb);
    }
  }
  // tile 0 must have landed for every wave before anyone reads it (three more tiles of 2 * NP loads may be in flight)
  if (nk > RING_AHEAD - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * NP) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  u16x8 fa[2][MI], fb[2][NI];
  // Read unit U of a half step: the B fragments first, then the A fragments.
  auto read_unit = [&](auto unit_tag, auto s_tag, const SBAMD_LDS unsigned char* st, u16x8 (&xa)[MI], u16x8 (&xb)[NI]) {
    constexpr int U = decltype(unit_tag)::value, S = decltype(s_tag)::value;
    if constexpr (U < NI) xb[U] = OpB::template fragment<S>(st + RING_OPERAND_BYTES, wn * WTN + U * 32, lane);
    else xa[U - NI] = OpA::template fragment<S>(st, wm * WTM + (U - NI) * 32, lane);
  };
  using yes = std::integral_constant<bool, true>;
  using no = std::integral_constant<bool, false>;

  // One half step with its issue order written out: matrix instruction n, then read
  // unit n of the NEXT half step (six units for eight matrix instructions), a load
  // piece behind the third and the sixth.  sched_barrier(0) keeps the compiler from
  // regrouping; the operands of the matrix instructions were waited for before the
  // last barrier, so nothing in here stalls on memory.
  auto half_step = [&](auto s_next_tag, auto loads_tag, auto reads_tag, auto piece_tag, const SBAMD_LDS unsigned char* st_next,
                       SBAMD_LDS unsigned char* fill, u16x8 (&ca)[MI], u16x8 (&cb)[NI], u16x8 (&na)[MI], u16x8 (&nb)[NI]) {
    constexpr bool LOADS = decltype(loads_tag)::value, READS = decltype(reads_tag)::value;
    constexpr int PIECE = decltype(piece_tag)::value;
    auto slot = [&](auto n_tag) {
      constexpr int N = decltype(n_tag)::value;
      constexpr int I = N / NI, J = N % NI;
      if constexpr (!(ABLATE & 2)) acc[I][J] = Mfma16<Tag>::mma(cb[J], ca[I], acc[I][J]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (READS && N < MI + NI && !(ABLATE & 4)) read_unit(std::integral_constant<int, N>{}, s_next_tag, st_next, na, nb);
      if constexpr (LOADS && !(ABLATE & 1)) {
        // this half step's load instructions (A's LPH pieces, then B's) behind matrix instructions 2 and 5 of
        // eight, 1 and 3 of four, 2 / 6 / 10 / 14 of sixteen
        static_for<2 * LPH>([&](auto l_tag) {
          constexpr int L = decltype(l_tag)::value;
          constexpr int AT = NP == 2 ? (L == 0 ? NM / 4 : NM / 2 + NM / 4 - (NM > 4 ? 1 : 0)) : (2 * L + 1) * NM / (4 * LPH);
          if constexpr (N == AT) {
            if constexpr (L < LPH) oa.template issue_piece<PIECE * LPH + L>(fill, wave);
            else ob.template issue_piece<PIECE * LPH + L - LPH>(fill + RING_OPERAND_BYTES, wave);
          }
        });
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    static_for<NM>(slot);
  };
  static_assert((NM == 16 || NM == 8 || NM == 4) && MI + NI <= NM, "wave tiles of 128 x 128, 128 x 64 and 64 x 64");

  // first fragments of tile 0
  {
    static_for<MI + NI>([&](auto u) { read_unit(u, std::integral_constant<int, 0>{}, smem, fa[0], fb[0]); });
  }

  // One K step.  LOADS: tile t + RING_AHEAD exists and is issued here; NEXT: tile
  // t + 1 exists.
  auto step = [&](const SBAMD_LDS unsigned char* cur, const SBAMD_LDS unsigned char* next, SBAMD_LDS unsigned char* fill,
                  auto loads_tag, auto next_tag) {
    constexpr bool LOADS = decltype(loads_tag)::value;

    // the fragments fetched during the previous half step must have arrived (the
    // transposing reads are inline assembly: the compiler does not wait for them)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    // first half step: matrix work on k 0..15, fragments of k 16..31, piece 0 of A and B
    half_step(std::integral_constant<int, 1>{}, loads_tag, yes{}, std::integral_constant<int, 0>{}, cur, fill, fa[0], fb[0],
              fa[1], fb[1]);

    // every fragment read of tile t has returned; own pieces of tile t + 1 have landed
    // (in flight and allowed to stay so: tiles t + 2, t + 3 and the first half of tile t + 4, 5 * NP loads)
    if (LOADS && !(ABLATE & 1)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(5 * NP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // second half step: matrix work on k 16..31, first fragments of tile t + 1, piece 1
    half_step(std::integral_constant<int, 0>{}, loads_tag, next_tag, std::integral_constant<int, 1>{}, next, fill, fa[1],
              fb[1], fa[0], fb[0]);
    if (LOADS) {
      oa.advance(g.lda);
      ob.advance(g.ldb);
    }
  };
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the first fragments
  __builtin_amdgcn_sched_barrier(0);

  // Steady state: five steps per trip, one per ring stage, so that every LDS address is
  // the lane's offset plus a compile-time constant.
  int t = 0;
  for (; t + RING_STAGES - 1 + RING_AHEAD < nk; t += RING_STAGES) {
    step(smem + 0 * RING_STAGE_BYTES, smem + 1 * RING_STAGE_BYTES, smem + 4 * RING_STAGE_BYTES, yes{}, yes{});
    step(smem + 1 * RING_STAGE_BYTES, smem + 2 * RING_STAGE_BYTES, smem + 0 * RING_STAGE_BYTES, yes{}, yes{});
    step(smem + 2 * RING_STAGE_BYTES, smem + 3 * RING_STAGE_BYTES, smem + 1 * RING_STAGE_BYTES, yes{}, yes{});
    step(smem + 3 * RING_STAGE_BYTES, smem + 4 * RING_STAGE_BYTES, smem + 2 * RING_STAGE_BYTES, yes{}, yes{});
    step(smem + 4 * RING_STAGE_BYTES, smem + 0 * RING_STAGE_BYTES, smem + 3 * RING_STAGE_BYTES, yes{}, yes{});
  }
  // the remaining steps (t is a multiple of five here) with the stage computed at run time
  for (; t + RING_AHEAD < nk; ++t) step(stage_ptr(t), stage_ptr(t + 1), stage_ptr(t + RING_AHEAD), yes{}, yes{});
  for (; t + 1 < nk; ++t) step(stage_ptr(t), stage_ptr(t + 1), stage_ptr(t + RING_AHEAD), no{}, yes{});
  if (t < nk) step(stage_ptr(t), stage_ptr(t + 1), stage_ptr(t + RING_AHEAD), no{}, no{});

  // K tail (k % 32 valid k): one more tile, zero-padded, staged through registers into stage 0 once every
  // wavefront has finished the last ring step; the zeros add nothing to the sums.
  const int k_rem = int(k_len - int64_t(nk) * RING_BK);
  if (k_rem > 0) {
    __syncthreads();
    u16x8 ta[NP], tb[NP];
    static_for<NP>([&](auto i) { ta[decltype(i)::value] = oa.template load_tail<decltype(i)::value>(k_rem, tid); });
    static_for<NP>([&](auto i) { tb[decltype(i)::value] = ob.template load_tail<decltype(i)::value>(k_rem, tid); });
    static_for<NP>([&](auto i) { OpA::template store_tail<decltype(i)::value>(smem, tid, ta[decltype(i)::value]); });
    static_for<NP>([&](auto i) { OpB::template store_tail<decltype(i)::value>(smem + RING_OPERAND_BYTES, tid, tb[decltype(i)::value]); });
    __syncthreads();
    auto tail_half = [&](auto s_tag) {
      static_for<MI + NI>([&](auto u) { read_unit(u, s_tag, smem, fa[0], fb[0]); });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the transposing reads are not tracked by the compiler
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = Mfma16<Tag>::mma(fb[0][j], fa[0][i], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    };
    tail_half(std::integral_constant<int, 0>{});
    if (k_rem > 16) tail_half(std::integral_constant<int, 1>{});
  }

  const int frag_mn = lane & 31, frag_h = lane >> 5;
  auto value = [&](int i, int j, int r) -> float { return g.alpha * acc[i][j][r]; };
  if (m0 + BM <= g.m && n0 + BN <= g.n && g.ldc * int64_t(WTN * sizeof(TOut)) < (int64_t(1) << 31)) {
    store_tiles_interior<MI, NI, 16, 32, float, TOut>(
        !g.beta_zero, g.beta, c + (m0 + wm * WTM) + (n0 + wn * WTN) * g.ldc, g.ldc, frag_mn, 4 * frag_h,
        [](int r) { return (r & 3) + 8 * (r >> 2); }, value);
    return;
  }
  store_tiles<MI, NI, 16, float>(
      !g.beta_zero, g.beta,
      [&](int i, int j, int r) -> TOut* {
        const int64_t mm = m0 + wm * WTM + i * 32 + frag_mn;
        const int64_t nn = n0 + wn * WTN + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * frag_h;
        return (mm < g.m && nn < g.n) ? c + mm + nn * g.ldc : nullptr;
      },
      value, [](const TOut* p) -> float { return load_out(p); }, [](TOut* p, float v) { store_out(p, v); });
}


}  // namespace sbamd
