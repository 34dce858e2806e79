// LDS-tiled MFMA GEMM for gfx950 (fp32 and fp64 storage).
//
// One workgroup of WM*WN wavefronts owns a BM x BN tile of C and walks K in
// steps of BK through two LDS buffers.  Both operand tiles are kept in LDS as
// [k][mn] images (the M or N index fastest), so a wavefront's MFMA fragment is
// one conflict-free row read per lane and the matrix instruction consumes k in
// increasing order: the accumulator of every C element is the plain
// fused-multiply-add chain over k = 0, 1, 2, ... that the reference kernels
// build with sycl::mad (src/operations/blas3/gemm_ref.hpp:234-251,
// src/operations/blas3/gemm_local.hpp:737-773), which makes the result
// reproducible bit for bit on the CPU.
//
// The role this kernel plays in the reference is that of
// Gemm<..., local, standard, ...> (src/operations/blas3/gemm_local.hpp:74-875)
// and, with SPLIT-K slices, GemmPartial (gemm_partial_local.hpp:32-457).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "gemm_kloop_asm.hpp"

namespace sbamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// Pointers into LDS carry their address space so every access is a ds_*
// instruction (a generic pointer would fall back to flat loads).
#define SBAMD_LDS __attribute__((address_space(3)))

// Timing-only diagnostic (tools/ablate.hip): bits switch off parts of the
// pipelined main loop so their cost can be read from the run time.  The
// results of such a build are wrong by construction; the library is always
// built with 0.   1: LDS stores  2: global loads  4: barrier  8: fragment reads
#ifndef SBAMD_ABLATE
#define SBAMD_ABLATE 0
#endif
// Wavefront tiles of at most this many matrix tiles take the K loop with fragments two groups ahead and the
// operand traffic spread over the step (run_pipelined_far).
#ifndef SBAMD_FAR_MAX_TILES
#define SBAMD_FAR_MAX_TILES 1
#endif
// Lab switch (tools/step_lab.hip): 0 leaves the part of a K step behind its barrier to the compiler's scheduler
// (the state before the fences and the pattern there were added), 1 is what the library is built with.
#ifndef SBAMD_LAB_VARIANT
#define SBAMD_LAB_VARIANT 1
#endif
// 1: interior tiles of the tile shapes that have one run their steady-state K loop as the hand-scheduled
// assembly block of gemm_kloop_asm.hpp; 0: the compiled loop everywhere (same bits either way).
#ifndef SBAMD_KLOOP_ASM
#define SBAMD_KLOOP_ASM 1
#endif

// Arguments shared by every kernel in this file.  Pointers are typed in the
// kernels; sizes are 64-bit so 65536 x 65536 operands index correctly.
template <typename T>
struct GemmArgs {
  const T* a;
  const T* b;
  T* c;
  int64_t m, n, k;
  int64_t lda, ldb, ldc;
  int64_t stride_a, stride_b, stride_c;  // per batch item (elements)
  int32_t batch;                         // >= 1
  int32_t split_k;                       // >= 1; > 1 only with batch == 1
  int64_t k_chunk;                       // K range of one split-K slice
  int64_t slice_stride;                  // elements between slices of the cube
  int32_t tiles_m, tiles_n;
  int32_t vec_a, vec_b;                  // A / B: base, leading dimension and batch stride are 16-byte aligned (direct-to-LDS staging needs it)
  T alpha, beta;
  int32_t beta_zero;                     // never read C
  int32_t group_m;                       // rows of tiles per band of the workgroup -> tile map (0: the default)
};

// ---------------------------------------------------------------- MFMA shapes
template <typename T>
struct Mfma;

// v_mfma_f32_32x32x2_f32: lane l feeds op0[i = l & 31][k = l >> 5] and
// op1[k = l >> 5][j = l & 31]; it receives D[i][j = l & 31] for
// i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), r = 0..15.
template <>
struct Mfma<float> {
  static constexpr int T = 32;  // square tile edge
  static constexpr int TK = 2;  // k consumed per instruction
  static constexpr int NACC = 16;
  using acc_t = f32x16;
  using vec_t = f32x4;
  static constexpr int V = 4;  // elements per 16-byte vector
  __device__ static inline acc_t mma(float op0, float op1, acc_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(op0, op1, c, 0, 0, 0);
  }
  __device__ static inline int lane_mn(int lane) { return lane & 31; }
  __device__ static inline int lane_k(int lane) { return lane >> 5; }
  __device__ static inline int acc_i(int r, int lane) {
    return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
  }
  static constexpr int GS = 2;      // MFMA k-steps served by one fragment fetch
  static constexpr int PAD_MN = 0;  // [k][mn] image: 32 consecutive lanes, 32 consecutive banks
};

// v_mfma_f64_16x16x4_f64: lane l feeds op0[i = l & 15][k = l >> 4] and
// op1[k = l >> 4][j = l & 15]; it receives D[i][j = l & 15] for
// i = (l >> 4) + 4 * r, r = 0..3.
template <>
struct Mfma<double> {
  static constexpr int T = 16;
  static constexpr int TK = 4;
  static constexpr int NACC = 4;
  using acc_t = f64x4;
  using vec_t = f64x2;
  static constexpr int V = 2;
  __device__ static inline acc_t mma(double op0, double op1, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(op0, op1, c, 0, 0, 0);
  }
  __device__ static inline int lane_mn(int lane) { return lane & 15; }
  __device__ static inline int lane_k(int lane) { return lane >> 4; }
  __device__ static inline int acc_i(int r, int lane) { return (lane >> 4) + 4 * r; }
  static constexpr int GS = 1;
  // [k][mn] image: row stride = 16 (mod 32) doubles, so the 16 lanes of k and
  // the 16 lanes of k + 1 in one ds_read_b64 lane group land on disjoint
  // halves of the 64 banks.
  static constexpr int PAD_MN = 16;
};

// v_mfma_f32_16x16x4_f32 (the small fp32 shape, for tiles that are multiples of 48): lane l feeds
// op0[i = l & 15][k = l >> 4] and op1[k = l >> 4][j = l & 15]; it receives D[i][j = l & 15] for
// i = 4 * (l >> 4) + r, r = 0..3.  Its four products are added in k order, each a fused multiply-add
// (tools/mfma16_order.hip: 256 of 256 elements of a K = 64 product equal the fmaf chain), so a kernel built on
// it produces the same bits as one built on 32x32x2.
struct MfmaF32Small {
  static constexpr int T = 16;
  static constexpr int TK = 4;
  static constexpr int NACC = 4;
  using acc_t = f32x4;
  using vec_t = f32x4;
  static constexpr int V = 4;
  __device__ static inline acc_t mma(float op0, float op1, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(op0, op1, c, 0, 0, 0);
  }
  __device__ static inline int lane_mn(int lane) { return lane & 15; }
  __device__ static inline int lane_k(int lane) { return lane >> 4; }
  __device__ static inline int acc_i(int r, int lane) { return 4 * (lane >> 4) + r; }
  static constexpr int GS = 1;
  // [k][mn] image: a row stride of 16 (mod 32) floats puts the lanes of k and k + 1 on the two halves of the banks
  static constexpr int PAD_MN = 16;
};

// ------------------------------------------------------------- tile loading
//
// An operand tile covers BMN indices along M (or N) and BK along K.
// KMAJOR = the K index is the contiguous one in global memory
//          (A with trans != 'n', B with trans == 'n').
// Each thread moves NV 16-byte vectors per tile: global -> registers -> LDS.
//
// LDS images (chosen so that neither the store nor the fragment read
// transposes or conflicts):
//   MN-major operand: [BK][BMN + PAD_MN]; a fragment is one 4/8-byte read per
//     lane and k-step, consecutive lanes on consecutive banks.
//   K-major operand:  [BMN][BK + 2], k fastest as in global memory.  The odd
//     multiple of 8 bytes in the row stride spreads the lanes of a fragment
//     read over all banks.  For fp32 every aligned group of four k is stored
//     as (k0, k2, k1, k3), so lane-half h reads the 8 bytes (k = 4j + h,
//     k = 4j + 2 + h): exactly its operands for MFMA steps 2j and 2j + 1.
template <typename T, int BMN, int BK, int NT, bool KMAJOR, typename MS = Mfma<T>>
struct TileLoader {
  using M = MS;
  static constexpr int V = M::V;
  static constexpr int S = KMAJOR ? (BK + 2) : (BMN + M::PAD_MN);  // LDS row stride (elements)
  static constexpr int ELEMS = KMAJOR ? BMN * S : BK * S;
  static constexpr int NVEC = BMN * BK / V;
  static_assert(NVEC % NT == 0, "tile must split evenly over the workgroup");
  static_assert(BK % 4 == 0, "K step is handled in groups of four");
  static constexpr int NV = NVEC / NT;
  static constexpr int PER_ROW = KMAJOR ? (BK / V) : (BMN / V);
  static constexpr int NWRITE = NV * ((KMAJOR && M::GS == 2) ? 2 : 1);  // LDS store instructions per tile

  T reg[NV][V];
  // Global memory takes a 16-byte load at element alignment; this type tells the compiler so.
  typedef typename M::vec_t vec_any_t __attribute__((aligned(sizeof(T))));

  // Interior tile, whole K step, 16-byte aligned: straight-line vector loads
  // through per-thread pointers that are set up once per workgroup and then
  // only advanced by a uniform stride (address arithmetic per load and K step
  // would otherwise steal issue slots from the matrix instructions).
  const T* ptr[NV];

  //
  // Tiles that hang over the matrix edge use the same path: a vector whose M/N
  // index lies past the edge is pointed at the last valid row (K-major) or, when it
  // ends past the edge, at the last valid vector (MN-major; needs mn_lim >= V).  What
  // it loads only feeds C elements that are never stored -- except the one vector that
  // straddles the edge when mn_lim % V != 0: store_ragged puts its elements in place.
  __device__ inline void init_fast(const T* __restrict__ x, int64_t ld, int64_t mn0, int64_t mn_lim, int64_t k0,
                                   int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + i * NT;
      const int row = v / PER_ROW;
      const int col = (v % PER_ROW) * V;
      if (KMAJOR) {
        int64_t mn = mn0 + row;
        mn = mn < mn_lim ? mn : mn_lim - 1;
        ptr[i] = x + (k0 + col) + mn * ld;
      } else {
        int64_t mn = mn0 + col;
        mn = mn + V <= mn_lim ? mn : mn_lim - V;
        ptr[i] = x + mn + (k0 + row) * ld;
      }
    }
  }

  // Load the tile the pointers address, then move them one K step on (DIR = -1: back).
  template <int DIR = 1>
  __device__ inline void load_fast(int64_t ld) {
    const int64_t step = DIR * (KMAJOR ? int64_t(BK) : int64_t(BK) * ld);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const typename M::vec_t vv = *reinterpret_cast<const vec_any_t*>(ptr[i]);
#pragma unroll
      for (int e = 0; e < V; ++e) reg[i][e] = vv[e];
      ptr[i] += step;
    }
  }

  // One vector of the tile: load_fast / store taken apart so that a K loop can spread them between its
  // matrix instructions (store vector i, then request its successor into the registers just freed).
  template <int DIR = 1>
  __device__ inline void load_one(int64_t ld, int i) {
    const typename M::vec_t vv = *reinterpret_cast<const vec_any_t*>(ptr[i]);
#pragma unroll
    for (int e = 0; e < V; ++e) reg[i][e] = vv[e];
    ptr[i] += DIR * (KMAJOR ? int64_t(BK) : int64_t(BK) * ld);
  }

  // Edge tile, K tail or unaligned operand: element loads.  Rows/columns past
  // the matrix edge re-read the last valid one (they only feed C elements that
  // are never stored); k past the end contributes exactly zero.
  __device__ inline void load_edge(const T* __restrict__ x, int64_t ld, int64_t mn0, int64_t mn_lim,
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
        const int64_t ks = live ? kk : (k_end - 1);  // k_end > k0 >= 0 here
        const T val = KMAJOR ? x[ks + mn * ld] : x[mn + ks * ld];
        reg[i][e] = live ? val : T(0);
      }
    }
  }

  __device__ inline void store(SBAMD_LDS T* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) store_one(lds, tid, i);
  }

  // store() for a tile whose M/N extent ends inside a vector (tile_valid = mn_lim - mn0, not a multiple
  // of V): the vector that straddles the edge was fetched from the last valid vector, V - (its valid
  // elements) positions early, and its elements go to their own columns one by one (the leading ones land
  // on what the neighbouring vector stores there too -- the same values).
  __device__ inline void store_ragged(SBAMD_LDS T* lds, int tid, int tile_valid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = tid + i * NT;
      const int row = v / PER_ROW;
      const int col = (v % PER_ROW) * V;
      if (!KMAJOR && col < tile_valid && col + V > tile_valid) {
        const int first = tile_valid - V;  // column of reg[i][0]; negative in a tile narrower than a vector
#pragma unroll
        for (int e = 0; e < V; ++e)
          if (first + e >= 0) lds[row * S + first + e] = reg[i][e];
      } else {
        store_one(lds, tid, i);
      }
    }
  }

  __device__ inline void store_one(SBAMD_LDS T* lds, int tid, int i) const {
    {
      const int v = tid + i * NT;
      const int row = v / PER_ROW;
      const int col = (v % PER_ROW) * V;
      SBAMD_LDS T* p = lds + row * S + col;
      if (KMAJOR && M::GS == 2) {
        // (k0, k2, k1, k3), written element by element.  (Built from 8-byte vector stores, the register
        // pairs (k0, k2) / (k1, k3) have to be assembled first, and hipcc assembles them right behind the
        // global loads at the bottom of the K loop -- waiting there for loads it has just issued, every
        // step.)  volatile keeps the vectoriser from re-forming those pairs; the stores stay visible to
        // the compiler's wait counting.
        volatile SBAMD_LDS T* q = p;
        q[0] = reg[i][0];
        q[1] = reg[i][2];
        q[2] = reg[i][1];
        q[3] = reg[i][3];
      } else {
        typename M::vec_t vv;
#pragma unroll
        for (int e = 0; e < V; ++e) vv[e] = reg[i][e];
        *reinterpret_cast<SBAMD_LDS typename M::vec_t*>(p) = vv;
      }
    }
  }

  // Fragments of one lane for the GS consecutive MFMA k-steps of group `grp`
  // (GS = 2 for fp32, 1 for fp64): out[e] belongs to step grp*GS + e.
  // `mn` is the lane's M/N index inside the tile.
  __device__ static inline void fragment(const SBAMD_LDS T* lds, int grp, int mn, int lane,
                                         T (&out)[M::GS]) {
    const int lk = M::lane_k(lane);
    if (KMAJOR) {
      if (M::GS == 2) {
        typedef T pair_t __attribute__((ext_vector_type(2)));
        const pair_t pv = *reinterpret_cast<const SBAMD_LDS pair_t*>(lds + mn * S + 4 * grp + 2 * lk);
        out[0] = pv[0];
        out[M::GS - 1] = pv[1];
      } else {
        out[0] = lds[mn * S + M::TK * grp + lk];
      }
    } else {
#pragma unroll
      for (int e = 0; e < M::GS; ++e) out[e] = lds[((grp * M::GS + e) * M::TK + lk) * S + mn];
    }
  }
};

// XCD-aware, bijective remap of the linear workgroup id: workgroups are dealt
// round-robin over the 8 XCDs, so give each XCD one contiguous run of logical
// ids (neighbouring tiles then share an L2).
__device__ inline int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7;
  const int q = nwg >> 3, r = nwg & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// Rows of tiles are walked in bands of GROUP_M so that a run of consecutive
// logical ids touches few distinct A and B panels.
// The 32 workgroups resident on one XCD (one per compute unit) then form a GROUP_M x 32/GROUP_M patch
// of C and share (GROUP_M * BM + 32/GROUP_M * BN) of the 32 * (BM + BN) operand rows they request per
// K step from that XCD's L2: 8 x 4 for square tiles, 4 x 8 for the 256 x 128 tile of double.
constexpr int GROUP_M = 8;
__device__ inline void tile_coords(int t, int tiles_m, int tiles_n, int& tm, int& tn, int group_m = GROUP_M) {
  if (group_m <= 0) group_m = GROUP_M;
  const int band = group_m * tiles_n;
  const int g = t / band;
  const int first_m = g * group_m;
  const int gsz = (tiles_m - first_m) < group_m ? (tiles_m - first_m) : group_m;
  const int in = t - g * band;
  tm = first_m + in % gsz;
  tn = in / gsz;
}

// ------------------------------------------------------------------ the kernel
template <typename T, int BM, int BN, int BK, int WM, int WN, bool TA, bool TB, typename MS = Mfma<T>>
struct GemmTile {
  using M = MS;
  static constexpr int NT = WM * WN * 64;
  static constexpr int WTM = BM / WM;  // wavefront tile
  static constexpr int WTN = BN / WN;
  static constexpr int MI = WTM / M::T;
  static constexpr int NI = WTN / M::T;
  static_assert(MI >= 1 && NI >= 1, "wave tile smaller than one MFMA");
  static constexpr int NGRP = BK / (M::TK * M::GS);
  static_assert(BK % (M::TK * M::GS) == 0, "BK must hold whole fragment groups");
  // A is K-major when transposed; B is K-major when NOT transposed.
  using LoadA = TileLoader<T, BM, BK, NT, TA, MS>;
  using LoadB = TileLoader<T, BN, BK, NT, !TB, MS>;
  static constexpr int A_ELEMS = (LoadA::ELEMS + 3) & ~3;  // keep B 16-byte aligned
  static constexpr int B_ELEMS = (LoadB::ELEMS + 3) & ~3;
  static constexpr int STAGE_ELEMS = A_ELEMS + B_ELEMS;
  static constexpr size_t LDS_BYTES = 2 * size_t(STAGE_ELEMS) * sizeof(T);
};

template <typename T, int BM, int BN, int BK, int WM, int WN, bool TA, bool TB, bool FENCE_LOADS = false, typename MS = Mfma<T>>
struct GemmBody {
  using K = GemmTile<T, BM, BN, BK, WM, WN, TA, TB, MS>;
  using M = MS;
  using LoadA = typename K::LoadA;
  using LoadB = typename K::LoadB;

  typename M::acc_t acc[K::MI][K::NI];
  LoadA la;
  LoadB lb;

  // All MFMAs of one K step.  The fragments of group g + 1 are fetched while
  // the matrix instructions of group g run: the source is software-pipelined
  // and the sched_group_barrier sequence pins the interleave (one LDS read
  // slot behind each MFMA), because the scheduler otherwise sinks every read
  // next to its first use and exposes the LDS latency once per group.
  template <int NVMEM>
  __device__ inline void compute(const SBAMD_LDS T* as, const SBAMD_LDS T* bs, int a_mn,
                                 int b_mn, int lane) {
    constexpr int NM = K::MI * K::NI * M::GS;  // MFMAs per group
    T fa[2][K::MI][M::GS], fb[2][K::NI][M::GS];
#pragma unroll
    for (int i = 0; i < K::MI; ++i) LoadA::fragment(as, 0, a_mn + i * M::T, lane, fa[0][i]);
#pragma unroll
    for (int j = 0; j < K::NI; ++j) LoadB::fragment(bs, 0, b_mn + j * M::T, lane, fb[0][j]);
#pragma unroll
    for (int grp = 0; grp < K::NGRP; ++grp) {
      const int cur = grp & 1, nxt = cur ^ 1;
      if (grp + 1 < K::NGRP) {
#pragma unroll
        for (int i = 0; i < K::MI; ++i) LoadA::fragment(as, grp + 1, a_mn + i * M::T, lane, fa[nxt][i]);
#pragma unroll
        for (int j = 0; j < K::NI; ++j) LoadB::fragment(bs, grp + 1, b_mn + j * M::T, lane, fb[nxt][j]);
      }
#pragma unroll
      for (int e = 0; e < M::GS; ++e)
#pragma unroll
        for (int i = 0; i < K::MI; ++i)
#pragma unroll
          for (int j = 0; j < K::NI; ++j)
            // op0 carries the N index and op1 the M index, so a lane ends up
            // holding one M index (contiguous in column-major C).
            acc[i][j] = M::mma(fb[cur][j][e], fa[cur][i][e], acc[i][j]);
    }
    // Desired order (0x008 = MFMA, 0x020 = VMEM read, 0x100 = DS read): the
    // first group's fragments; then behind each MFMA one of the NVMEM global
    // loads of the next tile (issued early so they land before the LDS store
    // at the end of the step), and behind every STRIDE-th MFMA one fragment
    // read, which keeps the reads about one group ahead of their use.
    constexpr int SLOTS = K::MI + K::NI;  // fragment reads per group after pairing
    constexpr int STRIDE = NM / SLOTS > 0 ? NM / SLOTS : 1;
    constexpr int PER = SLOTS / NM > 0 ? SLOTS / NM : 1;
    __builtin_amdgcn_sched_group_barrier(0x100, SLOTS, 0);
#pragma unroll
    for (int n = 0; n < K::NGRP * NM; ++n) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (n < NVMEM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      if (n % STRIDE == 0) __builtin_amdgcn_sched_group_barrier(0x100, PER, 0);
    }
  }

  // ---- pipelined path (interior tiles, aligned operands) -------------------
  //
  // Iteration kt computes tile kt out of LDS buffer kt & 1 while
  //   * the registers loaded one whole iteration ago (tile kt + 1) are stored
  //     to the other buffer at the top of the iteration, and
  //   * the global loads of tile kt + 2 are issued right behind that store,
  // so every global load has a full K step of matrix work to land.  The one
  // barrier per step sits in front of the LAST fragment group's MFMAs: by then
  // every fragment of tile kt is in registers (the buffer may be overwritten
  // next iteration) and every wave's store of tile kt + 1 has completed, so the
  // first fragments of tile kt + 1 are fetched behind the barrier and their
  // latency hides under the last group's matrix instructions.
  using frag_a_t = T[2][K::MI][M::GS];
  using frag_b_t = T[2][K::NI][M::GS];

  __device__ inline void fetch_group(const SBAMD_LDS T* as, const SBAMD_LDS T* bs, int grp, int a_mn,
                                     int b_mn, int lane, T (&fa)[K::MI][M::GS], T (&fb)[K::NI][M::GS]) {
#pragma unroll
    for (int i = 0; i < K::MI; ++i) LoadA::fragment(as, grp, a_mn + i * M::T, lane, fa[i]);
#pragma unroll
    for (int j = 0; j < K::NI; ++j) LoadB::fragment(bs, grp, b_mn + j * M::T, lane, fb[j]);
  }

  __device__ inline void mma_group(const T (&fa)[K::MI][M::GS], const T (&fb)[K::NI][M::GS]) {
#pragma unroll
    for (int e = 0; e < M::GS; ++e)
#pragma unroll
      for (int i = 0; i < K::MI; ++i)
#pragma unroll
        for (int j = 0; j < K::NI; ++j) acc[i][j] = M::mma(fb[j][e], fa[i][e], acc[i][j]);
  }

  // One K step of the pipelined path.  Group 0 of this tile is already in
  // fa[0] / fb[0]; on return group 0 of the next tile is (HAS_NEXT).
  // NWRITE / NVMEM: LDS stores and global loads issued by the caller at the top
  // of the step, placed behind the first MFMAs by the scheduling pattern.
  template <bool HAS_NEXT, int NWRITE, int NVMEM>
  __device__ inline void pipelined_step(const SBAMD_LDS T* as, const SBAMD_LDS T* bs,
                                        const SBAMD_LDS T* nas, const SBAMD_LDS T* nbs, int a_mn,
                                        int b_mn, int lane, frag_a_t& fa, frag_b_t& fb) {
    static_assert(K::NGRP % 2 == 0, "fragment double buffer must come back to slot 0");
    constexpr int NM = K::MI * K::NI * M::GS;
#pragma unroll
    for (int grp = 0; grp + 1 < K::NGRP; ++grp) {
      if (!(SBAMD_ABLATE & 8)) fetch_group(as, bs, grp + 1, a_mn, b_mn, lane, fa[(grp + 1) & 1], fb[(grp + 1) & 1]);
      mma_group(fa[grp & 1], fb[grp & 1]);
    }
    // 0x008 MFMA, 0x020 VMEM read, 0x100 DS read, 0x200 DS write
    constexpr int SLOTS = K::MI + K::NI;
    constexpr int STRIDE = NM / SLOTS > 0 ? NM / SLOTS : 1;
    constexpr int PER = SLOTS / NM > 0 ? SLOTS / NM : 1;
    // LDS store instructions the pattern claims: the caller's count, except that a K-contiguous fp32 operand is
    // written with four 4-byte stores per vector, not two 8-byte ones (claiming all of them: 8192^3 NN on the
    // 256 x 256 tile 145.4 -> 146.5-146.8 TFLOP/s; the 128 x 128 tile on eight wavefronts measured no gain and keeps
    // the caller's count).
    constexpr int NWR = (NWRITE > 0 && K::MI * K::NI >= 4)
                            ? LoadA::NV * ((TA && M::GS == 2) ? 4 : 1) + LoadB::NV * ((!TB && M::GS == 2) ? 4 : 1)
                            : NWRITE;
#pragma unroll
    for (int n = 0; n < (K::NGRP - 1) * NM; ++n) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (n < NWR) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      else if (n < NWR + NVMEM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      if (n % STRIDE == 0) __builtin_amdgcn_sched_group_barrier(0x100, PER, 0);
    }
    // The pattern above only orders what it claims.  Left alone, hipcc moved a third of the matrix instructions
    // of the last claimed group across the barrier and sank the next tile's first fragment reads to the bottom
    // of the loop, right in front of the wait at its top.  Fences keep the barrier where it is written, and
    // a second pattern puts one of those reads behind each of the first matrix instructions after it
    // (8192^3: fp32 NT 143.3 -> 144.6-145.6 TFLOP/s, fp64 NN 73.3 -> 73.8-74.1, fp32 NN unchanged; fp64 64 x 64
    // tile at 1024^3 51.7 -> 54.0).  Wavefront tiles of two matrix tiles (fp32 128 x 128 on eight wavefronts)
    // stay as they were: NN 2048^3 lost 1.6 % with it.
    constexpr bool PIN_TAIL = SBAMD_LAB_VARIANT >= 1 && K::MI * K::NI >= 4;
    if (PIN_TAIL) __builtin_amdgcn_sched_barrier(0);
    if (!(SBAMD_ABLATE & 4)) __syncthreads();
    if (PIN_TAIL) __builtin_amdgcn_sched_barrier(0);
    if (HAS_NEXT && !(SBAMD_ABLATE & 8)) fetch_group(nas, nbs, 0, a_mn, b_mn, lane, fa[0], fb[0]);
    mma_group(fa[(K::NGRP - 1) & 1], fb[(K::NGRP - 1) & 1]);
    if (PIN_TAIL && HAS_NEXT) {
#pragma unroll
      for (int n = 0; n < NM; ++n) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
        if (n < SLOTS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
      }
    }
  }

  // ---- the pipelined path for one matrix tile per wavefront (64 x 64 workgroup tile, 32 x 32 tile) --------
  //
  // A fragment group is then only two matrix instructions (128 cycles) long and a wavefront has no second
  // accumulator to turn to: a fragment that is read one group ahead is waited for about one matrix instruction
  // after its read was issued, less than an LDS read takes.  Here the fragments run TWO groups ahead through
  // four register slots (group g in slot g & 3; NGRP is a multiple of four, so a K step hands over in slots 0
  // and 1 again), and the order is fixed with scheduling fences instead of a pattern: reads of group g + 2,
  // then the matrix instructions of group g.  The barrier sits in front of the last two groups' matrix
  // instructions; behind it the first two groups of the next tile are fetched.
  using far_a_t = T[4][K::MI][M::GS];
  using far_b_t = T[4][K::NI][M::GS];

  // MOVE: this step also moves operand data -- vector by vector, one piece behind the fragment reads of each
  // of the first groups: store vector i of tile kt + 1 to LDS (STORE), then request vector i of tile kt + 2
  // into the registers just freed (LOAD).  Bunched at the top of the step instead, the four wavefronts issue
  // ten LDS stores and four 1-KiB loads each at the same moment and wait on the queues in front of the LDS
  // and the texture path before their first matrix instruction.
  template <bool HAS_NEXT, bool STORE, bool LOAD>
  __device__ inline void far_step(const GemmArgs<T>& g, const SBAMD_LDS T* as, const SBAMD_LDS T* bs, SBAMD_LDS T* nas,
                                  SBAMD_LDS T* nbs, int a_mn, int b_mn, int lane, int tid, far_a_t& fa, far_b_t& fb) {
    static_assert(K::NGRP % 4 == 0, "the fragment slots must come back to 0 / 1 at the end of a K step");
    constexpr int PIECES = LoadA::NV + LoadB::NV;
    constexpr int PER = (PIECES + K::NGRP - 3) / (K::NGRP - 2);  // pieces per group, all in front of the barrier
#pragma unroll
    for (int grp = 0; grp < K::NGRP; ++grp) {
      if (grp + 2 == K::NGRP) __syncthreads();
      if (grp + 2 < K::NGRP) fetch_group(as, bs, grp + 2, a_mn, b_mn, lane, fa[(grp + 2) & 3], fb[(grp + 2) & 3]);
      else if (HAS_NEXT) fetch_group(nas, nbs, grp + 2 - K::NGRP, a_mn, b_mn, lane, fa[(grp + 2) & 3], fb[(grp + 2) & 3]);
      if (STORE) {
#pragma unroll
        for (int q = grp * PER; q < (grp + 1) * PER && q < PIECES; ++q) {
          if (q < LoadA::NV) {
            la.store_one(nas, tid, q);
            if (LOAD) la.load_one(g.lda, q);
          } else {
            lb.store_one(nbs, tid, q - LoadA::NV);
            if (LOAD) lb.load_one(g.ldb, q - LoadA::NV);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      mma_group(fa[grp & 3], fb[grp & 3]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  __device__ inline void run_pipelined_far(const GemmArgs<T>& g, const T* a, const T* b, int64_t m0, int64_t n0,
                                           int64_t k_begin, int64_t k_end, SBAMD_LDS T* smem, int tid, int a_mn,
                                           int b_mn) {
    const int lane = tid & 63;
    const int nfull = int((k_end - k_begin) / BK);  // >= 1 (checked by the caller)
    SBAMD_LDS T* const buf0 = smem;
    SBAMD_LDS T* const buf1 = smem + K::STAGE_ELEMS;
    la.init_fast(a, g.lda, m0, g.m, k_begin, tid);
    lb.init_fast(b, g.ldb, n0, g.n, k_begin, tid);
    la.load_fast(g.lda);
    lb.load_fast(g.ldb);
    la.store(buf0, tid);
    lb.store(buf0 + K::A_ELEMS, tid);
    if (nfull > 1) {
      la.load_fast(g.lda);
      lb.load_fast(g.ldb);
    }
    __syncthreads();
    far_a_t fa;
    far_b_t fb;
    fetch_group(buf0, buf0 + K::A_ELEMS, 0, a_mn, b_mn, lane, fa[0], fb[0]);
    fetch_group(buf0, buf0 + K::A_ELEMS, 1, a_mn, b_mn, lane, fa[1], fb[1]);

    int kt = 0;
    if constexpr (KLOOP_ASM && KL::GA == 2) kt = run_kloop_asm(g, a, b, m0, n0, k_begin, nfull, smem, tid, a_mn, b_mn, fa, fb);
    for (; kt + 2 < nfull; ++kt) {
      SBAMD_LDS T* cur = (kt & 1) ? buf1 : buf0;
      SBAMD_LDS T* nxt = (kt & 1) ? buf0 : buf1;
      // tile kt + 1 (loaded one step ago) goes to LDS, tile kt + 2 is requested into the freed registers
      far_step<true, true, true>(g, cur, cur + K::A_ELEMS, nxt, nxt + K::A_ELEMS, a_mn, b_mn, lane, tid, fa, fb);
    }
    if (kt + 1 < nfull) {  // last but one: nothing left to request
      SBAMD_LDS T* cur = (kt & 1) ? buf1 : buf0;
      SBAMD_LDS T* nxt = (kt & 1) ? buf0 : buf1;
      far_step<true, true, false>(g, cur, cur + K::A_ELEMS, nxt, nxt + K::A_ELEMS, a_mn, b_mn, lane, tid, fa, fb);
      ++kt;
    }
    {
      SBAMD_LDS T* cur = (kt & 1) ? buf1 : buf0;
      far_step<false, false, false>(g, cur, cur + K::A_ELEMS, cur, cur + K::A_ELEMS, a_mn, b_mn, lane, tid, fa, fb);
    }
    const int64_t k_tail = k_begin + int64_t(nfull) * BK;
    if (k_tail < k_end) {
      la.load_edge(a, g.lda, m0, g.m, k_tail, k_end, tid);
      lb.load_edge(b, g.ldb, n0, g.n, k_tail, k_end, tid);
      la.store(buf0, tid);
      lb.store(buf0 + K::A_ELEMS, tid);
      __syncthreads();
      compute<0>(buf0, buf0 + K::A_ELEMS, a_mn, b_mn, lane);
    }
  }

  // The steady-state K loop in assembly (gemm_kloop_asm.hpp) for the tile shapes that have a block there.  Called
  // with the state the compiled loop has at kt = 0 (tile 0 in buffer 0, tile 1 on its way into the staging
  // registers, the pointers at tile 2, the first GA fragment groups of tile 0 in fa[0..GA) / fb[0..GA)); runs an
  // even number of K steps, leaves the same state for kt = that number and returns it (0: not taken).  The
  // block addresses global memory as a scalar base (the tile's origin) plus a 32-bit per-lane offset: taken
  // whenever the leading dimensions keep those offsets below 2^32.
  using KL = KLoop<T, BM, BN, BK, WM, WN, TA, TB>;
  static constexpr bool KLOOP_ASM = SBAMD_KLOOP_ASM && !SBAMD_ABLATE && KL::available && KL::MI == K::MI && KL::NI == K::NI &&
                                    (KL::SMALL ? std::is_same<MS, MfmaF32Small>::value : std::is_same<MS, Mfma<T>>::value);
  template <int NS>
  __device__ inline int run_kloop_asm(const GemmArgs<T>& g, const T* a, const T* b, int64_t m0, int64_t n0, int64_t k_begin,
                                      int nfull, SBAMD_LDS T* smem, int tid, int a_mn, int b_mn, T (&fa)[NS][K::MI][M::GS],
                                      T (&fb)[NS][K::NI][M::GS]) {
    if constexpr (KLOOP_ASM) {
      constexpr bool A_KM = TA, B_KM = !TB;
      constexpr int ES = int(sizeof(T)), V = M::V, GS = M::GS;
      static_assert(KL::NVA == LoadA::NV && KL::NVB == LoadB::NV && KL::GA <= NS, "the block was generated for this tile");
      const int iters = (nfull - 2) / 2;
      // largest per-lane byte offset inside a tile: (rows - 1) * ld + the last vector of a row
      constexpr int64_t ROWS_A = int64_t(LoadA::NVEC) / LoadA::PER_ROW, ROWS_B = int64_t(LoadB::NVEC) / LoadB::PER_ROW;
      const bool offsets_fit = ((ROWS_A - 1) * g.lda + LoadA::PER_ROW * V) * ES < (int64_t(1) << 32) &&
                               ((ROWS_B - 1) * g.ldb + LoadB::PER_ROW * V) * ES < (int64_t(1) << 32);
      if (iters < 1 || !offsets_fit) return 0;
      const int lk = M::lane_k(tid & 63);
      const uint32_t lds0 = uint32_t(uintptr_t(smem));
      const int64_t k2 = k_begin + 2 * BK;  // the pointers stand at tile 2
      // Per-lane offsets from the tile's origin to where the lane's pointers stand: for a tile that hangs over the
      // edge those pointers were clamped to the last valid row / vector by init_fast, and the offsets carry the
      // clamping (they stay non-negative: the tile's origin is its smallest address).
      const T* origin_a = A_KM ? a + k2 + m0 * g.lda : a + m0 + k2 * g.lda;
      const T* origin_b = B_KM ? b + k2 + n0 * g.ldb : b + n0 + k2 * g.ldb;
      uint32_t goa[KL::NVA], gob[KL::NVB];
#pragma unroll
      for (int i = 0; i < KL::NVA; ++i) goa[i] = uint32_t((la.ptr[i] - origin_a) * ES);
#pragma unroll
      for (int i = 0; i < KL::NVB; ++i) gob[i] = uint32_t((lb.ptr[i] - origin_b) * ES);
      // wave-uniform by construction; say so to the compiler (the block takes them in scalar registers)
      auto uniform = [](uint64_t v) {
        const uint32_t lo = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v))));
        const uint32_t hi = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(v >> 32))));
        return (uint64_t(hi) << 32) | uint64_t(lo);
      };
      uint64_t base_a = uniform(uint64_t(uintptr_t(origin_a)));
      uint64_t base_b = uniform(uint64_t(uintptr_t(origin_b)));
      const uint64_t step_a = uniform(A_KM ? uint64_t(BK * ES) : uint64_t(g.lda) * (BK * ES));
      const uint64_t step_b = uniform(B_KM ? uint64_t(BK * ES) : uint64_t(g.ldb) * (BK * ES));
      const int lk_a = A_KM && GS == 2 ? 2 * lk : lk, lk_b = B_KM && GS == 2 ? 2 * lk : lk;
      const uint32_t rda = lds0 + uint32_t(A_KM ? (a_mn * LoadA::S + lk_a) * ES : (lk * LoadA::S + a_mn) * ES);
      const uint32_t rdb = lds0 + uint32_t(K::A_ELEMS * ES) + uint32_t(B_KM ? (b_mn * LoadB::S + lk_b) * ES : (lk * LoadB::S + b_mn) * ES);
      // every per-lane input in one register tuple (the block names the registers)
      typename KL::lane_t lanes;
#pragma unroll
      for (int i = 0; i < KL::LANE_N; ++i) lanes[i] = 0;
#pragma unroll
      for (int i = 0; i < KL::NVA; ++i) lanes[KL::L_GOA + i] = goa[i];
#pragma unroll
      for (int i = 0; i < KL::NVB; ++i) lanes[KL::L_GOB + i] = gob[i];
      lanes[KL::L_RDA] = rda;
      lanes[KL::L_RDB] = rdb;
      // LDS address of the lane's vector i of a tile (TileLoader::store_one); one per operand where the others
      // follow by a constant, one per vector where a row's vectors do not divide the workgroup
#pragma unroll
      for (int i = 0; i < KL::NWA; ++i) {
        const int v = tid + i * K::NT;
        lanes[KL::L_WRA + i] = lds0 + uint32_t(((v / LoadA::PER_ROW) * LoadA::S + (v % LoadA::PER_ROW) * V) * ES);
      }
#pragma unroll
      for (int i = 0; i < KL::NWB; ++i) {
        const int v = tid + i * K::NT;
        lanes[KL::L_WRB + i] = lds0 + uint32_t(K::A_ELEMS * ES) + uint32_t(((v / LoadB::PER_ROW) * LoadB::S + (v % LoadB::PER_ROW) * V) * ES);
      }
      typename KL::acc_t accv[KL::ACC_CHUNKS];
      typename KL::sta_t sta;
      typename KL::stb_t stb;
      typename KL::fa_t fav[KL::GA];
      typename KL::fb_t fbv[KL::GA];
#pragma unroll
      for (int i = 0; i < K::MI; ++i)
#pragma unroll
        for (int j = 0; j < K::NI; ++j)
#pragma unroll
          for (int r = 0; r < M::NACC; ++r) {
            const int L = (i * K::NI + j) * M::NACC + r;
            accv[L / KL::CHUNK][L % KL::CHUNK] = acc[i][j][r];
          }
#pragma unroll
      for (int i = 0; i < KL::NVA; ++i)
#pragma unroll
        for (int e = 0; e < V; ++e) sta[i * V + e] = la.reg[i][e];
#pragma unroll
      for (int i = 0; i < KL::NVB; ++i)
#pragma unroll
        for (int e = 0; e < V; ++e) stb[i * V + e] = lb.reg[i][e];
#pragma unroll
      for (int s = 0; s < KL::GA; ++s) {
#pragma unroll
        for (int i = 0; i < K::MI; ++i)
#pragma unroll
          for (int e = 0; e < GS; ++e) fav[s][i * GS + e] = fa[s][i][e];
#pragma unroll
        for (int j = 0; j < K::NI; ++j)
#pragma unroll
          for (int e = 0; e < GS; ++e) fbv[s][j * GS + e] = fb[s][j][e];
      }
      KL::run(accv, sta, stb, fav, fbv, lanes, base_a, base_b, step_a, step_b, __builtin_amdgcn_readfirstlane(iters));
#pragma unroll
      for (int i = 0; i < K::MI; ++i)
#pragma unroll
        for (int j = 0; j < K::NI; ++j)
#pragma unroll
          for (int r = 0; r < M::NACC; ++r) {
            const int L = (i * K::NI + j) * M::NACC + r;
            acc[i][j][r] = accv[L / KL::CHUNK][L % KL::CHUNK];
          }
#pragma unroll
      for (int i = 0; i < KL::NVA; ++i)
#pragma unroll
        for (int e = 0; e < V; ++e) la.reg[i][e] = sta[i * V + e];
#pragma unroll
      for (int i = 0; i < KL::NVB; ++i)
#pragma unroll
        for (int e = 0; e < V; ++e) lb.reg[i][e] = stb[i * V + e];
#pragma unroll
      for (int s = 0; s < KL::GA; ++s) {
#pragma unroll
        for (int i = 0; i < K::MI; ++i)
#pragma unroll
          for (int e = 0; e < GS; ++e) fa[s][i][e] = fav[s][i * GS + e];
#pragma unroll
        for (int j = 0; j < K::NI; ++j)
#pragma unroll
          for (int e = 0; e < GS; ++e) fb[s][j][e] = fbv[s][j * GS + e];
      }
      const int64_t adv_a = int64_t(2 * iters) * (A_KM ? int64_t(BK) : int64_t(BK) * g.lda);
      const int64_t adv_b = int64_t(2 * iters) * (B_KM ? int64_t(BK) : int64_t(BK) * g.ldb);
#pragma unroll
      for (int i = 0; i < KL::NVA; ++i) la.ptr[i] += adv_a;
#pragma unroll
      for (int i = 0; i < KL::NVB; ++i) lb.ptr[i] += adv_b;
      return 2 * iters;
    }
    return 0;
  }

  __device__ inline void run_pipelined(const GemmArgs<T>& g, const T* a, const T* b, int64_t m0,
                                       int64_t n0, int64_t k_begin, int64_t k_end, SBAMD_LDS T* smem,
                                       int tid, int a_mn, int b_mn) {
    const int lane = tid & 63;
    const int64_t klen = k_end - k_begin;
    const int nfull = int(klen / BK);  // >= 1 (checked by the caller)
    if constexpr (K::MI * K::NI <= SBAMD_FAR_MAX_TILES && K::NGRP % 4 == 0 && !SBAMD_ABLATE) {
      run_pipelined_far(g, a, b, m0, n0, k_begin, k_end, smem, tid, a_mn, b_mn);
      return;
    }
    constexpr int NW = LoadA::NWRITE + LoadB::NWRITE;
    constexpr int NL = LoadA::NV + LoadB::NV;
    SBAMD_LDS T* const buf0 = smem;
    SBAMD_LDS T* const buf1 = smem + K::STAGE_ELEMS;

    la.init_fast(a, g.lda, m0, g.m, k_begin, tid);
    lb.init_fast(b, g.ldb, n0, g.n, k_begin, tid);
    la.load_fast(g.lda);
    lb.load_fast(g.ldb);
    la.store(buf0, tid);
    lb.store(buf0 + K::A_ELEMS, tid);
    if (nfull > 1) {
      la.load_fast(g.lda);
      lb.load_fast(g.ldb);
    }
    __syncthreads();
    frag_a_t fa;
    frag_b_t fb;
    fetch_group(buf0, buf0 + K::A_ELEMS, 0, a_mn, b_mn, lane, fa[0], fb[0]);

    int kt = 0;
    if constexpr (KLOOP_ASM && KL::GA == 1) kt = run_kloop_asm(g, a, b, m0, n0, k_begin, nfull, smem, tid, a_mn, b_mn, fa, fb);
    for (; kt + 2 < nfull; ++kt) {
      SBAMD_LDS T* cur = (kt & 1) ? buf1 : buf0;
      SBAMD_LDS T* nxt = (kt & 1) ? buf0 : buf1;
      if (!(SBAMD_ABLATE & 1)) {
        la.store(nxt, tid);
        lb.store(nxt + K::A_ELEMS, tid);
      }
      if (!(SBAMD_ABLATE & 2)) {
        la.load_fast(g.lda);
        lb.load_fast(g.ldb);
      }
      // One matrix tile per wavefront (the 64 x 64 workgroup tile, one wavefront per SIMD): nothing else
      // hides a load that is waited for early, and left to itself the scheduler sinks these loads to the
      // bottom of the step, a few hundred cycles in front of the LDS store that needs them.  Keep them
      // here, a whole K step ahead of that store.
      // Wavefront tiles of two matrix tiles (fp32 128 x 128 on eight wavefronts) with a K-contiguous operand:
      // the same sinking (the loads ended up behind the last matrix instructions, a vmcnt(0) at the top of the
      // next step waiting for them); fenced, NN 2048^3 130 -> 134 TFLOP/s, 4096^3 141.8 -> 143.9, TN 139.0 -> 141.6.
      // With both operands M/N-contiguous the pattern places them behind the stores by itself (fenced: -1 %).
      // (Also the 256 x 128 x 32 tile that runs the K slabs of fp32 TN shapes: 8192^3 136.9 -> 140.1 TFLOP/s.  The
      // 256 x 256 tiles lose 5 % when fenced: there the pattern keeps the loads behind the stores by itself.)
      constexpr bool SLAB_TN = sizeof(T) == 4 && BM == 256 && BN == 128 && BK == 32 && TA && !TB;
      if constexpr (K::MI * K::NI == 1 || (K::MI * K::NI == 2 && (TA || !TB)) || SLAB_TN || FENCE_LOADS) __builtin_amdgcn_sched_barrier(0);
      pipelined_step<true, NW, NL>(cur, cur + K::A_ELEMS, nxt, nxt + K::A_ELEMS, a_mn, b_mn, lane, fa, fb);
    }
    if (kt + 1 < nfull) {  // last but one: nothing left to load
      SBAMD_LDS T* cur = (kt & 1) ? buf1 : buf0;
      SBAMD_LDS T* nxt = (kt & 1) ? buf0 : buf1;
      la.store(nxt, tid);
      lb.store(nxt + K::A_ELEMS, tid);
      pipelined_step<true, NW, 0>(cur, cur + K::A_ELEMS, nxt, nxt + K::A_ELEMS, a_mn, b_mn, lane, fa, fb);
      ++kt;
    }
    {
      SBAMD_LDS T* cur = (kt & 1) ? buf1 : buf0;
      pipelined_step<false, 0, 0>(cur, cur + K::A_ELEMS, cur, cur + K::A_ELEMS, a_mn, b_mn, lane, fa, fb);
    }
    // K tail (< BK): one unpipelined step.  The barrier inside the last step
    // came after every fragment read, so both buffers are free.
    const int64_t k_tail = k_begin + int64_t(nfull) * BK;
    if (k_tail < k_end) {
      la.load_edge(a, g.lda, m0, g.m, k_tail, k_end, tid);
      lb.load_edge(b, g.ldb, n0, g.n, k_tail, k_end, tid);
      la.store(buf0, tid);
      lb.store(buf0 + K::A_ELEMS, tid);
      __syncthreads();
      compute<0>(buf0, buf0 + K::A_ELEMS, a_mn, b_mn, lane);
    }
  }

  // ---- chunked accumulation: every CHUNK of k is its own product, folded into the
  // running tile before the next one starts:
  //     run <- -(A_chunk * B_chunk) + run        (product and sum rounded apart)
  // which is, element for element and rounding for rounding, what a sequence of
  // k / CHUNK separate GEMM calls with alpha = -1, beta = 1 on the same C computes -- in one pass
  // over C instead of k / CHUNK passes.  A K step of the pipeline holds BK / CHUNK
  // chunks.  REVERSE walks the chunks from the high end of K down (k inside a chunk
  // still ascends).  Needs k % BK == 0, k >= BK and operands the vector path can load.
  //
  // One K step: the pipelined step above with the fragment groups visited chunk by
  // chunk (REVERSE: last chunk of the tile first) and a fold behind each chunk.
  template <bool HAS_NEXT, int NWRITE, int NVMEM, bool REVERSE, int GPC, typename Fold>
  __device__ inline void chunked_step(const SBAMD_LDS T* as, const SBAMD_LDS T* bs, const SBAMD_LDS T* nas,
                                      const SBAMD_LDS T* nbs, int a_mn, int b_mn, int lane, frag_a_t& fa,
                                      frag_b_t& fb, Fold&& fold) {
    static_assert(K::NGRP % 2 == 0 && K::NGRP % GPC == 0, "whole chunks, fragment double buffer back at slot 0");
    constexpr int NM = K::MI * K::NI * M::GS;
#pragma unroll
    for (int i = 0; i + 1 < K::NGRP; ++i) {
      fetch_group(as, bs, chunk_order<REVERSE, GPC>(i + 1), a_mn, b_mn, lane, fa[(i + 1) & 1], fb[(i + 1) & 1]);
      mma_group(fa[i & 1], fb[i & 1]);
      if ((i + 1) % GPC == 0) fold();
    }
    constexpr int SLOTS = K::MI + K::NI;
    constexpr int STRIDE = NM / SLOTS > 0 ? NM / SLOTS : 1;
    constexpr int PER = SLOTS / NM > 0 ? SLOTS / NM : 1;
    if (K::NGRP == GPC) {  // no fold inside the region: the pinned interleave of the plain step
#pragma unroll
      for (int n = 0; n < (K::NGRP - 1) * NM; ++n) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (n < NWRITE) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        else if (n < NWRITE + NVMEM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        if (n % STRIDE == 0) __builtin_amdgcn_sched_group_barrier(0x100, PER, 0);
      }
    }
    __syncthreads();
    if (HAS_NEXT) fetch_group(nas, nbs, chunk_order<REVERSE, GPC>(0), a_mn, b_mn, lane, fa[0], fb[0]);
    mma_group(fa[(K::NGRP - 1) & 1], fb[(K::NGRP - 1) & 1]);
    fold();
  }

  // Position i in the visiting order -> fragment group of the tile.
  template <bool REVERSE, int GPC>
  __device__ static constexpr int chunk_order(int i) {
    return REVERSE ? (K::NGRP / GPC - 1 - i / GPC) * GPC + i % GPC : i;
  }

  template <bool REVERSE, int CHUNK>
  __device__ inline void run_chunked(const GemmArgs<T>& g, const T* a, const T* b, int64_t m0, int64_t n0,
                                     int64_t m_lim, int64_t n_lim, SBAMD_LDS T* smem, int tid, int a_mn, int b_mn,
                                     typename M::acc_t (&run)[K::MI][K::NI]) {
    constexpr int DIR = REVERSE ? -1 : 1;
    constexpr int GPC = CHUNK / (M::TK * M::GS);  // fragment groups per chunk
    static_assert(BK % CHUNK == 0 && CHUNK % (M::TK * M::GS) == 0, "a K step holds whole chunks");
    const int lane = tid & 63;
    const int nfull = int(g.k / BK);
    constexpr int NW = LoadA::NWRITE + LoadB::NWRITE;
    constexpr int NL = LoadA::NV + LoadB::NV;
    SBAMD_LDS T* const buf0 = smem;
    SBAMD_LDS T* const buf1 = smem + K::STAGE_ELEMS;
    // alpha is -1 here (checked on the host): (-1 * acc) + run is run - acc, bit for bit.
    auto fold = [&]() {
#pragma unroll
      for (int i = 0; i < K::MI; ++i)
#pragma unroll
        for (int j = 0; j < K::NI; ++j)
#pragma unroll
          for (int r = 0; r < M::NACC; ++r) {
            run[i][j][r] = run[i][j][r] - acc[i][j][r];
            acc[i][j][r] = T(0);
          }
    };

    const int64_t k_first = REVERSE ? g.k - BK : 0;
    la.init_fast(a, g.lda, m0, m_lim, k_first, tid);
    lb.init_fast(b, g.ldb, n0, n_lim, k_first, tid);
    la.template load_fast<DIR>(g.lda);
    lb.template load_fast<DIR>(g.ldb);
    la.store(buf0, tid);
    lb.store(buf0 + K::A_ELEMS, tid);
    if (nfull > 1) {
      la.template load_fast<DIR>(g.lda);
      lb.template load_fast<DIR>(g.ldb);
    }
    __syncthreads();
    frag_a_t fa;
    frag_b_t fb;
    fetch_group(buf0, buf0 + K::A_ELEMS, chunk_order<REVERSE, GPC>(0), a_mn, b_mn, lane, fa[0], fb[0]);

    int kt = 0;
    for (; kt + 2 < nfull; ++kt) {
      SBAMD_LDS T* cur = (kt & 1) ? buf1 : buf0;
      SBAMD_LDS T* nxt = (kt & 1) ? buf0 : buf1;
      la.store(nxt, tid);
      lb.store(nxt + K::A_ELEMS, tid);
      la.template load_fast<DIR>(g.lda);
      lb.template load_fast<DIR>(g.ldb);
      chunked_step<true, NW, NL, REVERSE, GPC>(cur, cur + K::A_ELEMS, nxt, nxt + K::A_ELEMS, a_mn, b_mn, lane, fa, fb,
                                               fold);
    }
    if (kt + 1 < nfull) {
      SBAMD_LDS T* cur = (kt & 1) ? buf1 : buf0;
      SBAMD_LDS T* nxt = (kt & 1) ? buf0 : buf1;
      la.store(nxt, tid);
      lb.store(nxt + K::A_ELEMS, tid);
      chunked_step<true, NW, 0, REVERSE, GPC>(cur, cur + K::A_ELEMS, nxt, nxt + K::A_ELEMS, a_mn, b_mn, lane, fa, fb,
                                              fold);
      ++kt;
    }
    {
      SBAMD_LDS T* cur = (kt & 1) ? buf1 : buf0;
      chunked_step<false, 0, 0, REVERSE, GPC>(cur, cur + K::A_ELEMS, cur, cur + K::A_ELEMS, a_mn, b_mn, lane, fa, fb,
                                              fold);
    }
  }

  // ---- edge tiles whose M (N) extent ends inside a 16-byte vector of a contiguous-M (N) operand: the
  // vector loads of the pipelined path with the straddling vector put in place by store_ragged, in the
  // simple loop below (one barrier per K step, loads of the next tile issued before the compute).  The K
  // tail is one step of element loads.  Needs m >= V (n >= V) for such an operand.
  __device__ inline void run_ragged(const GemmArgs<T>& g, const T* a, const T* b, int64_t m0, int64_t n0,
                                    int64_t k_begin, int64_t k_end, SBAMD_LDS T* smem, int tid, int a_mn,
                                    int b_mn) {
    const int lane = tid & 63;
    const int64_t klen = k_end > k_begin ? k_end - k_begin : 0;
    const int nk = int((klen + BK - 1) / BK);
    const int nfull = int(klen / BK);
    if (nk == 0) return;
    const int valid_m = int(g.m - m0 < BM ? g.m - m0 : BM);
    const int valid_n = int(g.n - n0 < BN ? g.n - n0 : BN);
    la.init_fast(a, g.lda, m0, g.m, k_begin, tid);
    lb.init_fast(b, g.ldb, n0, g.n, k_begin, tid);
    auto load = [&](int kt) {
      if (kt < nfull) {
        la.load_fast(g.lda);
        lb.load_fast(g.ldb);
      } else {
        la.load_edge(a, g.lda, m0, g.m, k_begin + int64_t(kt) * BK, k_end, tid);
        lb.load_edge(b, g.ldb, n0, g.n, k_begin + int64_t(kt) * BK, k_end, tid);
      }
    };
    auto store = [&](int kt) {
      SBAMD_LDS T* st = smem + (kt & 1) * K::STAGE_ELEMS;
      if (kt < nfull) {
        la.store_ragged(st, tid, valid_m);
        lb.store_ragged(st + K::A_ELEMS, tid, valid_n);
      } else {
        la.store(st, tid);
        lb.store(st + K::A_ELEMS, tid);
      }
    };
    load(0);
    store(0);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; ++kt) {
      load(kt + 1);
      compute<0>(smem + (kt & 1) * K::STAGE_ELEMS, smem + (kt & 1) * K::STAGE_ELEMS + K::A_ELEMS, a_mn, b_mn, lane);
      store(kt + 1);
      __syncthreads();
    }
    compute<0>(smem + (kt & 1) * K::STAGE_ELEMS, smem + (kt & 1) * K::STAGE_ELEMS + K::A_ELEMS, a_mn, b_mn, lane);
  }

  // ---- simple path (K ranges shorter than a K step, matrices narrower than a vector): element loads, one
  // barrier per K step, loads of the next tile issued before the compute.
  __device__ inline void run_edge(const GemmArgs<T>& g, const T* a, const T* b, int64_t m0, int64_t n0,
                                  int64_t k_begin, int64_t k_end, SBAMD_LDS T* smem, int tid, int a_mn,
                                  int b_mn) {
    const int lane = tid & 63;
    const int64_t klen = k_end > k_begin ? k_end - k_begin : 0;
    const int nk = int((klen + BK - 1) / BK);
    if (nk == 0) return;
    la.load_edge(a, g.lda, m0, g.m, k_begin, k_end, tid);
    lb.load_edge(b, g.ldb, n0, g.n, k_begin, k_end, tid);
    la.store(smem, tid);
    lb.store(smem + K::A_ELEMS, tid);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; ++kt) {
      const int64_t k0 = k_begin + int64_t(kt + 1) * BK;
      la.load_edge(a, g.lda, m0, g.m, k0, k_end, tid);
      lb.load_edge(b, g.ldb, n0, g.n, k0, k_end, tid);
      compute<0>(smem + (kt & 1) * K::STAGE_ELEMS, smem + (kt & 1) * K::STAGE_ELEMS + K::A_ELEMS, a_mn, b_mn, lane);
      la.store(smem + ((kt + 1) & 1) * K::STAGE_ELEMS, tid);
      lb.store(smem + ((kt + 1) & 1) * K::STAGE_ELEMS + K::A_ELEMS, tid);
      __syncthreads();
    }
    compute<0>(smem + (kt & 1) * K::STAGE_ELEMS, smem + (kt & 1) * K::STAGE_ELEMS + K::A_ELEMS, a_mn, b_mn, lane);
  }
};

// Epilogue of the tile kernels: C <- value (+ beta * C) for the MI x NI matrix tiles of a wavefront, NACC
// elements per lane and tile.  ptr(i, j, r) is the element's address, or nullptr past the edge of C;
// value(i, j, r) is alpha * (its sum).  The products and the sum round separately (-ffp-contract=off).
// Element by element (load, add, store, next load): every load sits behind a store to memory the compiler
// cannot tell apart from it, so with C read each element costs a memory round trip.  Tiles at the edge of C
// and 16-bit outputs live with that; interior tiles take store_tiles_interior below.
template <int MI, int NI, int NACC, typename V, typename PtrF, typename ValF, typename LoadF, typename StoreF>
__device__ inline void store_tiles(bool read_c, V beta, PtrF&& ptr, ValF&& value, LoadF&& load, StoreF&& store) {
#pragma unroll
  for (int t = 0; t < MI * NI; ++t)
#pragma unroll
    for (int r = 0; r < NACC; ++r) {
      auto* p = ptr(t / NI, t % NI, r);
      if (p) {
        V out = value(t / NI, t % NI, r);
        if (read_c) {
          const V bc = beta * load(p);
          out = out + bc;
        }
        store(p, out);
      }
    }
}

// The same epilogue for a wavefront tile that lies wholly inside C (elements of 2, 4 or 8 bytes in memory; 2-byte
// elements are halves, widened to / rounded from the float the arithmetic runs in).  Addresses are a
// wave-uniform buffer descriptor at the tile's origin, one 32-bit per-lane byte offset (the lane's row and its
// share of the column index) and a scalar offset per element (the rest of the column index times ldc): no
// 64-bit address per element, no exec-mask branch.  With C read, the loads of the next batch of 16 (8-byte
// elements: 8) are issued before the current batch is stored, so a lane has a batch of loads in flight instead
// of one (SGEMM 8192 x 8192 x 256, beta != 0: 518 us element by element against 284 us with beta == 0).
// `origin` must be wave-uniform; lane_row = the lane's row inside the tile, lane_col = its column share
// (acc_i(0, lane)), col_of(r) = acc_i(r, 0); the caller checks WTN * ldc * sizeof(TMem) < 2^31.
template <int MI, int NI, int NACC, int MT, typename V, typename TMem, typename ColF, typename ValF>
__device__ inline void store_tiles_interior(bool read_c, V beta, TMem* origin, int64_t ldc, int lane_row, int lane_col,
                                            ColF&& col_of, ValF&& value) {
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  static_assert(sizeof(TMem) == sizeof(V) || (sizeof(TMem) == 2 && sizeof(V) == 4), "same type, or halves behind floats");
  static_assert(sizeof(V) == 4 || sizeof(V) == 8, "4- or 8-byte arithmetic");
  constexpr int E = MI * NI * NACC;
  // elements per batch: 16 four-byte or 8 eight-byte loads in flight, less where the tile does not divide that way
  constexpr int B = sizeof(V) == 4 ? (E % 16 == 0 ? 16 : E % 12 == 0 ? 12 : E % 8 == 0 ? 8 : 4) : (E % 8 == 0 ? 8 : 4);
  static_assert(E % B == 0, "whole batches");
  // (readfirstlane returns a signed int: widen through uint32_t, or a low half with its top bit set smears
  // ones over the high half)
  const uint32_t origin_hi = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(uintptr_t(origin) >> 32))));
  const uint32_t origin_lo = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(uintptr_t(origin)))));
  const uint64_t uniform = (uint64_t(origin_hi) << 32) | uint64_t(origin_lo);
  const __amdgpu_buffer_rsrc_t rc =
      __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<TMem*>(uniform), 0, 0x7fffffff, 0x00020000);
  const int ldc_bytes = int(ldc) * int(sizeof(TMem));
  const int voff0 = lane_row * int(sizeof(TMem)) + lane_col * ldc_bytes;
  auto voff = [&](int e) { return voff0 + ((e / NACC) / NI) * MT * int(sizeof(TMem)); };
  auto soff = [&](int e) { return (((e / NACC) % NI) * MT + col_of(e % NACC)) * ldc_bytes; };
  auto load = [&](int e) -> V {
    if constexpr (sizeof(TMem) == 2) {
      const unsigned short h = __builtin_amdgcn_raw_buffer_load_b16(rc, voff(e), soff(e), 0);
      return V(__builtin_bit_cast(_Float16, h));
    } else if constexpr (sizeof(V) == 4) {
      return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b32(rc, voff(e), soff(e), 0));
    } else {
      return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b64(rc, voff(e), soff(e), 0));
    }
  };
  auto store = [&](int e, V v) {
    if constexpr (sizeof(TMem) == 2)
      __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, _Float16(v)), rc, voff(e), soff(e), 0);
    else if constexpr (sizeof(V) == 4)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), rc, voff(e), soff(e), 0);
    else
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rc, voff(e), soff(e), 0);
  };
  auto val = [&](int e) -> V { return value((e / NACC) / NI, (e / NACC) % NI, e % NACC); };
  if (!read_c) {
#pragma unroll
    for (int e = 0; e < E; ++e) store(e, val(e));
    return;
  }
  V cin[2][B];
#pragma unroll
  for (int q = 0; q < B; ++q) cin[0][q] = load(q);
#pragma unroll
  for (int b0 = 0; b0 < E; b0 += B) {
    const int cur = (b0 / B) & 1;
    if (b0 + B < E) {
#pragma unroll
      for (int q = 0; q < B; ++q) cin[cur ^ 1][q] = load(b0 + B + q);
    }
#pragma unroll
    for (int q = 0; q < B; ++q) {
      V out = val(b0 + q);
      const V bc = beta * cin[cur][q];
      out = out + bc;
      store(b0 + q, out);
    }
  }
}

// OCC = waves per SIMD the register allocation must leave room for.
template <typename T, int BM, int BN, int BK, int WM, int WN, bool TA, bool TB, int OCC, typename MS = Mfma<T>>
__global__ void __launch_bounds__(WM* WN * 64, OCC)
    gemm_mfma_kernel(const GemmArgs<T> g) {
  using K = GemmTile<T, BM, BN, BK, WM, WN, TA, TB, MS>;
  using M = MS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  SBAMD_LDS T* smem = (SBAMD_LDS T*)smem_raw;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave % WM;
  const int wn = wave / WM;

  // ---- which tile / batch item / K slice
  const int ntiles = g.tiles_m * g.tiles_n;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int z = logical / ntiles;
  int tm, tn;
  tile_coords(logical - z * ntiles, g.tiles_m, g.tiles_n, tm, tn, g.group_m);
  const int64_t m0 = int64_t(tm) * BM;
  const int64_t n0 = int64_t(tn) * BN;

  const T* a = g.a;
  const T* b = g.b;
  T* c = g.c;
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

  GemmBody<T, BM, BN, BK, WM, WN, TA, TB, false, MS> body;
#pragma unroll
  for (int i = 0; i < K::MI; ++i)
#pragma unroll
    for (int j = 0; j < K::NI; ++j)
#pragma unroll
      for (int r = 0; r < M::NACC; ++r) body.acc[i][j][r] = T(0);

  const int frag_mn = M::lane_mn(lane);
  const int a_mn = wm * K::WTM + frag_mn;
  const int b_mn = wn * K::WTN + frag_mn;
  // The vector path also serves tiles that hang over the edge, unless a 16-byte
  // vector along M (N) could straddle it.
  const bool interior = (m0 + BM <= g.m) && (n0 + BN <= g.n);
  const bool clampable = (TA || g.m % M::V == 0) && (!TB || g.n % M::V == 0);
  const bool vectors_fit = (TA || g.m >= M::V) && (!TB || g.n >= M::V);
  if ((interior || clampable) && k_end - k_begin >= BK) {
    body.run_pipelined(g, a, b, m0, n0, k_begin, k_end, smem, tid, a_mn, b_mn);
  } else if (!interior && !clampable && vectors_fit) {
    body.run_ragged(g, a, b, m0, n0, k_begin, k_end, smem, tid, a_mn, b_mn);
  } else {
    body.run_edge(g, a, b, m0, n0, k_begin, k_end, smem, tid, a_mn, b_mn);
  }

  // ---- epilogue: C = alpha * acc (+ beta * C); products and the sum round
  // separately (the file is built with -ffp-contract=off).
  const T alpha = g.alpha;
  const bool read_c = !g.beta_zero && g.split_k <= 1;
  auto value = [&](int i, int j, int r) -> T { return alpha * body.acc[i][j][r]; };
  if (interior && g.ldc * int64_t(K::WTN * sizeof(T)) < (int64_t(1) << 31)) {
    store_tiles_interior<K::MI, K::NI, M::NACC, M::T, T, T>(
        read_c, g.beta, c + (m0 + wm * K::WTM) + (n0 + wn * K::WTN) * g.ldc, g.ldc, frag_mn, M::acc_i(0, lane),
        [](int r) { return M::acc_i(r, 0); }, value);
    return;
  }
  store_tiles<K::MI, K::NI, M::NACC, T>(
      read_c, g.beta,
      [&](int i, int j, int r) -> T* {
        const int64_t mm = m0 + a_mn + i * M::T;
        const int64_t nn = n0 + wn * K::WTN + j * M::T + M::acc_i(r, lane);
        return (mm < g.m && nn < g.n) ? c + mm + nn * g.ldc : nullptr;
      },
      value, [](const T* p) -> T { return *p; }, [](T* p, T v) { *p = v; });
}

// ------------------------------------------ two K halves in one workgroup
// Split-K with depth 2 without the cube: the workgroup holds two sets of WM x WN
// wavefronts, set 0 works on the first half of K and set 1 on the second, each with
// its own pair of LDS buffers, in lockstep (the halves are equal and whole K steps,
// so both sets run the same number of steps and meet at the same barriers).  At the
// end set 1 hands alpha * acc over through LDS and set 0 stores
//     (alpha * acc0 + alpha * acc1) + beta * C,
// which is, rounding for rounding, what the cube route produces for two slabs
// (splitk_reduce_kernel): the same result without the second launch and without
// writing and re-reading the partial sums through HBM.  It doubles the wavefronts per
// SIMD for shapes whose tiles alone leave the matrix pipes half idle (one 4-wave
// workgroup per compute unit).  The host checks that k splits into two equal halves of at least one K step whose start stays
// 16-byte aligned (both sets then run the same steps, K tail included) and sets
// g.k_chunk = k / 2.
// SETS = 4 is the same with four sets on four quarters of K, summed in slab order
//     ((alpha * acc0 + alpha * acc1) + alpha * acc2) + alpha * acc3,  then + beta * C:
// the cube route's result for four slabs.
template <typename T, int BM, int BN, int BK, int WM, int WN, bool TA, bool TB, int OCC, typename MS = Mfma<T>, int SETS = 2>
__global__ void __launch_bounds__(SETS * WM* WN * 64, OCC)
    gemm_mfma_halves_kernel(const GemmArgs<T> g) {
  using K = GemmTile<T, BM, BN, BK, WM, WN, TA, TB, MS>;
  using M = MS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  SBAMD_LDS T* smem = (SBAMD_LDS T*)smem_raw;

  const int half = threadIdx.x / K::NT;  // which set: wave-uniform, NT is a multiple of 64
  const int tid = threadIdx.x - half * K::NT;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave % WM;
  const int wn = wave / WM;

  const int ntiles = g.tiles_m * g.tiles_n;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int z = logical / ntiles;
  int tm, tn;
  tile_coords(logical - z * ntiles, g.tiles_m, g.tiles_n, tm, tn, g.group_m);
  const int64_t m0 = int64_t(tm) * BM;
  const int64_t n0 = int64_t(tn) * BN;
  const T* a = g.a + int64_t(z) * g.stride_a;
  const T* b = g.b + int64_t(z) * g.stride_b;
  T* c = g.c + int64_t(z) * g.stride_c;
  const int64_t k_begin = int64_t(half) * g.k_chunk;
  const int64_t k_end = k_begin + g.k_chunk;

  // (fp32 128 x 128 tile, A 'n', B 't': in this kernel hipcc sinks the global loads of the K step behind its last
  // matrix instructions, which it does not do for the same tile in gemm_mfma_kernel; fenced, +1-2 %)
  GemmBody<T, BM, BN, BK, WM, WN, TA, TB, (sizeof(T) == 4 && K::MI * K::NI == 2 && !TA && TB), MS> body;
#pragma unroll
  for (int i = 0; i < K::MI; ++i)
#pragma unroll
    for (int j = 0; j < K::NI; ++j)
#pragma unroll
      for (int r = 0; r < M::NACC; ++r) body.acc[i][j][r] = T(0);

  const int frag_mn = M::lane_mn(lane);
  const int a_mn = wm * K::WTM + frag_mn;
  const int b_mn = wn * K::WTN + frag_mn;
  SBAMD_LDS T* mine = smem + half * (2 * K::STAGE_ELEMS);
  const bool interior = (m0 + BM <= g.m) && (n0 + BN <= g.n);
  const bool clampable = (TA || g.m % M::V == 0) && (!TB || g.n % M::V == 0);
  const bool vectors_fit = (TA || g.m >= M::V) && (!TB || g.n >= M::V);
  if (interior || clampable) {
    body.run_pipelined(g, a, b, m0, n0, k_begin, k_end, mine, tid, a_mn, b_mn);
  } else if (vectors_fit) {
    body.run_ragged(g, a, b, m0, n0, k_begin, k_end, mine, tid, a_mn, b_mn);
  } else {
    body.run_edge(g, a, b, m0, n0, k_begin, k_end, mine, tid, a_mn, b_mn);
  }

  // hand-over: [set - 1][wave][subtile][register][lane], lane fastest
  const T alpha = g.alpha, beta = g.beta;
  constexpr int SET_SLOTS = WM * WN * K::MI * K::NI * M::NACC * 64;
  __syncthreads();
  SBAMD_LDS T* slot = smem + (half > 0 ? (half - 1) * SET_SLOTS : 0) + (wave * K::MI * K::NI * M::NACC) * 64 + lane;
  if (half >= 1) {
#pragma unroll
    for (int i = 0; i < K::MI; ++i)
#pragma unroll
      for (int j = 0; j < K::NI; ++j)
#pragma unroll
        for (int r = 0; r < M::NACC; ++r) slot[((i * K::NI + j) * M::NACC + r) * 64] = alpha * body.acc[i][j][r];
  }
  __syncthreads();
  if (half >= 1) return;
  store_tiles<K::MI, K::NI, M::NACC, T>(
      !g.beta_zero, beta,
      [&](int i, int j, int r) -> T* {
        const int64_t mm = m0 + a_mn + i * M::T;
        const int64_t nn = n0 + wn * K::WTN + j * M::T + M::acc_i(r, lane);
        return (mm < g.m && nn < g.n) ? c + mm + nn * g.ldc : nullptr;
      },
      [&](int i, int j, int r) -> T {
        T sum = alpha * body.acc[i][j][r];
#pragma unroll
        for (int other = 0; other < SETS - 1; ++other) sum = sum + slot[other * SET_SLOTS + ((i * K::NI + j) * M::NACC + r) * 64];
        return sum;
      },
      [](const T* p) -> T { return *p; }, [](T* p, T v) { *p = v; });
}

// ------------------------------------------- operand tiles straight into LDS (fp32)
// The register-staged pipeline above spends a global load, a wait, an LDS store and their
// address arithmetic per 16 bytes of operand; a compute unit that holds only a few wavefronts
// (mid-size shapes: one or two 64x64 tiles per unit) cannot hide that (timing-only ablation at
// 1024^3: 29 us, 19 us with the staging switched off).  Here the tiles go global -> LDS with
// global_load_lds_dwordx4: no registers, no LDS store instruction, and the only wait is a counted
// s_waitcnt in front of the one barrier per K step.  A load instruction writes 64 lanes x 16 bytes
// = 1 KiB of LDS contiguously, so the images are linear:
//   M/N-contiguous operand: [BK][BMN], exactly the register-staged image (fp32 needs no padding);
//   K-contiguous operand:   [BMN][BK], rows of BK / 4 16-byte pieces; piece c of row r sits at
//                           position c ^ ((r >> 1) & (BK / 4 - 1)), chosen by WHICH piece each lane
//                           fetches.  A fragment is the dword pair (k, k + 2) of one piece.
// The matrix instructions, their order and the epilogue are those of gemm_mfma_kernel, so the two
// kernels agree bit for bit.  The host checks: at least two whole K steps, 16-byte-loadable operands, and
// no 16-byte vector straddling an edge (rows / vectors past the edge are clamped as above).
typedef __attribute__((address_space(1))) const void* dma_src_t;
typedef __attribute__((address_space(3))) void* dma_dst_t;

template <int BMN, int BK, int NWAVES, bool KMAJOR>
struct DmaOperand {
  static constexpr int ELEMS = BMN * BK;             // floats per image
  static constexpr int PIECES = ELEMS * 4 / 1024;    // load instructions per image
  static_assert(PIECES % NWAVES == 0, "the image splits evenly over the wavefronts");
  static constexpr int PER_WAVE = PIECES / NWAVES;
  static constexpr int ROW_PIECES = BK / 4;          // K-contiguous rows: 16-byte pieces per row
  static constexpr int ROWS_PER_LOAD = 64 / ROW_PIECES;
  static constexpr int KROWS_PER_LOAD = 256 / BMN;   // M/N-contiguous: k rows per load instruction

  const float* src[PER_WAVE];

  __device__ inline void init(const float* __restrict__ x, int64_t ld, int64_t mn0, int64_t mn_lim, int64_t k0,
                              int wave, int lane) {
#pragma unroll
    for (int u = 0; u < PER_WAVE; ++u) {
      const int j = wave + NWAVES * u;
      if (KMAJOR) {
        const int row = j * ROWS_PER_LOAD + lane / ROW_PIECES;
        const int c = (lane % ROW_PIECES) ^ ((row >> 1) & (ROW_PIECES - 1));
        int64_t mn = mn0 + row;
        mn = mn < mn_lim ? mn : mn_lim - 1;
        src[u] = x + mn * ld + k0 + c * 4;
      } else {
        const int krow = j * KROWS_PER_LOAD + lane / (BMN / 4);
        int64_t mn = mn0 + (lane % (BMN / 4)) * 4;
        mn = mn < mn_lim ? mn : mn_lim - 4;
        src[u] = x + mn + (k0 + krow) * ld;
      }
    }
  }

  // Request the tile the pointers address into the image at `lds`, then move on one K step.
  __device__ inline void issue(SBAMD_LDS float* lds, int64_t ld, int wave) {
    const int64_t step = KMAJOR ? int64_t(BK) : int64_t(BK) * ld;
#pragma unroll
    for (int u = 0; u < PER_WAVE; ++u) {
      __builtin_amdgcn_global_load_lds((dma_src_t)src[u], (dma_dst_t)(lds + (wave + NWAVES * u) * 256), 16, 0, 0);
      src[u] += step;
    }
  }

  // Operands of one lane for MFMA k-steps 2 * grp and 2 * grp + 1 (k = 4 grp + lk and 4 grp + 2 + lk).
  __device__ static inline void fragment(const SBAMD_LDS float* lds, int grp, int mn, int lane, float (&out)[2]) {
    const int lk = lane >> 5;
    if (KMAJOR) {
      const SBAMD_LDS float* p = lds + mn * BK + ((grp ^ ((mn >> 1) & (ROW_PIECES - 1))) << 2) + lk;
      out[0] = p[0];
      out[1] = p[2];
    } else {
      out[0] = lds[(4 * grp + lk) * BMN + mn];
      out[1] = lds[(4 * grp + 2 + lk) * BMN + mn];
    }
  }
};

template <int BM, int BN, int BK, int WM, int WN, bool TA, bool TB, int OCC, int STAGES>
__global__ void __launch_bounds__(WM* WN * 64, OCC)
    gemm_mfma_dma_kernel(const GemmArgs<float> g) {
  using M = Mfma<float>;
  constexpr int NWAVES = WM * WN;
  constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 32, NI = WTN / 32, NGRP = BK / 4;
  using OpA = DmaOperand<BM, BK, NWAVES, TA>;
  using OpB = DmaOperand<BN, BK, NWAVES, !TB>;
  constexpr int STAGE = OpA::ELEMS + OpB::ELEMS;
  constexpr int IN_FLIGHT = (STAGES - 2) * (OpA::PER_WAVE + OpB::PER_WAVE);  // loads that may stay out at the barrier
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
  SBAMD_LDS float* smem = (SBAMD_LDS float*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int ntiles = g.tiles_m * g.tiles_n;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int z = logical / ntiles;
  int tm, tn;
  tile_coords(logical - z * ntiles, g.tiles_m, g.tiles_n, tm, tn, g.group_m);
  const int64_t m0 = int64_t(tm) * BM, n0 = int64_t(tn) * BN;

  const float* a = g.a;
  const float* b = g.b;
  float* c = g.c;
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
  const int nk = int((k_end - k_begin) / BK);  // whole K steps; the rest is the tail step at the end

  OpA oa;
  OpB ob;
  oa.init(a, g.lda, m0, g.m, k_begin, wave, lane);
  ob.init(b, g.ldb, n0, g.n, k_begin, wave, lane);

  M::acc_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  const int a_mn = wm * WTM + (lane & 31), b_mn = wn * WTN + (lane & 31);

  // tiles 0 .. STAGES-2 go in flight; tile 0 has landed (this wave's pieces) before the barrier
#pragma unroll
  for (int t = 0; t < STAGES - 1; ++t)
    if (t < nk) {
      oa.issue(smem + t * STAGE, g.lda, wave);
      ob.issue(smem + t * STAGE + OpA::ELEMS, g.ldb, wave);
    }
  if (nk >= STAGES - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IN_FLIGHT) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  static_assert(NGRP % 2 == 0, "the fragment double buffer must come back to slot 0");
  float fa[2][MI][2], fb[2][NI][2];
  auto fetch = [&](const SBAMD_LDS float* st, int grp, int slot) {
#pragma unroll
    for (int i = 0; i < MI; ++i) OpA::fragment(st, grp, a_mn + i * 32, lane, fa[slot][i]);
#pragma unroll
    for (int j = 0; j < NI; ++j) OpB::fragment(st + OpA::ELEMS, grp, b_mn + j * 32, lane, fb[slot][j]);
  };
  auto mma = [&](int slot) {
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = M::mma(fb[slot][j][e], fa[slot][i][e], acc[i][j]);
  };
  if (nk > 0) fetch(smem, 0, 0);

  // One K step: the barrier sits in front of the LAST fragment group's matrix instructions.  By then
  // every fragment of tile t is in registers (its stage is refilled at the top of the next step) and this
  // wave's pieces of tile t + 1 have landed, so the first fragments of tile t + 1 are fetched behind the
  // barrier and their latency hides under the last group's matrix instructions.
  int cur = 0, fill = STAGES - 1;
  for (int t = 0; t < nk; ++t) {
    const bool more = t + STAGES - 1 < nk;
    if (more) {
      oa.issue(smem + fill * STAGE, g.lda, wave);
      ob.issue(smem + fill * STAGE + OpA::ELEMS, g.ldb, wave);
    }
    const SBAMD_LDS float* st = smem + cur * STAGE;
#pragma unroll
    for (int grp = 0; grp + 1 < NGRP; ++grp) {
      fetch(st, grp + 1, (grp + 1) & 1);
      mma(grp & 1);
    }
    if (more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(IN_FLIGHT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    cur = cur + 1 == STAGES ? 0 : cur + 1;
    fill = fill + 1 == STAGES ? 0 : fill + 1;
    if (t + 1 < nk) fetch(smem + cur * STAGE, 0, 0);
    mma((NGRP - 1) & 1);
  }

  // K tail (< BK): one more step through registers.  The M/N-contiguous register-staged image is the same
  // linear [k][mn] image, k past the end stored as exact zeros; every wave has passed the last barrier and no
  // load is outstanding, so stage 0 is free.
  const int64_t k_tail = k_begin + int64_t(nk) * BK;
  if (k_tail < k_end) {
    TileLoader<float, BM, BK, NWAVES * 64, false> ea;
    TileLoader<float, BN, BK, NWAVES * 64, false> eb;
    ea.load_edge(a, g.lda, m0, g.m, k_tail, k_end, tid);
    eb.load_edge(b, g.ldb, n0, g.n, k_tail, k_end, tid);
    ea.store(smem, tid);
    eb.store(smem + OpA::ELEMS, tid);
    __syncthreads();
    fetch(smem, 0, 0);
#pragma unroll
    for (int grp = 0; grp < NGRP; ++grp) {
      if (grp + 1 < NGRP) fetch(smem, grp + 1, (grp + 1) & 1);
      mma(grp & 1);
    }
  }

  const float alpha = g.alpha;
  const bool read_c = !g.beta_zero && g.split_k <= 1;
  store_tiles<MI, NI, 16, float>(
      read_c, g.beta,
      [&](int i, int j, int r) -> float* {
        const int64_t mm = m0 + a_mn + i * 32;
        const int64_t nn = n0 + wn * WTN + j * 32 + M::acc_i(r, lane);
        return (mm < g.m && nn < g.n) ? c + mm + nn * g.ldc : nullptr;
      },
      [&](int i, int j, int r) -> float { return alpha * acc[i][j][r]; }, [](const float* p) -> float { return *p; },
      [](float* p, float v) { *p = v; });
}

// ------------------------------------------- batches of one-tile, one-step items
// Strided batches whose items are a single tile with a single K step (m, n <= BMN, k == BK: the
// batched 64x64x64 class) are bound by HBM, not by the matrix pipes: each item moves three
// tiles and does one K step of work.  A workgroup per item leaves the loads of an item exposed
// behind its launch; here a fixed set of workgroups walks the batch (item z, z + grid, ...) and
// the operands of the NEXT item are requested before the matrix work of the current one starts,
// so every resident workgroup keeps a whole item in flight all the time.  One LDS stage: the
// registers are the second buffer.  Per element the arithmetic is the chain of the tile kernel
// (same fragments, same order, same epilogue), so the two kernels agree bit for bit.  The host
// checks k == BK, 16-byte-loadable operands, that no vector straddles the edge and that the
// byte offsets inside an item fit 32 bits.  NT_LOADS marks the operand loads non-temporal
// (cache policy bit 1 of the buffer load): a batch larger than the 256 MB Infinity Cache
// streams past it (fp32 x 16384 items: 156 -> 136 us), a smaller one is better off cached.
template <typename T, int BMN, int BK, bool TA, bool TB, bool READ_C, bool NT_LOADS, int OCC>
__global__ void __launch_bounds__(256, OCC)
    gemm_batch_stream_kernel(const GemmArgs<T> g) {
  using K = GemmTile<T, BMN, BMN, BK, 2, 2, TA, TB>;
  using M = Mfma<T>;
  using LoadA = typename K::LoadA;
  using LoadB = typename K::LoadB;
  using vec_t = typename M::vec_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  SBAMD_LDS T* smem = (SBAMD_LDS T*)smem_raw;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave & 1;
  const int wn = wave >> 1;
  const int frag_mn = M::lane_mn(lane);
  const int a_mn = wm * K::WTM + frag_mn;
  const int b_mn = wn * K::WTN + frag_mn;
  const T alpha = g.alpha, beta = g.beta;

  // Tile walk (g.tiles_m > 0): the items are the BMN x BMN tiles of ONE product whose K is the K step (the
  // host checks that M and N are multiples of BMN) -- tile (tm, tn) reads row panel tm of A and column panel
  // tn of B, which stay in the L2 (the panels of a whole 8192 x 8192 x 64 product are 4 MB), and the walk is
  // bound by writing C once; the next tile's panels are requested before the current tile's matrix work
  // and its C stores drain while the next tile is multiplied, which a launch of one workgroup per tile
  // (or one per compute unit) does not overlap.
  const bool walk = g.tiles_m > 0;
  const int64_t nitems = walk ? int64_t(g.tiles_m) * int64_t(g.tiles_n) : int64_t(g.batch);
  auto item_a = [&](int64_t item) -> const T* {
    if (!walk) return g.a + item * g.stride_a;
    return g.a + (item % g.tiles_m) * BMN * (TA ? g.lda : int64_t(1));
  };
  auto item_b = [&](int64_t item) -> const T* {
    if (!walk) return g.b + item * g.stride_b;
    return g.b + (item / g.tiles_m) * BMN * (TB ? int64_t(1) : g.ldb);
  };
  auto item_c = [&](int64_t item) -> T* {
    if (!walk) return g.c + item * g.stride_c;
    return g.c + (item % g.tiles_m) * BMN + (item / g.tiles_m) * BMN * g.ldc;
  };

  // Where a thread's vectors sit inside an item does not depend on the item: 32-bit byte
  // offsets computed once, added to the item's (uniform) base address by the load itself.
  // Rows / vectors past the edge point at the last valid one, as in the tile kernel.
  const uint32_t m = walk ? uint32_t(BMN) : uint32_t(g.m), n = walk ? uint32_t(BMN) : uint32_t(g.n);
  const uint32_t lda = uint32_t(g.lda), ldb = uint32_t(g.ldb), ldc = uint32_t(g.ldc);
  uint32_t offa[LoadA::NV], offb[LoadB::NV];
#pragma unroll
  for (int i = 0; i < LoadA::NV; ++i) {
    const uint32_t v = tid + i * 256, row = v / LoadA::PER_ROW, col = (v % LoadA::PER_ROW) * M::V;
    offa[i] = uint32_t(sizeof(T)) * (TA ? col + (row < m ? row : m - 1) * lda : (col < m ? col : m - M::V) + row * lda);
  }
#pragma unroll
  for (int i = 0; i < LoadB::NV; ++i) {
    const uint32_t v = tid + i * 256, row = v / LoadB::PER_ROW, col = (v % LoadB::PER_ROW) * M::V;
    offb[i] = uint32_t(sizeof(T)) * (!TB ? col + (row < n ? row : n - 1) * ldb : (col < n ? col : n - M::V) + row * ldb);
  }
  const uint32_t c_row = a_mn;                                   // this lane's M index
  const uint32_t c_col0 = wn * K::WTN + M::acc_i(0, lane);       // its first N index
  const uint32_t offc0 = uint32_t(sizeof(T)) * (c_row + c_col0 * ldc);

  GemmBody<T, BMN, BMN, BK, 2, 2, TA, TB> body;
  // Buffer addressing: the item's base goes into a (uniform) descriptor and the per-thread
  // part stays a 32-bit offset, so no 64-bit address per vector has to live across the
  // matrix work; accesses whose offset is out of range are dropped by the range check,
  // which is how elements past the edge of C are masked (offset all ones).
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  constexpr int RANGE = 0x7fffffff;
  auto request = [&](int64_t item) {
    const __amdgpu_buffer_rsrc_t ra =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(item_a(item)), 0, RANGE, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(item_b(item)), 0, RANGE, 0x00020000);
#pragma unroll
    for (int i = 0; i < LoadA::NV; ++i) {
      const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(ra, int(offa[i]), 0, NT_LOADS ? 2 : 0);
      const vec_t vv = __builtin_bit_cast(vec_t, raw);
#pragma unroll
      for (int e = 0; e < M::V; ++e) body.la.reg[i][e] = vv[e];
    }
#pragma unroll
    for (int i = 0; i < LoadB::NV; ++i) {
      const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rb, int(offb[i]), 0, NT_LOADS ? 2 : 0);
      const vec_t vv = __builtin_bit_cast(vec_t, raw);
#pragma unroll
      for (int e = 0; e < M::V; ++e) body.lb.reg[i][e] = vv[e];
    }
  };
  // one C element of this lane: byte offset, or all ones past the edge
  auto c_offset = [&](int i, int j, int r) -> int {
    const uint32_t dm = i * M::T, dn = j * M::T + M::acc_i(r, 0);
    const bool live = c_row + dm < m && c_col0 + dn < n;
    return live ? int(offc0 + uint32_t(sizeof(T)) * (dm + dn * ldc)) : -1;
  };
  auto c_load = [&](__amdgpu_buffer_rsrc_t rc, int off) -> T {
    if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(rc, off, 0, 0));
    else return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(rc, off, 0, 0));
  };
  auto c_store = [&](__amdgpu_buffer_rsrc_t rc, int off, T v) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    if constexpr (sizeof(T) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), rc, off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rc, off, 0, 0);
  };

  int64_t z = blockIdx.x;
  if (z >= nitems) return;  // uniform over the workgroup
  request(z);

  for (;;) {
    body.la.store(smem, tid);
    body.lb.store(smem + K::A_ELEMS, tid);
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(item_c(z), 0, RANGE, 0x00020000);
    // beta * C operands first, then the next item's tiles: the memory counter retires
    // in order, so the epilogue can wait for the former while the latter stay in flight.
    T cin[K::MI][K::NI][M::NACC];
    if (READ_C) {
#pragma unroll
      for (int i = 0; i < K::MI; ++i)
#pragma unroll
        for (int j = 0; j < K::NI; ++j)
#pragma unroll
          for (int r = 0; r < M::NACC; ++r) cin[i][j][r] = c_load(rc, c_offset(i, j, r));
    }
    const int64_t zn = z + gridDim.x;
#ifdef SBAMD_WALK_ABLATE
    if (zn < nitems && !(SBAMD_WALK_ABLATE & 2)) request(zn);
#else
    if (zn < nitems) request(zn);
#endif
    __builtin_amdgcn_sched_barrier(0);  // the requests go out before the matrix work
#pragma unroll
    for (int i = 0; i < K::MI; ++i)
#pragma unroll
      for (int j = 0; j < K::NI; ++j)
#pragma unroll
        for (int r = 0; r < M::NACC; ++r) body.acc[i][j][r] = T(0);
#ifdef SBAMD_WALK_ABLATE
    if (!(SBAMD_WALK_ABLATE & 4))
#endif
    body.template compute<0>(smem, smem + K::A_ELEMS, a_mn, b_mn, lane);
#pragma unroll
    for (int i = 0; i < K::MI; ++i)
#pragma unroll
      for (int j = 0; j < K::NI; ++j)
#pragma unroll
        for (int r = 0; r < M::NACC; ++r) {
          T out = alpha * body.acc[i][j][r];
          if (READ_C) {
            const T bc = beta * cin[i][j][r];
            out = out + bc;
          }
#ifdef SBAMD_WALK_ABLATE
          if ((SBAMD_WALK_ABLATE & 1) && zn < nitems) continue;  // only the last tile of a workgroup is stored
#endif
          c_store(rc, c_offset(i, j, r), out);
        }
    if (zn >= nitems) break;
    z = zn;
    __syncthreads();  // every wave is done with the fragments before the stage is overwritten
  }
}

// ------------------------------------------------ chunked-accumulation GEMM
// C <- the result of k / CHUNK successive GEMM calls, chunk q being
//     C <- -1 * op(A)[:, chunk q] * op(B)[chunk q, :] + beta_q * C,
// beta_0 = g.beta (C is not read when g.beta_zero) and beta_q = 1 afterwards, with C
// held in registers between the chunks.  Multiplying by one is exact, so beta is applied once, when C is read.
// The triangular solver uses it to apply the updates of many 16-wide diagonal blocks
// to a panel in one launch with the rounding of the reference's block-by-block GEMM
// sequence (src/interface/trsm_interface.hpp:227-380).  The host checks k % BK == 0,
// k >= BK (BK = the K step of the pipeline, a multiple of CHUNK) and 16-byte alignment of
// both operands.  m_lim / n_lim bound the 16-byte loads along M / N: the matrix extent for a
// K-major operand, the extent rounded up to a whole vector for an M/N-major one -- whose
// storage must then exist up to there (what a straddling vector picks up only feeds rows or
// columns that are never stored).
template <typename T, int BM, int BN, int BK, int CHUNK, int WM, int WN, bool TA, bool TB, bool REVERSE, int OCC>
__global__ void __launch_bounds__(WM* WN * 64, OCC)
    gemm_chunked_kernel(const GemmArgs<T> g, int64_t m_lim, int64_t n_lim) {
  using K = GemmTile<T, BM, BN, BK, WM, WN, TA, TB>;
  using M = Mfma<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  SBAMD_LDS T* smem = (SBAMD_LDS T*)smem_raw;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave % WM;
  const int wn = wave / WM;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  int tm, tn;
  tile_coords(logical, g.tiles_m, g.tiles_n, tm, tn, g.group_m);
  const int64_t m0 = int64_t(tm) * BM;
  const int64_t n0 = int64_t(tn) * BN;
  const int frag_mn = M::lane_mn(lane);
  const int a_mn = wm * K::WTM + frag_mn;
  const int b_mn = wn * K::WTN + frag_mn;

  GemmBody<T, BM, BN, BK, WM, WN, TA, TB> body;
  typename M::acc_t run[K::MI][K::NI];
  const T beta = g.beta;
#pragma unroll
  for (int i = 0; i < K::MI; ++i) {
    const int64_t mm = m0 + a_mn + i * M::T;
#pragma unroll
    for (int j = 0; j < K::NI; ++j) {
#pragma unroll
      for (int r = 0; r < M::NACC; ++r) {
        const int64_t nn = n0 + wn * K::WTN + j * M::T + M::acc_i(r, lane);
        body.acc[i][j][r] = T(0);
        run[i][j][r] = (!g.beta_zero && mm < g.m && nn < g.n) ? beta * g.c[mm + nn * g.ldc] : T(0);
      }
    }
  }

  body.template run_chunked<REVERSE, CHUNK>(g, g.a, g.b, m0, n0, m_lim, n_lim, smem, tid, a_mn, b_mn, run);

#pragma unroll
  for (int i = 0; i < K::MI; ++i) {
    const int64_t mm = m0 + a_mn + i * M::T;
#pragma unroll
    for (int j = 0; j < K::NI; ++j) {
#pragma unroll
      for (int r = 0; r < M::NACC; ++r) {
        const int64_t nn = n0 + wn * K::WTN + j * M::T + M::acc_i(r, lane);
        if (mm < g.m && nn < g.n) g.c[mm + nn * g.ldc] = run[i][j][r];
      }
    }
  }
}

// --------------------------------------------------- split-K cube reduction
// out = sum over slices of the cube (+ beta * C).  Slices are added in index
// order, one C element per thread.  Counterpart of the reference's
// Reduction<AddOperator, outer> + the `temp + beta*C` assignment
// (src/operations/extension/reduction.hpp:152-230,
//  src/sb_handle/portblas_handle.hpp:354-395).
template <typename T>
__global__ void __launch_bounds__(256)
    splitk_reduce_kernel(const T* __restrict__ cube, int64_t slice_stride, int depth, T* c,
                         int64_t m, int64_t n, int64_t ldc, T beta, int beta_zero) {
  const int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= m * n) return;
  const int64_t row = idx % m;
  const int64_t col = idx / m;
  // Loads go out eight slices at a time (they are independent); the additions
  // keep the slice order, so the rounding does not depend on the batching.
  const T* src = cube + idx;
  T sum = *src;
  int d = 1;
  for (; d + 8 <= depth; d += 8) {
    T v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = src[int64_t(d + e) * slice_stride];
#pragma unroll
    for (int e = 0; e < 8; ++e) sum = sum + v[e];
  }
  for (; d < depth; ++d) sum = sum + src[int64_t(d) * slice_stride];
  T* p = c + row + col * ldc;
  if (!beta_zero) {
    const T bc = beta * (*p);
    sum = sum + bc;
  }
  *p = sum;
}

__device__ inline float fma_rn(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ inline double fma_rn(double a, double b, double c) { return __builtin_fma(a, b, c); }

// ------------------------------------------------------------- naive kernel
// One thread per C element, k in increasing order, fused multiply-add:
// the device twin of the reference's naive Gemm::eval
// (src/operations/blas3/gemm_ref.hpp:195-260).
template <typename T, bool TA, bool TB>
__global__ void __launch_bounds__(256) gemm_naive_kernel(const GemmArgs<T> g) {
  const int64_t per = g.m * g.n;
  const int64_t gid = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (gid >= per * g.batch) return;
  const int64_t z = gid / per;
  const int64_t idx = gid - z * per;
  const int64_t row = idx % g.m;
  const int64_t col = idx / g.m;
  const T* a = g.a + z * g.stride_a + (TA ? row * g.lda : row);
  const T* b = g.b + z * g.stride_b + (TB ? col : col * g.ldb);
  const int64_t sa = TA ? 1 : g.lda;
  const int64_t sb = TB ? g.ldb : 1;
  T acc = T(0);
  for (int64_t k = 0; k < g.k; ++k) acc = fma_rn(a[k * sa], b[k * sb], acc);
  T* p = g.c + z * g.stride_c + row + col * g.ldc;
  T out = g.alpha * acc;
  if (!g.beta_zero) {
    const T bc = g.beta * (*p);
    out = out + bc;
  }
  *p = out;
}

// ------------------------------------- register-block kernel for batches of very small items
// Strided batches of items of at most 16 x 16: one lane owns an IR x IC block of one item's C (the lanes of an
// item are neighbours, so they share its cache lines) and walks k with fused multiply-adds -- per element the
// same chain as the naive kernel, with IR + IC loads feeding IR * IC multiply-adds instead of two feeding one
// (fp32 16^3 x 65536: 162 -> see the batch table of the design notes).
template <typename T, bool TA, bool TB, int IR, int IC>
__global__ void __launch_bounds__(256) gemm_block_kernel(const GemmArgs<T> g) {
  const int64_t tiles_r = (g.m + IR - 1) / IR;
  const int64_t tiles_c = (g.n + IC - 1) / IC;
  const int64_t per = tiles_r * tiles_c;
  const int64_t gid = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (gid >= per * g.batch) return;
  const int64_t z = gid / per;
  const int64_t t = gid - z * per;
  const int64_t r0 = (t % tiles_r) * IR;
  const int64_t c0 = (t / tiles_r) * IC;

  T acc[IR][IC];
#pragma unroll
  for (int i = 0; i < IR; ++i)
#pragma unroll
    for (int j = 0; j < IC; ++j) acc[i][j] = T(0);

  // clamped rows and columns feed unstored elements only
  const T* pa[IR];
  const T* pb[IC];
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
    T av[IR], bv[IC];
#pragma unroll
    for (int i = 0; i < IR; ++i) {
      av[i] = *pa[i];
      pa[i] += step_a;
    }
#pragma unroll
    for (int j = 0; j < IC; ++j) {
      bv[j] = *pb[j];
      pb[j] += step_b;
    }
#pragma unroll
    for (int i = 0; i < IR; ++i)
#pragma unroll
      for (int j = 0; j < IC; ++j) acc[i][j] = fma_rn(av[i], bv[j], acc[i][j]);
  }

#pragma unroll
  for (int j = 0; j < IC; ++j) {
#pragma unroll
    for (int i = 0; i < IR; ++i) {
      if (r0 + i < g.m && c0 + j < g.n) {
        T* p = g.c + z * g.stride_c + (r0 + i) + (c0 + j) * g.ldc;
        T out = g.alpha * acc[i][j];
        if (!g.beta_zero) {
          const T bc = g.beta * (*p);
          out = out + bc;
        }
        *p = out;
      }
    }
  }
}

// ------------------------------------------------- interleaved batched GEMM
// Element (r, c) of batch item b lives at (r + c*ld)*batch + b, so the batch
// index is the contiguous one (test/unittest/blas3/blas3_gemm_common.hpp:54-69).
// Consecutive lanes take consecutive batch items -- every access is coalesced
// with no staging -- and each thread keeps an IR x IC register block of its
// item's C, walking k in order.  Counterpart of the reference's
// Gemm<..., interleaved> (src/operations/blas3/gemm_interleaved.hpp:104-537),
// which likewise vectorises over the batch and uses no local memory.
template <typename T, bool TA, bool TB, int IR, int IC>
__global__ void __launch_bounds__(256) gemm_interleaved_kernel(const GemmArgs<T> g) {
  const int64_t nb = g.batch;
  const int64_t tiles_r = (g.m + IR - 1) / IR;
  const int64_t tiles_c = (g.n + IC - 1) / IC;
  const int64_t gid = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (gid >= nb * tiles_r * tiles_c) return;
  const int64_t b = gid % nb;
  const int64_t t = gid / nb;
  const int64_t r0 = (t % tiles_r) * IR;
  const int64_t c0 = (t / tiles_r) * IC;

  T acc[IR][IC];
#pragma unroll
  for (int i = 0; i < IR; ++i)
#pragma unroll
    for (int j = 0; j < IC; ++j) acc[i][j] = T(0);

  // Row / column pointers are set up once (clamped rows and columns feed unstored
  // elements only) and advanced by a uniform stride per k.
  const T* pa[IR];
  const T* pb[IC];
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
    T av[IR], bv[IC];
#pragma unroll
    for (int i = 0; i < IR; ++i) {
      av[i] = *pa[i];
      pa[i] += step_a;
    }
#pragma unroll
    for (int j = 0; j < IC; ++j) {
      bv[j] = *pb[j];
      pb[j] += step_b;
    }
#pragma unroll
    for (int i = 0; i < IR; ++i)
#pragma unroll
      for (int j = 0; j < IC; ++j) acc[i][j] = fma_rn(av[i], bv[j], acc[i][j]);
  }

#pragma unroll
  for (int j = 0; j < IC; ++j) {
#pragma unroll
    for (int i = 0; i < IR; ++i) {
      if (r0 + i < g.m && c0 + j < g.n) {
        T* p = g.c + ((r0 + i) + (c0 + j) * g.ldc) * nb + b;
        T out = g.alpha * acc[i][j];
        if (!g.beta_zero) {
          const T bc = g.beta * (*p);
          out = out + bc;
        }
        *p = out;
      }
    }
  }
}

// ----------------------------------------- interleaved <-> strided repacking
// Large interleaved problems are repacked into one column-major matrix per batch
// item, run through the strided kernels (matrix instructions need the matrix
// indices on the lanes, the interleaved layout puts the batch index there) and
// repacked on the way out.  A 32 x 32 LDS tile turns the batch-contiguous side
// into the row-contiguous side, so both the loads and the stores are coalesced.
//   interleaved: element (r, c) of item b at (r + c*ld)*batch + b
//   packed:      element (r, c) of item b at b*rows*cols + r + c*rows
// TO_PACKED = false writes the rows x cols window back into the interleaved
// buffer (padding rows between ld and rows are never touched).
template <typename T, bool TO_PACKED>
__global__ void __launch_bounds__(256)
    repack_interleaved_kernel(T* __restrict__ inter, T* __restrict__ packed, int64_t rows, int64_t cols, int64_t ld,
                              int64_t batch) {
  __shared__ T tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8 threads
  const int64_t tiles_r = (rows + 31) / 32, tiles_b = (batch + 31) / 32;
  int64_t t = blockIdx.x;
  const int64_t tb = t % tiles_b;
  t /= tiles_b;
  const int64_t tr = t % tiles_r;
  const int64_t c = t / tiles_r;
  const int64_t r0 = tr * 32, b0 = tb * 32;
  if (TO_PACKED) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t r = r0 + ty + 8 * i, b = b0 + tx;  // lanes along the batch
      if (r < rows && b < batch) tile[ty + 8 * i][tx] = inter[(r + c * ld) * batch + b];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t r = r0 + tx, b = b0 + ty + 8 * i;  // lanes along the rows
      if (r < rows && b < batch) packed[b * rows * cols + r + c * rows] = tile[tx][ty + 8 * i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t r = r0 + tx, b = b0 + ty + 8 * i;
      if (r < rows && b < batch) tile[tx][ty + 8 * i] = packed[b * rows * cols + r + c * rows];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t r = r0 + ty + 8 * i, b = b0 + tx;
      if (r < rows && b < batch) inter[(r + c * ld) * batch + b] = tile[ty + 8 * i][tx];
    }
  }
}

// The same repacking with a padded packed image: element (r, c) of item b at
// b*ld_packed*cols + r + c*ld_packed (ld_packed >= rows), so that every packed column starts
// on a 16-byte boundary whatever `rows` is.  Rows between `rows` and ld_packed are never
// written and never read as operands.
template <typename T, bool TO_PACKED>
__global__ void __launch_bounds__(256)
    repack_interleaved_ld_kernel(T* __restrict__ inter, T* __restrict__ packed, int64_t rows, int64_t cols, int64_t ld,
                                 int64_t ld_packed, int64_t batch) {
  __shared__ T tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8 threads
  const int64_t tiles_r = (rows + 31) / 32, tiles_b = (batch + 31) / 32;
  int64_t t = blockIdx.x;
  const int64_t tb = t % tiles_b;
  t /= tiles_b;
  const int64_t tr = t % tiles_r;
  const int64_t c = t / tiles_r;
  const int64_t r0 = tr * 32, b0 = tb * 32;
  const int64_t item = ld_packed * cols;
  if (TO_PACKED) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t r = r0 + ty + 8 * i, b = b0 + tx;  // lanes along the batch
      if (r < rows && b < batch) tile[ty + 8 * i][tx] = inter[(r + c * ld) * batch + b];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t r = r0 + tx, b = b0 + ty + 8 * i;  // lanes along the rows
      if (r < rows && b < batch) packed[b * item + r + c * ld_packed] = tile[tx][ty + 8 * i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t r = r0 + tx, b = b0 + ty + 8 * i;
      if (r < rows && b < batch) tile[tx][ty + 8 * i] = packed[b * item + r + c * ld_packed];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t r = r0 + ty + 8 * i, b = b0 + tx;
      if (r < rows && b < batch) inter[(r + c * ld) * batch + b] = tile[ty + 8 * i][tx];
    }
  }
}

// ---------------------------------------------------------------- C = beta*C
// alpha == 0 path (src/interface/gemm_interface.hpp:112-139).  assign_zero is
// the whole-vector _scal with a zero factor, which stores zeros
// (src/interface/blas1_interface.hpp:444-448); otherwise every element of the
// rows x cols window is multiplied, so NaN * 0 stays NaN (:498-506).
template <typename T>
__global__ void __launch_bounds__(256)
    scale_matrix_kernel(T* c, int64_t rows, int64_t cols, int64_t ld, int64_t stride,
                        int64_t batch, T beta, int assign_zero) {
  const int64_t per = rows * cols;
  const int64_t gid = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (gid >= per * batch) return;
  const int64_t z = gid / per;
  const int64_t idx = gid - z * per;
  T* p = c + z * stride + (idx % rows) + (idx / rows) * ld;
  *p = assign_zero ? T(0) : beta * (*p);
}

}  // namespace sbamd
