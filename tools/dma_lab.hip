// Lab (timing + check): fp32 64x64x32 tile whose operand tiles go global -> LDS directly
// (global_load_lds_dwordx4, three LDS stages, counted waits), NT layout only (both LDS images are
// the linear [k][mn] image, so no swizzle is needed), against the library's register-staged
// kernel on the same problem.  One workgroup per compute unit at 1024^3: does taking the staging
// out of the instruction stream close the gap there?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Isycl_blas_amd/csrc tools/dma_lab.hip -o tools/bin/dma_lab
#include <cstdio>
#include <vector>

#include "gemm_mfma.hpp"

using namespace sbamd;

#define CHECK(x)                                                              \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 1;                                                               \
    }                                                                         \
  } while (0)

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int BK, int STAGES>
__global__ void __launch_bounds__(256, 2) dma_nt_kernel(const GemmArgs<float> g) {
  using K = GemmTile<float, 64, 64, BK, 2, 2, false, true>;
  using M = Mfma<float>;
  constexpr int OPERAND = BK * 64;            // floats per operand image
  constexpr int STAGE = 2 * OPERAND;          // floats per stage
  constexpr int PIECES = OPERAND * 4 / 1024;  // 1 KiB load instructions per operand image
  constexpr int PER_WAVE = PIECES / 4;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
  SBAMD_LDS float* smem = (SBAMD_LDS float*)smem_raw;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  int tm, tn;
  tile_coords(logical, g.tiles_m, g.tiles_n, tm, tn);
  const int64_t m0 = int64_t(tm) * 64, n0 = int64_t(tn) * 64;

  // lane i of piece j covers k row 4 j + i / 16, columns (i % 16) * 4 .. + 3
  const float* pa[PER_WAVE];
  const float* pb[PER_WAVE];
#pragma unroll
  for (int u = 0; u < PER_WAVE; ++u) {
    const int j = wave + 4 * u;
    const int krow = 4 * j + (lane >> 4), mn = (lane & 15) * 4;
    pa[u] = g.a + (m0 + mn) + int64_t(krow) * g.lda;
    pb[u] = g.b + (n0 + mn) + int64_t(krow) * g.ldb;
  }
  auto issue = [&](int stage) {
    SBAMD_LDS float* st = smem + stage * STAGE;
#pragma unroll
    for (int u = 0; u < PER_WAVE; ++u) {
      const int j = wave + 4 * u;
      __builtin_amdgcn_global_load_lds((gptr_t)pa[u], (lptr_t)(st + j * 256), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)pb[u], (lptr_t)(st + OPERAND + j * 256), 16, 0, 0);
      pa[u] += int64_t(BK) * g.lda;
      pb[u] += int64_t(BK) * g.ldb;
    }
  };

  GemmBody<float, 64, 64, BK, 2, 2, false, true> body;
#pragma unroll
  for (int r = 0; r < 16; ++r) body.acc[0][0][r] = 0.0f;
  const int a_mn = wm * 32 + (lane & 31), b_mn = wn * 32 + (lane & 31);
  const int nk = int(g.k / BK);

  // tiles 0 .. STAGES-2 in flight
#pragma unroll
  for (int t = 0; t < STAGES - 1; ++t)
    if (t < nk) issue(t);
  if (nk >= STAGES - 1) {
    if (STAGES == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_WAVE) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PER_WAVE) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();

  int cur = 0, fill = STAGES - 1;
  for (int t = 0; t < nk; ++t) {
    const bool more = t + STAGES - 1 < nk;
    if (more) issue(fill);
    const SBAMD_LDS float* st = smem + cur * STAGE;
    body.template compute<0>(st, st + OPERAND, a_mn, b_mn, lane);
    // tile t + 1 (this wave's pieces) has landed; the newest STAGES - 2 tiles may still be in flight
    if (more) {
      if (STAGES == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_WAVE) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PER_WAVE) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    cur = cur + 1 == STAGES ? 0 : cur + 1;
    fill = fill + 1 == STAGES ? 0 : fill + 1;
  }

  const int64_t mm = m0 + a_mn;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t nn = n0 + wn * 32 + M::acc_i(r, lane);
    g.c[mm + nn * g.ldc] = g.alpha * body.acc[0][0][r];
  }
}

template <typename F>
int time_it(const char* name, int n, int reps, F&& launch) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  // warm up as long as the timed part: the clock keeps rising over the first milliseconds of load, which
  // would otherwise favour whichever variant is timed second
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  std::printf("DMA_LAB %-44s %d^3: %7.1f us  %6.1f TFLOP/s\n", name, n, ms * 1e3, 2.0 * n * n * double(n) / ms / 1e9);
  return 0;
}

template <int BK, int STAGES>
int run_dma(const char* name, const GemmArgs<float>& g, int n, int reps) {
  auto kernel = dma_nt_kernel<BK, STAGES>;
  const int lds = STAGES * 2 * BK * 64 * 4;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  return time_it(name, n, reps, [&] { hipLaunchKernelGGL(kernel, dim3(g.tiles_m * g.tiles_n), dim3(256), lds, 0, g); });
}

template <int RM, int RN, int RK, int RWM, int RWN, int ROCC, int BM, int BN, int BK, int WM, int WN, int OCC, int STAGES>
int compare_tiles(const char* name, int n) {
  float *a, *b, *c, *c_ref;
  CHECK(hipMalloc(&a, sizeof(float) * n * n));
  CHECK(hipMalloc(&b, sizeof(float) * n * n));
  CHECK(hipMalloc(&c, sizeof(float) * n * n));
  CHECK(hipMalloc(&c_ref, sizeof(float) * n * n));
  std::vector<float> host(size_t(n) * n);
  for (size_t i = 0; i < host.size(); ++i) host[i] = float((i * 2654435761u % 7000) / 1000.0 - 2.0);
  CHECK(hipMemcpy(a, host.data(), sizeof(float) * n * n, hipMemcpyHostToDevice));
  for (size_t i = 0; i < host.size(); ++i) host[i] = float((i * 40503u % 7000) / 1000.0 - 2.0);
  CHECK(hipMemcpy(b, host.data(), sizeof(float) * n * n, hipMemcpyHostToDevice));
  GemmArgs<float> g{};
  g.a = a; g.b = b; g.c = c_ref;
  g.m = g.n = g.k = n;
  g.lda = g.ldb = g.ldc = n;
  g.batch = 1; g.split_k = 1; g.k_chunk = n;
  g.vec_a = g.vec_b = 1;
  g.alpha = 1.0f; g.beta = 0.0f; g.beta_zero = 1;
  const int reps = n >= 8192 ? 5 : n >= 4096 ? 20 : 100;
  std::printf("DMA_LAB -- %s\n", name);
  {
    using K = GemmTile<float, RM, RN, RK, RWM, RWN, false, true>;
    auto kernel = gemm_mfma_kernel<float, RM, RN, RK, RWM, RWN, false, true, ROCC>;
    g.tiles_m = n / RM; g.tiles_n = n / RN;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(K::LDS_BYTES)));
    if (time_it("register-staged", n, reps,
                [&] { hipLaunchKernelGGL(kernel, dim3(g.tiles_m * g.tiles_n), dim3(K::NT), K::LDS_BYTES, 0, g); }))
      return 1;
  }
  {
    auto kernel = gemm_mfma_dma_kernel<BM, BN, BK, WM, WN, false, true, OCC, STAGES>;
    const int lds = STAGES * (BM + BN) * BK * 4;
    g.c = c;
    g.tiles_m = n / BM; g.tiles_n = n / BN;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    if (time_it("direct-to-LDS", n, reps,
                [&] { hipLaunchKernelGGL(kernel, dim3(g.tiles_m * g.tiles_n), dim3(WM * WN * 64), lds, 0, g); }))
      return 1;
  }
  std::vector<float> got(size_t(n) * n), want(size_t(n) * n);
  CHECK(hipMemcpy(got.data(), c, sizeof(float) * n * n, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(want.data(), c_ref, sizeof(float) * n * n, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < got.size(); ++i) bad += got[i] != want[i];
  std::printf("DMA_LAB    %zu of %zu elements differ\n", bad, got.size());
  hipFree(a); hipFree(b); hipFree(c); hipFree(c_ref);
  return 0;
}

int main() {
  for (int n : {1024, 2048}) {
    float *a, *b, *c, *c_ref;
    CHECK(hipMalloc(&a, sizeof(float) * n * n));
    CHECK(hipMalloc(&b, sizeof(float) * n * n));
    CHECK(hipMalloc(&c, sizeof(float) * n * n));
    CHECK(hipMalloc(&c_ref, sizeof(float) * n * n));
    std::vector<float> host(size_t(n) * n);
    for (size_t i = 0; i < host.size(); ++i) host[i] = float((i * 2654435761u % 7000) / 1000.0 - 2.0);
    CHECK(hipMemcpy(a, host.data(), sizeof(float) * n * n, hipMemcpyHostToDevice));
    for (size_t i = 0; i < host.size(); ++i) host[i] = float((i * 40503u % 7000) / 1000.0 - 2.0);
    CHECK(hipMemcpy(b, host.data(), sizeof(float) * n * n, hipMemcpyHostToDevice));
    GemmArgs<float> g{};
    g.a = a; g.b = b; g.c = c_ref;
    g.m = g.n = g.k = n;
    g.lda = g.ldb = g.ldc = n;
    g.batch = 1; g.split_k = 1; g.k_chunk = n;
    g.tiles_m = n / 64; g.tiles_n = n / 64;
    g.vec_a = g.vec_b = 1;
    g.alpha = 1.0f; g.beta = 0.0f; g.beta_zero = 1;
    {
      using K = GemmTile<float, 64, 64, 32, 2, 2, false, true>;
      auto kernel = gemm_mfma_kernel<float, 64, 64, 32, 2, 2, false, true, 1>;
      CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(K::LDS_BYTES)));
      if (time_it("library kernel 64x64x32 (register-staged) NT", n, 200,
                  [&] { hipLaunchKernelGGL(kernel, dim3(g.tiles_m * g.tiles_n), dim3(256), K::LDS_BYTES, 0, g); }))
        return 1;
    }
    g.c = c;
    if (run_dma<32, 3>("direct-to-LDS 64x64x32, 3 stages", g, n, 200)) return 1;
    std::vector<float> got(size_t(n) * n), want(size_t(n) * n);
    CHECK(hipMemcpy(got.data(), c, sizeof(float) * n * n, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(want.data(), c_ref, sizeof(float) * n * n, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < got.size(); ++i) bad += got[i] != want[i];
    std::printf("DMA_LAB %d^3: %zu of %zu elements differ from the library kernel\n", n, bad, got.size());
    if (run_dma<32, 4>("direct-to-LDS 64x64x32, 4 stages", g, n, 200)) return 1;
    if (run_dma<64, 3>("direct-to-LDS 64x64x64, 3 stages", g, n, 200)) return 1;
    CHECK(hipMemcpy(got.data(), c, sizeof(float) * n * n, hipMemcpyDeviceToHost));
    bad = 0;
    for (size_t i = 0; i < got.size(); ++i) bad += got[i] != want[i];
    std::printf("DMA_LAB %d^3 (last variant): %zu of %zu elements differ from the library kernel\n", n, bad, got.size());
    hipFree(a); hipFree(b); hipFree(c); hipFree(c_ref);
  }
  // the library's direct-to-LDS kernel against the register-staged kernel
  for (int round = 0; round < 2; ++round) {
    if (compare_tiles<64, 64, 32, 2, 2, 1, 64, 64, 32, 2, 2, 3, 3>("64x64x32 reg  vs 64x64x32 dma 3 stages", 1024)) return 1;
    if (compare_tiles<64, 64, 32, 2, 2, 1, 64, 64, 32, 2, 2, 3, 3>("64x64x32 reg  vs 64x64x32 dma 3 stages", 2048)) return 1;
  }
  if (compare_tiles<128, 128, 32, 4, 2, 2, 128, 128, 16, 2, 2, 2, 3>("128x128x32 reg (8 waves) vs 128x128x16 dma, 4 waves of 64x64", 2048)) return 1;
  if (compare_tiles<128, 128, 32, 4, 2, 2, 128, 128, 32, 2, 2, 1, 3>("128x128x32 reg (8 waves) vs 128x128x32 dma, 4 waves of 64x64", 2048)) return 1;
  if (compare_tiles<128, 128, 32, 4, 2, 2, 128, 128, 16, 2, 2, 2, 3>("128x128x32 reg (8 waves) vs 128x128x16 dma, 4 waves of 64x64", 4096)) return 1;
  if (compare_tiles<128, 128, 32, 4, 2, 2, 128, 128, 32, 2, 2, 1, 3>("128x128x32 reg (8 waves) vs 128x128x32 dma, 4 waves of 64x64", 4096)) return 1;
  if (compare_tiles<128, 128, 32, 4, 2, 2, 128, 128, 16, 2, 4, 2, 3>("128x128x32 reg (8 waves) vs 128x128x16 dma, 8 waves of 64x32", 2048)) return 1;
  if (compare_tiles<128, 128, 32, 4, 2, 2, 128, 128, 32, 4, 2, 2, 3>("128x128x32 reg  vs 128x128x32 dma 3 stages", 2048)) return 1;
  if (compare_tiles<128, 128, 32, 4, 2, 2, 128, 128, 16, 4, 2, 2, 3>("128x128x32 reg  vs 128x128x16 dma 3 stages", 2048)) return 1;
  if (compare_tiles<128, 128, 32, 4, 2, 2, 128, 128, 16, 4, 2, 2, 4>("128x128x32 reg  vs 128x128x16 dma 4 stages", 4096)) return 1;
  if (compare_tiles<128, 128, 32, 4, 2, 2, 128, 128, 32, 4, 2, 2, 3>("128x128x32 reg  vs 128x128x32 dma 3 stages", 4096)) return 1;
  if (compare_tiles<256, 256, 16, 4, 2, 2, 256, 256, 16, 4, 2, 2, 3>("256x256x16 reg  vs 256x256x16 dma 3 stages", 8192)) return 1;
  if (compare_tiles<256, 256, 16, 4, 2, 2, 256, 256, 16, 4, 2, 2, 4>("256x256x16 reg  vs 256x256x16 dma 4 stages", 8192)) return 1;
  if (compare_tiles<256, 256, 16, 4, 2, 2, 256, 256, 32, 4, 2, 2, 2>("256x256x16 reg  vs 256x256x32 dma 2 stages", 8192)) return 1;
  return 0;
}
