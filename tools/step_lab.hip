// Lab for the K-step schedule of the fp32 / fp64 tile kernels (SBAMD_LAB_VARIANT in gemm_mfma.hpp): one binary per
// variant, each times the headline tile at 8192^3 (and optionally the 128 x 128 tile at 4096^3) back to back and
// prints a checksum of C so that variants can be compared bit for bit.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DSBAMD_LAB_VARIANT=<v> \
//         -Isycl_blas_amd/csrc -Iinclude tools/step_lab.hip -o tools/bin/step_lab_<v>
#include <cstdio>
#include <cstdint>
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

template <typename T, int BM, int BN, int BK, int WM, int WN, int OCC, bool TA, bool TB, typename MS = Mfma<T>>
int run(const char* name, int n, int reps, int group_m = 0, int kmul = 1) {
  using K = GemmTile<T, BM, BN, BK, WM, WN, TA, TB, MS>;
  T *a, *b, *c;
  CHECK(hipMalloc(&a, sizeof(T) * n * n * kmul));
  CHECK(hipMalloc(&b, sizeof(T) * n * n * kmul));
  CHECK(hipMalloc(&c, sizeof(T) * n * n));
  std::vector<T> host(size_t(n) * n);
  for (size_t i = 0; i < host.size(); ++i) host[i] = T((i * 2654435761u % 7000) / 1000.0 - 2.0);
  for (int q = 0; q < kmul; ++q) CHECK(hipMemcpy(a + size_t(q) * n * n, host.data(), sizeof(T) * n * n, hipMemcpyHostToDevice));
  for (size_t i = 0; i < host.size(); ++i) host[i] = T((i * 40503u % 7000) / 1000.0 - 2.0);
  for (int q = 0; q < kmul; ++q) CHECK(hipMemcpy(b + size_t(q) * n * n, host.data(), sizeof(T) * n * n, hipMemcpyHostToDevice));
  GemmArgs<T> g{};
  g.a = a; g.b = b; g.c = c;
  g.m = g.n = n;
  g.k = int64_t(n) * kmul;
  g.lda = TA ? g.k : n;
  g.ldc = n;
  g.ldb = TB ? n : g.k;
  g.batch = 1; g.split_k = 1; g.k_chunk = g.k;
  g.tiles_m = n / BM; g.tiles_n = n / BN;
  g.vec_a = g.vec_b = 1;
  g.alpha = T(1); g.beta = T(0); g.beta_zero = 1;
  g.group_m = group_m;
  auto kernel = gemm_mfma_kernel<T, BM, BN, BK, WM, WN, TA, TB, OCC, MS>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            int(K::LDS_BYTES)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  double best = 1e30, sum = 0;
  const int rounds = 3;
  for (int r = 0; r < rounds; ++r) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kernel, dim3(g.tiles_m * g.tiles_n), dim3(K::NT), K::LDS_BYTES, 0, g);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kernel, dim3(g.tiles_m * g.tiles_n), dim3(K::NT), K::LDS_BYTES, 0, g);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    best = ms < best ? ms : best;
    sum += ms;
  }
  CHECK(hipMemcpy(host.data(), c, sizeof(T) * n * n, hipMemcpyDeviceToHost));
  uint64_t h = 1469598103934665603ull;
  const unsigned char* p = reinterpret_cast<const unsigned char*>(host.data());
  for (size_t i = 0; i < host.size() * sizeof(T); i += 7) h = (h ^ p[i]) * 1099511628211ull;
  const double fl = 2.0 * n * n * double(n) * kmul;
  std::printf("variant=%d %-44s %d^3: best %.4f ms %.1f TFLOP/s  mean %.4f ms %.1f TFLOP/s  checksum %016llx\n",
              SBAMD_LAB_VARIANT, name, n, best, fl / best / 1e9, sum / rounds, fl / (sum / rounds) / 1e9,
              (unsigned long long)h);
  (void)hipFree(a); (void)hipFree(b); (void)hipFree(c);
  return 0;
}

int main(int argc, char** argv) {
  const char mode = argc > 1 ? argv[1][0] : 'h';
  if (mode == 'h') {
    if (run<float, 256, 256, 16, 2, 4, 2, false, false>("f32 256x256x16 NN (2x4 waves)", 8192, 10)) return 1;
    if (run<float, 256, 256, 16, 4, 2, 2, false, true>("f32 256x256x16 NT (4x2 waves)", 8192, 10)) return 1;
    if (run<double, 256, 128, 16, 4, 2, 1, false, false>("f64 256x128x16 NN (4x2 waves)", 8192, 5, 4)) return 1;
  } else if (mode == 'k') {  // K four times as long: what a tile's prologue and epilogue cost
    if (run<float, 256, 256, 16, 2, 4, 2, false, false>("f32 256x256x16 NN (2x4 waves) K x4", 8192, 3, 0, 4)) return 1;
  } else if (mode == 't') {  // kernels whose compiled K loop has the global loads sunk towards its end
    if (run<float, 256, 256, 16, 4, 2, 2, true, false>("f32 256x256x16 TN (4x2 waves)", 8192, 10)) return 1;
    if (run<float, 256, 256, 16, 2, 4, 2, true, true>("f32 256x256x16 TT (2x4 waves)", 8192, 10)) return 1;
    if (run<float, 256, 128, 32, 4, 2, 1, true, false>("f32 256x128x32 TN (4x2 waves)", 8192, 10)) return 1;
    if (run<float, 256, 128, 32, 4, 2, 1, true, true>("f32 256x128x32 TT (4x2 waves)", 8192, 10)) return 1;
  } else if (mode == 's') {  // 64 x 64 tile on fewer, bigger wavefront tiles (the four-wavefront form reads 128 B of LDS per clock when a unit is full)
    for (int n : {1536, 1024, 3072}) {
      const int reps = n == 3072 ? 20 : 100;
      if (run<float, 64, 64, 32, 2, 2, 2, false, false>("f32 64x64x32 NN (4 waves of 32x32)", n, reps)) return 1;
      if (run<float, 64, 64, 32, 1, 2, 2, false, false>("f32 64x64x32 NN (2 waves of 64x32)", n, reps)) return 1;
      if (run<float, 64, 64, 32, 2, 1, 2, false, false>("f32 64x64x32 NN (2 waves of 32x64)", n, reps)) return 1;
      if (run<float, 64, 64, 32, 1, 1, 1, false, false>("f32 64x64x32 NN (1 wave of 64x64)", n, reps)) return 1;
      if (run<float, 128, 64, 32, 2, 2, 2, false, false>("f32 128x64x32 NN (4 waves of 64x32)", n, reps)) return 1;
      if (run<float, 64, 128, 32, 2, 2, 2, false, false>("f32 64x128x32 NN (4 waves of 32x64)", n, reps)) return 1;
    }
  } else if (mode == 'm') {  // 96 x 96 tile on the 16x16x4 instruction against the 64 x 64 and 128 x 128 tiles
    for (int n : {1536, 3072}) {
      const int reps = n == 3072 ? 20 : 100;
      if (run<float, 64, 64, 32, 2, 2, 2, false, false>("f32 64x64x32 NN (4 waves of 32x32)", n, reps)) return 1;
      if (run<float, 128, 128, 32, 4, 2, 2, false, false>("f32 128x128x32 NN (8 waves of 32x64)", n, reps)) return 1;
      if (run<float, 96, 96, 32, 2, 2, 1, false, false, MfmaF32Small>("f32 96x96x32 NN (4 waves of 48x48, 16x16x4) occ1", n, reps)) return 1;
      if (run<float, 96, 96, 32, 2, 2, 2, false, false, MfmaF32Small>("f32 96x96x32 NN (4 waves of 48x48, 16x16x4) occ2", n, reps)) return 1;
      if (run<float, 96, 96, 32, 2, 2, 2, true, false, MfmaF32Small>("f32 96x96x32 TN (16x16x4) occ2", n, reps)) return 1;
      if (run<float, 96, 96, 32, 2, 2, 2, false, true, MfmaF32Small>("f32 96x96x32 NT (16x16x4) occ2", n, reps)) return 1;
    }
  } else if (mode == 'n') {  // the 16x16x4 instruction on the other tile sizes (more independent accumulators per wavefront)
    if (run<float, 64, 64, 32, 2, 2, 2, false, false>("f32 64x64x32 NN (32x32x2, far pipeline)", 1024, 200)) return 1;
    if (run<float, 64, 64, 32, 2, 2, 2, false, false, MfmaF32Small>("f32 64x64x32 NN (16x16x4, 4 waves of 32x32)", 1024, 200)) return 1;
    if (run<float, 64, 64, 64, 2, 2, 2, false, false, MfmaF32Small>("f32 64x64x64 NN (16x16x4, 4 waves of 32x32)", 1024, 200)) return 1;
    if (run<float, 64, 64, 32, 2, 2, 2, false, true, MfmaF32Small>("f32 64x64x32 NT (16x16x4, 4 waves of 32x32)", 1024, 200)) return 1;
    if (run<float, 128, 128, 32, 4, 2, 2, false, false>("f32 128x128x32 NN (32x32x2, 8 waves of 32x64)", 2048, 100)) return 1;
    if (run<float, 128, 128, 32, 2, 2, 2, false, false, MfmaF32Small>("f32 128x128x32 NN (16x16x4, 4 waves of 64x64)", 2048, 100)) return 1;
    if (run<float, 128, 128, 32, 4, 2, 2, false, false, MfmaF32Small>("f32 128x128x32 NN (16x16x4, 8 waves of 32x64)", 2048, 100)) return 1;
    if (run<float, 128, 128, 32, 4, 2, 2, false, false>("f32 128x128x32 NN (32x32x2, 8 waves of 32x64)", 4096, 20)) return 1;
    if (run<float, 128, 128, 32, 2, 2, 2, false, false, MfmaF32Small>("f32 128x128x32 NN (16x16x4, 4 waves of 64x64)", 4096, 20)) return 1;
    if (run<float, 128, 128, 32, 4, 2, 2, false, false, MfmaF32Small>("f32 128x128x32 NN (16x16x4, 8 waves of 32x64)", 4096, 20)) return 1;
    if (run<float, 96, 96, 32, 2, 2, 2, false, false, MfmaF32Small>("f32 96x96x32 NN (16x16x4)", 768, 200)) return 1;
    if (run<float, 64, 64, 32, 2, 2, 2, false, false>("f32 64x64x32 NN (32x32x2, far pipeline)", 768, 200)) return 1;
    if (run<float, 64, 64, 32, 2, 2, 2, false, false, MfmaF32Small>("f32 64x64x32 NN (16x16x4)", 768, 200)) return 1;
  } else if (mode == 'b') {  // 192 x 192 on the 16x16x4 instruction: 256 tiles at 3072^3
    if (run<float, 96, 96, 32, 2, 2, 2, false, false, MfmaF32Small>("f32 96x96x32 NN (16x16x4)", 3072, 20)) return 1;
    if (run<float, 192, 192, 32, 2, 4, 2, false, false, MfmaF32Small>("f32 192x192x32 NN (16x16x4, 8 waves of 96x48)", 3072, 20)) return 1;
    if (run<float, 192, 192, 32, 4, 2, 2, false, false, MfmaF32Small>("f32 192x192x32 NN (16x16x4, 8 waves of 48x96)", 3072, 20)) return 1;
    if (run<float, 192, 192, 32, 2, 2, 1, false, false, MfmaF32Small>("f32 192x192x32 NN (16x16x4, 4 waves of 96x96)", 3072, 20)) return 1;
    if (run<float, 192, 192, 32, 2, 4, 2, false, true, MfmaF32Small>("f32 192x192x32 NT (16x16x4, 8 waves of 96x48)", 3072, 20)) return 1;
    if (run<float, 192, 192, 32, 2, 4, 2, true, false, MfmaF32Small>("f32 192x192x32 TN (16x16x4, 8 waves of 96x48)", 3072, 20)) return 1;
    if (run<float, 192, 96, 32, 2, 2, 2, false, false, MfmaF32Small>("f32 192x96x32 NN (16x16x4, 4 waves of 96x48)", 3072, 20)) return 1;
  } else if (mode == 'd') {  // the same tile edges for double (its instruction is 16x16x4 already)
    for (int n : {1536, 3072}) {
      const int reps = n == 3072 ? 10 : 50;
      if (run<double, 64, 64, 16, 2, 2, 2, false, false>("f64 64x64x16 NN", n, reps)) return 1;
      if (run<double, 128, 128, 16, 2, 2, 2, false, false>("f64 128x128x16 NN", n, reps)) return 1;
      if (run<double, 96, 96, 16, 2, 2, 2, false, false>("f64 96x96x16 NN (4 waves of 48x48)", n, reps)) return 1;
      if (run<double, 192, 192, 16, 2, 4, 1, false, false>("f64 192x192x16 NN (8 waves of 96x48)", n, reps)) return 1;
    }
  } else if (mode == 'v') {  // deeper K steps on the 64 x 64 tile at 1024^3 (one tile per compute unit).  Read with care: run
    // back to back here the step of 128 read 22.3 us against 24.2, but behind the library's entry point, each form
    // timed after 15 ms of its own launches (benchmark/tune_gemm), every form of the 64 x 64 tile reads 22.1-22.3 us
    // at 1024^3, and the step of 128 is 6 % slower at 1000^3 and 4 % at 4096 x 512 x 4096 -- not taken
    if (run<float, 64, 64, 32, 2, 2, 2, false, false>("f32 64x64x32 NN", 1024, 200)) return 1;
    if (run<float, 64, 64, 64, 2, 2, 2, false, false>("f32 64x64x64 NN", 1024, 200)) return 1;
    if (run<float, 64, 64, 128, 2, 2, 1, false, false>("f32 64x64x128 NN", 1024, 200)) return 1;
    if (run<float, 64, 64, 128, 2, 2, 1, false, false, MfmaF32Small>("f32 64x64x128 NN (16x16x4)", 1024, 200)) return 1;
    if (run<float, 64, 64, 128, 2, 2, 1, false, true>("f32 64x64x128 NT", 1024, 200)) return 1;
    if (run<float, 64, 64, 128, 2, 2, 1, true, false>("f32 64x64x128 TN", 1024, 200)) return 1;
  } else if (mode == 'w') {  // 64 x 64 tile on eight wavefronts of the 16x16x4 instruction (two per SIMD, no K cut), one tile per unit:
    // 1024^3 22.7 us (16 x 32 wave tiles, K step 64) to 24.9 us against 22.0-22.1 for the four wavefronts of 32 x 32 -- not taken
    if (run<float, 64, 64, 64, 2, 2, 2, false, false>("f32 64x64x64 NN (4 waves of 32x32, 32x32x2)", 1024, 400)) return 1;
    if (run<float, 64, 64, 32, 4, 2, 2, false, false, MfmaF32Small>("f32 64x64x32 NN (8 waves of 16x32, 16x16x4)", 1024, 400)) return 1;
    if (run<float, 64, 64, 32, 2, 4, 2, false, false, MfmaF32Small>("f32 64x64x32 NN (8 waves of 32x16, 16x16x4)", 1024, 400)) return 1;
    if (run<float, 64, 64, 64, 4, 2, 2, false, false, MfmaF32Small>("f32 64x64x64 NN (8 waves of 16x32, 16x16x4)", 1024, 400)) return 1;
    if (run<float, 64, 64, 64, 2, 4, 2, false, false, MfmaF32Small>("f32 64x64x64 NN (8 waves of 32x16, 16x16x4)", 1024, 400)) return 1;
    if (run<float, 64, 64, 64, 2, 2, 2, false, false>("f32 64x64x64 NN (4 waves of 32x32, 32x32x2) again", 1024, 400)) return 1;
  } else if (mode == 'l') {
    if (run<float, 128, 128, 32, 4, 2, 2, false, false>("f32 128x128x32 NN (8 waves of 32x64)", 4096, 20)) return 1;
    if (run<float, 128, 128, 32, 4, 2, 2, false, false>("f32 128x128x32 NN (8 waves of 32x64)", 2048, 100)) return 1;
    if (run<double, 128, 128, 16, 2, 2, 2, false, false>("f64 128x128x16 NN (4 waves of 64x64)", 4096, 10)) return 1;
    if (run<double, 128, 128, 16, 2, 2, 2, false, false>("f64 128x128x16 NN (4 waves of 64x64)", 2048, 50)) return 1;
    if (run<double, 64, 64, 16, 2, 2, 2, false, false>("f64 64x64x16 NN (4 waves of 32x32)", 1024, 200)) return 1;
    if (run<float, 128, 128, 32, 4, 2, 2, true, false>("f32 128x128x32 TN (8 waves of 32x64)", 4096, 20)) return 1;
    if (run<float, 128, 128, 32, 4, 2, 2, false, true>("f32 128x128x32 NT (8 waves of 32x64)", 4096, 20)) return 1;
  }
  return 0;
}
