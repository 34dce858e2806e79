// Timing-only ablation of the fp32 NN 64x64 tile at 1024^3 (one workgroup per compute unit) (see SBAMD_ABLATE in
// gemm_mfma.hpp).  Build one binary per mask:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DSBAMD_ABLATE=<mask> \
//         -Isycl_blas_amd/csrc tools/ablate.hip -o /tmp/ablate_<mask>
// Outputs are meaningless except for mask 0; only the times are read.
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

template <typename T, int BM, int BN, int BK, int WM, int WN, int OCC>
int run(const char* name, int n, int reps = 5) {
  using K = GemmTile<T, BM, BN, BK, WM, WN, false, false>;
  T *a, *b, *c;
  CHECK(hipMalloc(&a, sizeof(T) * n * n));
  CHECK(hipMalloc(&b, sizeof(T) * n * n));
  CHECK(hipMalloc(&c, sizeof(T) * n * n));
  std::vector<T> host(size_t(n) * n);
  for (size_t i = 0; i < host.size(); ++i) host[i] = T((i * 2654435761u % 7000) / 1000.0 - 2.0);
  CHECK(hipMemcpy(a, host.data(), sizeof(T) * n * n, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(b, host.data(), sizeof(T) * n * n, hipMemcpyHostToDevice));
  GemmArgs<T> g{};
  g.a = a; g.b = b; g.c = c;
  g.m = g.n = g.k = n;
  g.lda = g.ldb = g.ldc = n;
  g.batch = 1; g.split_k = 1; g.k_chunk = n;
  g.tiles_m = n / BM; g.tiles_n = n / BN;
  g.vec_a = g.vec_b = 1;
  g.alpha = T(1); g.beta = T(0); g.beta_zero = 1;
  auto kernel = gemm_mfma_kernel<T, BM, BN, BK, WM, WN, false, false, OCC>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            int(K::LDS_BYTES)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kernel, dim3(g.tiles_m * g.tiles_n), dim3(K::NT), K::LDS_BYTES, 0, g);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kernel, dim3(g.tiles_m * g.tiles_n), dim3(K::NT), K::LDS_BYTES, 0, g);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  std::printf("ABLATE mask=%d %s %d^3: %.3f ms  %.1f TFLOP/s\n", SBAMD_ABLATE, name, n, ms, 2.0 * n * n * double(n) / ms / 1e9);
  hipFree(a); hipFree(b); hipFree(c);
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 'h') {  // the biggest tile at the headline size
    for (int round = 0; round < 2; ++round)
      if (run<float, 256, 256, 16, 2, 4, 2>("f32 256x256x16 (2x4 waves)", 8192, 5)) return 1;
#ifdef LAB_FOUR_WAVES
      if (run<float, 256, 256, 16, 2, 2, 1>("f32 256x256x16 (2x2 waves of 128x128)", 8192, 5)) return 1;
      if (run<float, 256, 256, 8, 2, 2, 1>("f32 256x256x8 (2x2 waves of 128x128)", 8192, 5)) return 1;
#endif
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'q') {  // 96-row tiles against 128 x 128 where the tile count quantises badly
    for (int round = 0; round < 2; ++round)
      for (int n : {1536, 3072}) {
        const int reps = n == 1536 ? 100 : 20;
        if (run<float, 128, 128, 32, 4, 2, 2>("f32 128x128x32 (8 waves) occ2", n, reps)) return 1;
        if (run<float, 96, 128, 32, 1, 4, 1>("f32 96x128x32 (1x4 waves) occ1", n, reps)) return 1;
        if (run<float, 96, 128, 32, 1, 4, 2>("f32 96x128x32 (1x4 waves) occ2", n, reps)) return 1;
        if (run<float, 192, 128, 32, 2, 4, 2>("f32 192x128x32 (2x4 waves) occ2", n, reps)) return 1;
        if (run<double, 128, 128, 16, 2, 2, 1>("f64 128x128x16 (4 waves) occ1", n, reps / 2)) return 1;
        if (run<double, 96, 128, 16, 1, 4, 1>("f64 96x128x16 (1x4 waves) occ1", n, reps / 2)) return 1;
        if (run<double, 96, 128, 16, 1, 4, 2>("f64 96x128x16 (1x4 waves) occ2", n, reps / 2)) return 1;
      }
    return 0;
  }
  for (int round = 0; round < 2; ++round) {
    if (run<float, 64, 64, 32, 2, 2, 1>("f32 64x64x32 occ1", 1024, 200)) return 1;
    if (run<float, 128, 128, 32, 4, 2, 2>("f32 128x128x32 (8 waves)", 2048, 50)) return 1;
    if (run<float, 128, 128, 32, 4, 2, 2>("f32 128x128x32 (8 waves)", 4096, 10)) return 1;
  }
  return 0;
}
