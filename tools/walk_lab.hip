// Lab: timing-only ablation of the tile walk of gemm_batch_stream_kernel on 8192 x 8192 x 64 (fp32 NN):
//   -DSBAMD_WALK_ABLATE=<mask>   1: no C stores  2: no operand requests inside the loop  4: no matrix work
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DSBAMD_WALK_ABLATE=<mask> -Isycl_blas_amd/csrc -Iinclude tools/walk_lab.hip -o tools/bin/walk_lab_<mask>
#include <cstdio>
#include <vector>

#include "gemm_mfma.hpp"

using namespace sbamd;
#ifndef SBAMD_WALK_ABLATE
#define SBAMD_WALK_ABLATE 0
#endif
#ifndef WALK_OCC
#define WALK_OCC 4
#endif
#define CHECK(x)                                                              \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 1;                                                               \
    }                                                                         \
  } while (0)

int main() {
  const int n = 8192, k = 64;
  float *a, *b, *c;
  CHECK(hipMalloc(&a, sizeof(float) * n * k));
  CHECK(hipMalloc(&b, sizeof(float) * n * k));
  CHECK(hipMalloc(&c, sizeof(float) * size_t(n) * n));
  std::vector<float> host(size_t(n) * k);
  for (size_t i = 0; i < host.size(); ++i) host[i] = float((i * 2654435761u % 7000) / 1000.0 - 2.0);
  CHECK(hipMemcpy(a, host.data(), sizeof(float) * n * k, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(b, host.data(), sizeof(float) * n * k, hipMemcpyHostToDevice));
  GemmArgs<float> g{};
  g.a = a; g.b = b; g.c = c;
  g.m = g.n = n; g.k = k;
  g.lda = n; g.ldb = k; g.ldc = n;
  g.batch = 1; g.split_k = 1; g.k_chunk = k;
  g.tiles_m = n / 64; g.tiles_n = n / 64;
  g.vec_a = g.vec_b = 1;
  g.alpha = 1.0f; g.beta = 0.0f; g.beta_zero = 1;
  using K = GemmTile<float, 64, 64, 64, 2, 2, false, false>;
  auto kernel = gemm_batch_stream_kernel<float, 64, 64, false, false, false, false, WALK_OCC>;
  const size_t lds = K::LDS_BYTES / 2;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int grid : {256 * WALK_OCC, 256 * 2, 256}) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, 0, g);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, 0, g);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("walk ablate=%d occ=%d grid=%d: %.1f us\n", SBAMD_WALK_ABLATE, WALK_OCC, grid, ms / 20 * 1e3);
  }
  return 0;
}
