// Development harness for the 16-bit kernels: runs several kernel variants in one
// process, interleaved over rounds, on uniform random bf16 operands, and checks each
// against the 128x128 kernel of the library bit for bit.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Isycl_blas_amd/csrc tools/bf16_lab.hip -o tools/bin/bf16_lab
//   tools/bin/bf16_lab [n=8192] [rounds=5] [trans=nn] [ld_pad=0]      (ld_pad: leading dimension of A and B = n + ld_pad)
#include <cmath>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "gemm_half_dma.hpp"

using namespace sbamd;

#define CHECK(x)                                                                         \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      std::exit(1);                                                                      \
    }                                                                                    \
  } while (0)

struct Variant {
  std::string name;
  std::function<void(const GemmArgs16<float>&)> launch;
  bool check;
  std::vector<float> ms;
};

template <bool TA, bool TB>
void add_variants(std::vector<Variant>& v) {
  {
    auto k = gemm_mfma16_kernel<Bf16Tag, float, 128, 128, 64, 2, 2, TA, TB>;
    constexpr size_t lds = 2 * size_t(TileLoader16<128, 64, 256, TA>::ELEMS + TileLoader16<128, 64, 256, !TB>::ELEMS) * 2;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    v.push_back({"128x128 register-staged", [k](const GemmArgs16<float>& g) {
                   GemmArgs16<float> h = g;
                   h.tiles_m = int((g.m + 127) / 128), h.tiles_n = int((g.n + 127) / 128);
                   hipLaunchKernelGGL(k, dim3(h.tiles_m * h.tiles_n), dim3(256), lds, 0, h);
                 }, false, {}});
  }
  auto add_ring = [&](const char* name, auto kernel, bool check, int threads = DMA_NT) {
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS_BYTES));
    v.push_back({name, [kernel, threads](const GemmArgs16<float>& g) {
                   GemmArgs16<float> h = g;
                   h.tiles_m = int((g.m + 255) / 256), h.tiles_n = int((g.n + 255) / 256);
                   hipLaunchKernelGGL(kernel, dim3(h.tiles_m * h.tiles_n), dim3(threads), RING_LDS_BYTES, 0, h);
                 }, check, {}});
  };
  add_ring("256x256 ring (5 x BK32)", gemm_mfma16_ring_kernel<Bf16Tag, float, TA, TB, 0>, true);
  add_ring("256x256 ring, 4 waves of 128x128", gemm_mfma16_ring_kernel<Bf16Tag, float, TA, TB, 0, 256, 256, 2, 2>, true, 256);
#ifdef LAB_ABLATE
  add_ring("ring, no loads in loop", gemm_mfma16_ring_kernel<Bf16Tag, float, TA, TB, 1>, false);
  add_ring("ring, no mfma", gemm_mfma16_ring_kernel<Bf16Tag, float, TA, TB, 2>, false);
  add_ring("ring, no mfma, no frag reads", gemm_mfma16_ring_kernel<Bf16Tag, float, TA, TB, 6>, false);
  add_ring("ring, no loads, no frag reads", gemm_mfma16_ring_kernel<Bf16Tag, float, TA, TB, 5>, false);
#endif
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 8192;
  const int rounds = argc > 2 ? std::atoi(argv[2]) : 5;
  const std::string trans = argc > 3 ? argv[3] : "nn";
  const int ld_pad = argc > 4 ? std::atoi(argv[4]) : 0;
  const size_t count = size_t(n) * n, count_ab = size_t(n + ld_pad) * n;
  std::vector<uint16_t> host(count_ab);
  uint64_t s = 88172645463325252ull;
  auto fill = [&]() {
    for (size_t i = 0; i < count_ab; ++i) {
      s ^= s << 13, s ^= s >> 7, s ^= s << 17;
      const float f = float(double(s >> 11) / double(1ull << 53) * 2.0 - 1.0);  // uniform [-1, 1)
      uint32_t bits;
      std::memcpy(&bits, &f, 4);
      host[i] = uint16_t((bits + 0x7fffu + ((bits >> 16) & 1u)) >> 16);
    }
  };
  uint16_t *a, *b;
  float *c, *cref;
  CHECK(hipMalloc(&a, count_ab * 2));
  CHECK(hipMalloc(&b, count_ab * 2));
  CHECK(hipMalloc(&c, count * 4));
  CHECK(hipMalloc(&cref, count * 4));
  fill();
  CHECK(hipMemcpy(a, host.data(), count_ab * 2, hipMemcpyHostToDevice));
  fill();
  CHECK(hipMemcpy(b, host.data(), count_ab * 2, hipMemcpyHostToDevice));

  GemmArgs16<float> g{};
  g.a = a, g.b = b, g.c = c;
  g.m = g.n = g.k = n;
  g.lda = g.ldb = n + ld_pad;
  g.ldc = n;
  g.batch = 1;
  g.vec_a = g.vec_b = 1;
  g.alpha = 1.0f, g.beta = 0.0f, g.beta_zero = 1;

  std::vector<Variant> vars;
  if (trans == "nn") add_variants<false, false>(vars);
  else if (trans == "nt") add_variants<false, true>(vars);
  else if (trans == "tn") add_variants<true, false>(vars);
  else add_variants<true, true>(vars);

  // reference result from the first variant
  GemmArgs16<float> gr = g;
  gr.c = cref;
  vars[0].launch(gr);
  CHECK(hipDeviceSynchronize());
  std::vector<float> want(count), got(count);
  CHECK(hipMemcpy(want.data(), cref, count * 4, hipMemcpyDeviceToHost));
  for (auto& v : vars) {
    if (!v.check) continue;
    CHECK(hipMemset(c, 0xff, count * 4));
    v.launch(g);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(got.data(), c, count * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, far = 0;
    double worst = 0;
    for (size_t i = 0; i < count; ++i) {
      bad += std::memcmp(&got[i], &want[i], 4) != 0;
      const double err = std::fabs(double(got[i]) - double(want[i]));
      worst = err > worst ? err : worst;
      far += !(err <= 1e-3 * (1.0 + std::fabs(double(want[i]))));  // also catches NaN
    }
    std::printf("check %-40s %s (%zu of %zu differ in bits, %zu beyond 1e-3 relative, max abs diff %.3g)\n", v.name.c_str(),
                far ? "MISMATCH" : "ok", bad, count, far, worst);
  }

  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int reps = 5;
  for (int r = 0; r < rounds; ++r)
    for (auto& v : vars) {
      v.launch(g);  // warm-up
      CHECK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) v.launch(g);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      v.ms.push_back(ms / reps);
    }
  for (auto& v : vars) {
    std::sort(v.ms.begin(), v.ms.end());
    const double flops = 2.0 * n * double(n) * n;
    std::printf("%-40s %s %d^3  median %.3f ms %7.1f TFLOP/s   best %.3f ms %7.1f TFLOP/s\n", v.name.c_str(), trans.c_str(), n,
                v.ms[v.ms.size() / 2], flops / v.ms[v.ms.size() / 2] / 1e9, v.ms[0], flops / v.ms[0] / 1e9);
  }
  return 0;
}
