// Prints the planner's choice for shapes read from stdin ("m n k" per line), without a GPU: the plan depends on
// the shape, the element type and the compute-unit count only.
// usage: plan_probe <float|double> [compute units = 256] < shapes     -> "m n k tile_m tile_n depth fused"
//   hipcc -O1 -std=c++17 --offload-arch=gfx950 -Iinclude -Isycl_blas_amd/csrc tools/plan_probe.hip -o tools/bin/plan_probe
#include <cstdio>
#include <cstring>

#include "gemm_dispatch.hpp"

template <typename T>
int run(int cus) {
  long long m, n, k;
  while (std::scanf("%lld %lld %lld", &m, &n, &k) == 3) {
    const sbamd::Plan p = sbamd::plan_gemm<T>(m, n, k, 1, cus);
    const int tm = p.tile == sbamd::TILE_256 ? 256 : p.tile == sbamd::TILE_128 ? 128 : p.tile == sbamd::TILE_96 ? 96 : p.tile == sbamd::TILE_192 ? 192 : p.tile == sbamd::TILE_32 ? 32 : 64;
    const int tn = p.tile == sbamd::TILE_256 ? sbamd::Tiles<T>::HUGE_BN : tm;
    std::printf("%lld %lld %lld %d %d %lld %d\n", m, n, k, tm, tn, (long long)p.depth, p.fused ? 1 : 0);
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) return std::printf("usage: plan_probe <float|double> [compute units]\n"), 1;
  const int cus = argc > 2 ? std::atoi(argv[2]) : 256;
  return std::strcmp(argv[1], "double") == 0 ? run<double>(cus) : run<float>(cus);
}
