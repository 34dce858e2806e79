// Lab: one fp32 NN product launched 20 times through the C ABI, for rocprofv3 traces and counter passes of a single
// shape (the planner's choice, or a pinned family).  usage: one_shape M N K [kernel [param]]
//   g++ -O2 -std=c++17 -Iinclude tools/one_shape.cpp -o tools/bin/one_shape -Lsycl_blas_amd/lib -lsbamd \
//       -Wl,-rpath,'$ORIGIN/../../sycl_blas_amd/lib'
#include <cstdio>
#include <cstdlib>

#include "sbamd_gemm.h"

int main(int argc, char** argv) {
  if (argc < 4) return std::printf("usage: one_shape M N K [kernel [param]]\n"), 1;
  const int64_t m = std::atoll(argv[1]), n = std::atoll(argv[2]), k = std::atoll(argv[3]);
  sbamd_handle_t h;
  if (sbamd_create(&h, 0, nullptr) != SBAMD_OK) return 1;
  void *a, *b, *c;
  if (sbamd_malloc(h, &a, size_t(m) * k * 4) || sbamd_malloc(h, &b, size_t(k) * n * 4) || sbamd_malloc(h, &c, size_t(m) * n * 4)) return 1;
  if (sbamd_fill_uniform(h, a, m * k, 0, 11, -1.0, 1.0) || sbamd_fill_uniform(h, b, k * n, 0, 12, -1.0, 1.0)) return 1;
  if (argc > 4 && sbamd_set_kernel(h, std::atoi(argv[4]), argc > 5 ? std::atoi(argv[5]) : 0) != SBAMD_OK) return 1;
  for (int i = 0; i < 20; ++i)
    if (sbamd_sgemm(h, 'n', 'n', m, n, k, 1.0f, (const float*)a, m, (const float*)b, k, 0.0f, (float*)c, m) != SBAMD_OK) return 2;
  if (sbamd_wait(h) != SBAMD_OK) return 3;
  int64_t depth = 0, chunk = 0;
  sbamd_last_splitk(h, &depth, &chunk);
  std::printf("kernel %d depth %lld chunk %lld\n", sbamd_last_kernel(h), (long long)depth, (long long)chunk);
  sbamd_free(h, a), sbamd_free(h, b), sbamd_free(h, c);
  sbamd_destroy(h);
  return 0;
}
