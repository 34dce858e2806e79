// Lab: one product (fp32 or fp64, optionally strided-batched) launched 12 times through the C ABI with the planner's
// choice or a pinned kernel family, for rocprofv3 traces and counter passes of ONE kernel family at a time.
// usage: one_family <f32|f64> M N K BATCH [kernel [param]]
//   g++ -O2 -std=c++17 -Iinclude tools/one_family.cpp -o tools/bin/one_family -Lsycl_blas_amd/lib -lsbamd \
//       -Wl,-rpath,'$ORIGIN/../../sycl_blas_amd/lib'
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "sbamd_gemm.h"

int main(int argc, char** argv) {
  if (argc < 6) return std::printf("usage: one_family <f32|f64> M N K BATCH [kernel [param]]\n"), 1;
  const bool f64 = std::strcmp(argv[1], "f64") == 0;
  const int64_t m = std::atoll(argv[2]), n = std::atoll(argv[3]), k = std::atoll(argv[4]), bs = std::atoll(argv[5]);
  const size_t es = f64 ? 8 : 4;
  sbamd_handle_t h;
  if (sbamd_create(&h, 0, nullptr) != SBAMD_OK) return 1;
  void *a, *b, *c;
  if (sbamd_malloc(h, &a, size_t(m) * k * bs * es) || sbamd_malloc(h, &b, size_t(k) * n * bs * es) || sbamd_malloc(h, &c, size_t(m) * n * bs * es)) return 1;
  if (sbamd_fill_uniform(h, a, m * k * bs, f64, 11, -1.0, 1.0) || sbamd_fill_uniform(h, b, k * n * bs, f64, 12, -1.0, 1.0)) return 1;
  if (argc > 6 && sbamd_set_kernel(h, std::atoi(argv[6]), argc > 7 ? std::atoi(argv[7]) : 0) != SBAMD_OK) return 1;
  auto call = [&]() {
    int st;
    if (f64)
      st = bs == 1 ? sbamd_dgemm(h, 'n', 'n', m, n, k, 1.0, (const double*)a, m, (const double*)b, k, 0.0, (double*)c, m)
                   : sbamd_dgemm_strided_batched(h, 'n', 'n', m, n, k, 1.0, (const double*)a, m, m * k, (const double*)b, k, k * n, 0.0,
                                                 (double*)c, m, m * n, bs);
    else
      st = bs == 1 ? sbamd_sgemm(h, 'n', 'n', m, n, k, 1.0f, (const float*)a, m, (const float*)b, k, 0.0f, (float*)c, m)
                   : sbamd_sgemm_strided_batched(h, 'n', 'n', m, n, k, 1.0f, (const float*)a, m, m * k, (const float*)b, k, k * n, 0.0f,
                                                 (float*)c, m, m * n, bs);
    return st;
  };
  for (int i = 0; i < 12; ++i)
    if (int st = call(); st != SBAMD_OK) return std::printf("status %d: %s\n", st, sbamd_status_string(st)), 2;
  if (sbamd_wait(h) != SBAMD_OK) return 3;
  if (std::getenv("ONE_FAMILY_TIME")) {  // best of five loops of 200 calls
    double best = 1e30;
    for (int round = 0; round < 5; ++round) {
      sbamd_event_t e0 = nullptr, e1 = nullptr;
      float ms = 0;
      sbamd_event_record(h, &e0);
      for (int i = 0; i < 200; ++i) call();
      sbamd_event_record(h, &e1);
      if (sbamd_wait(h) != SBAMD_OK || sbamd_event_elapsed_ms(e0, e1, &ms) != SBAMD_OK) return 3;
      sbamd_event_destroy(e0), sbamd_event_destroy(e1);
      if (ms / 200 < best) best = ms / 200;
    }
    std::printf("%.2f us per call  ", best * 1e3);
  }
  int64_t depth = 0, chunk = 0;
  sbamd_last_splitk(h, &depth, &chunk);
  std::printf("kernel %d depth %lld chunk %lld\n", sbamd_last_kernel(h), (long long)depth, (long long)chunk);
  sbamd_free(h, a), sbamd_free(h, b), sbamd_free(h, c);
  sbamd_destroy(h);
  return 0;
}
