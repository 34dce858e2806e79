// Drives the C ABI's host side over the stub HIP runtime (hip_stub.cpp) under AddressSanitizer and
// UndefinedBehaviorSanitizer: argument rules, planner and launch arithmetic for a walk of shapes and every pinned
// family, scratch / auxiliary buffer growth, the no-allocation entry point, copies, events, handle lifetime, the
// multi-GPU layer's partition helpers and a one-rank communicator.  Kernels do not run (launches are checked for a
// legal configuration and dropped), so nothing here checks results: the sanitizers' reports, the status codes and
// the allocation balance at the end are the outcome.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "sbamd_dist.h"
#include "sbamd_gemm.h"

extern "C" long hip_stub_launches();
extern "C" long hip_stub_live_allocations();

static int g_failed = 0;
#define EXPECT(cond)                                                            \
  do {                                                                          \
    if (!(cond)) {                                                              \
      std::fprintf(stderr, "%s:%d: expectation failed: %s\n", __FILE__, __LINE__, #cond); \
      ++g_failed;                                                               \
    }                                                                           \
  } while (0)

static int one_rank_allgather(void*, const void* send, void* recv, size_t bytes) {
  std::memcpy(recv, send, bytes);
  return 0;
}

int main() {
  sbamd_handle_t h = nullptr, h1 = nullptr, bad = nullptr;
  EXPECT(sbamd_create(nullptr, 0, nullptr) != SBAMD_OK);
  EXPECT(sbamd_create(&bad, 7, nullptr) != SBAMD_OK && bad == nullptr);
  EXPECT(sbamd_create(&h, 0, nullptr) == SBAMD_OK);
  EXPECT(sbamd_create(&h1, 1, nullptr) == SBAMD_OK);
  int cus = 0;
  EXPECT(sbamd_get_num_compute_units(h, &cus) == SBAMD_OK && cus == 256);
  void* stream = nullptr;
  EXPECT(sbamd_get_stream(h, &stream) == SBAMD_OK && stream);
  EXPECT(sbamd_wait(nullptr) != SBAMD_OK);
  EXPECT(std::strlen(sbamd_status_string(SBAMD_OK)) > 0 && std::strlen(sbamd_status_string(12345)) > 0);

  // one pool of "device" memory big enough for every shape of the walk (elements of 8 bytes)
  const int64_t cap = int64_t(1) << 22;
  void *a = nullptr, *b = nullptr, *c = nullptr;
  EXPECT(sbamd_malloc(h, &a, cap * 8) == SBAMD_OK && sbamd_malloc(h, &b, cap * 8) == SBAMD_OK && sbamd_malloc(h, &c, cap * 8) == SBAMD_OK);
  std::vector<double> host(1 << 12, 1.0);
  sbamd_event_t e0 = nullptr, e1 = nullptr;
  EXPECT(sbamd_event_record(h, &e0) == SBAMD_OK);
  EXPECT(sbamd_copy_to_device(h, a, host.data(), host.size() * 8) == SBAMD_OK);
  EXPECT(sbamd_memset(h, c, 0, cap * 8) == SBAMD_OK);
  EXPECT(sbamd_fill_uniform(h, b, cap, 1, 7, -2.0, 5.0) == SBAMD_OK);
  EXPECT(sbamd_fill_uniform(h, b, cap, 9, 7, -2.0, 5.0) != SBAMD_OK);
  EXPECT(sbamd_copy_to_host(h, host.data(), a, host.size() * 8) == SBAMD_OK);
  EXPECT(sbamd_event_record(h, &e1) == SBAMD_OK);
  EXPECT(sbamd_stream_wait_event(h, e0) == SBAMD_OK && sbamd_event_synchronize(e1) == SBAMD_OK);
  float ms = -1;
  EXPECT(sbamd_event_elapsed_ms(e0, e1, &ms) == SBAMD_OK && ms >= 0);
  EXPECT(sbamd_event_destroy(e0) == SBAMD_OK && sbamd_event_destroy(e1) == SBAMD_OK);

  // ---- argument rules (nothing may be launched)
  const long before_rules = hip_stub_launches();
  const float* fa = (const float*)a;
  const float* fb = (const float*)b;
  float* fc = (float*)c;
  EXPECT(sbamd_sgemm(nullptr, 'n', 'n', 4, 4, 4, 1, fa, 4, fb, 4, 0, fc, 4) == SBAMD_INVALID_HANDLE);
  EXPECT(sbamd_sgemm(h, 'x', 'n', 4, 4, 4, 1, fa, 4, fb, 4, 0, fc, 4) == SBAMD_INVALID_TRANSA);
  EXPECT(sbamd_sgemm(h, 'n', 'q', 4, 4, 4, 1, fa, 4, fb, 4, 0, fc, 4) == SBAMD_INVALID_TRANSB);
  EXPECT(sbamd_sgemm(h, 'n', 'n', -1, 4, 4, 1, fa, 4, fb, 4, 0, fc, 4) != SBAMD_OK);
  EXPECT(sbamd_sgemm(h, 'n', 'n', 4, 4, 4, 1, fa, -4, fb, 4, 0, fc, 4) != SBAMD_OK);  // (a leading dimension below the row count is the caller's business, as in the reference)
  EXPECT(sbamd_sgemm(h, 'n', 'n', 4, 4, 4, 1, nullptr, 4, fb, 4, 0, fc, 4) != SBAMD_OK);
  EXPECT(sbamd_sgemm_strided_batched(h, 'n', 'n', 4, 4, 4, 1, fa, 4, 16, fb, 4, 16, 0, fc, 4, 8, 3) != SBAMD_OK);   // stride_c < ldc * n
  EXPECT(sbamd_sgemm_strided_batched(h, 'n', 'n', 4, 4, 4, 1, fa, 4, -16, fb, 4, 16, 0, fc, 4, 16, 3) != SBAMD_OK);
  EXPECT(sbamd_sgemm_strided_batched(h, 'n', 'n', 4, 4, 4, 1, fa, 4, 16, fb, 4, 16, 0, fc, 4, 16, int64_t(1) << 31) == SBAMD_INVALID_VALUE);
  EXPECT(sbamd_sgemm_batched(h, 'n', 'n', 4, 4, 4, 1, fa, 4, fb, 4, 0, fc, 4, 3, 77) != SBAMD_OK);
  EXPECT(sbamd_ssymm(h, 'l', 'x', 4, 4, 1, fa, 4, fb, 4, 0, fc, 4) == SBAMD_INVALID_UPLO);
  EXPECT(sbamd_ssymm(h, 'x', 'u', 4, 4, 1, fa, 4, fb, 4, 0, fc, 4) == SBAMD_INVALID_SIDE);
  EXPECT(sbamd_strsm(h, 'l', 'u', 'n', 'x', 4, 4, 1, fa, 4, fc, 4) != SBAMD_OK);
  EXPECT(sbamd_set_kernel(h, 99, 0) != SBAMD_OK);
  EXPECT(hip_stub_launches() == before_rules);
  // empty problems and alpha == 0
  EXPECT(sbamd_sgemm(h, 'n', 'n', 0, 4, 4, 1, fa, 1, fb, 4, 0, fc, 1) == SBAMD_OK && sbamd_last_kernel(h) == -1);
  EXPECT(sbamd_sgemm(h, 'n', 'n', 4, 4, 0, 1, fa, 4, fb, 1, 0.5f, fc, 4) == SBAMD_OK);
  EXPECT(sbamd_sgemm(h, 'x', 'y', 4, 4, 4, 0, nullptr, 4, nullptr, 4, 0.5f, fc, 4) == SBAMD_OK);  // alpha == 0 comes before the flags
  EXPECT(sbamd_dgemm(h, 'n', 'n', 4, 4, 4, 0, nullptr, 4, nullptr, 4, 0.0, (double*)c, 6) == SBAMD_OK);

  // ---- a walk of shapes through the planner, every transpose, both types, the three batch forms
  std::mt19937_64 rng(20240607);
  auto pick = [&](int64_t lo, int64_t hi) { return lo + int64_t(rng() % uint64_t(hi - lo + 1)); };
  int64_t depth = 0, chunk = 0;
  int tm = 0, tn = 0, fused = 0;
  for (int i = 0; i < 600; ++i) {
    int64_t m = pick(1, 700), n = pick(1, 700), k = pick(1, 3000);
    if (i % 5 == 0) m = pick(1, 64), n = pick(1, 64), k = pick(1, 70);
    if (i % 7 == 0) k = pick(1, 40) * 512;
    if (i % 11 == 0) m = pick(1, 24) * 96, n = pick(1, 24) * 64, k = pick(1, 16) * 64;
    while (m * k > cap || k * n > cap || m * n > cap) k = (k + 1) / 2, m = (m + 1) / 2;
    const char ta = "nt"[rng() & 1], tb = "ntc"[rng() % 3];
    const int64_t lda = (ta == 'n' ? m : k) + (i % 3 == 0), ldb = (tb == 'n' ? k : n) + (i % 4 == 0), ldc = m + (i % 6 == 0);
    const float beta = i % 2 ? 0.5f : 0.0f;
    const bool f64 = i % 3 == 1;
    int st;
    if (i % 13 == 0) {
      const int64_t fit = cap / (std::max(lda, ldc) * std::max(std::max(k, n), std::max(m, ldb)) + 1);
      const int64_t bs = std::max<int64_t>(1, std::min<int64_t>(pick(2, 40), fit));
      const int layout = i % 26 == 0 ? SBAMD_BATCH_INTERLEAVED : SBAMD_BATCH_STRIDED;
      st = f64 ? sbamd_dgemm_batched(h, ta, tb, m, n, k, 1.5, (const double*)a, lda, (const double*)b, ldb, beta, (double*)c, ldc, bs, layout)
               : sbamd_sgemm_batched(h, ta, tb, m, n, k, 1.5f, fa, lda, fb, ldb, beta, fc, ldc, bs, layout);
    } else {
      st = f64 ? sbamd_dgemm(h, ta, tb, m, n, k, 1.5, (const double*)a, lda, (const double*)b, ldb, beta, (double*)c, ldc)
               : sbamd_sgemm(h, ta, tb, m, n, k, 1.5f, fa, lda, fb, ldb, beta, fc, ldc);
    }
    if (st != SBAMD_OK) std::fprintf(stderr, "walk case %d (%lld x %lld x %lld): %s\n", i, (long long)m, (long long)n, (long long)k, sbamd_status_string(st));
    EXPECT(st == SBAMD_OK);
    EXPECT(sbamd_last_splitk(h, &depth, &chunk) == SBAMD_OK && depth >= 1 && depth * chunk >= k);
    EXPECT(sbamd_last_tile(h, &tm, &tn, &fused) == SBAMD_OK && tm >= 0 && tn >= 0);
  }
  // ---- every pinned family, K cuts on every tile, the halves
  const int families[] = {SBAMD_KERNEL_LARGE, SBAMD_KERNEL_SMALL, SBAMD_KERNEL_HUGE, SBAMD_KERNEL_TINY, SBAMD_KERNEL_NAIVE, SBAMD_KERNEL_MEDIUM};
  for (int family : families)
    for (int param = 0; param < 2; ++param)
      for (int f64 = 0; f64 < 2; ++f64) {
        EXPECT(sbamd_set_kernel(h, family, param) == SBAMD_OK);
        const int st = f64 ? sbamd_dgemm(h, 't', 'n', 300, 200, 520, 1.0, (const double*)a, 520, (const double*)b, 520, 0.0, (double*)c, 300)
                           : sbamd_sgemm(h, 'n', 't', 300, 200, 520, 1.0f, fa, 300, fb, 200, 1.0f, fc, 300);
        EXPECT(st == SBAMD_OK);
      }
  for (int tile : {1, 2, 3, 6, 7})
    for (int d : {2, 3, 7, 64})
      for (int fuse = 0; fuse < 2; ++fuse) {
        EXPECT(sbamd_set_kernel(h, SBAMD_KERNEL_SPLITK, d | (tile << 16) | (fuse << 24)) == SBAMD_OK);
        EXPECT(sbamd_sgemm(h, 'n', 'n', 200, 130, 4096, 1.0f, fa, 200, fb, 4096, 0.5f, fc, 200) == SBAMD_OK);
        EXPECT(sbamd_dgemm(h, 't', 't', 97, 131, 1002, 1.0, (const double*)a, 1002, (const double*)b, 131, 0.5, (double*)c, 97) == SBAMD_OK);
        EXPECT(sbamd_last_kernel(h) == SBAMD_KERNEL_SPLITK);
      }
  EXPECT(sbamd_set_kernel(h, SBAMD_KERNEL_AUTO, 0) == SBAMD_OK);

  // ---- the no-allocation entry point: nothing allocated, a pinned cut beyond the fixed array refused
  const long live = hip_stub_live_allocations();
  EXPECT(sbamd_sgemm_noalloc(h1, 'n', 'n', 256, 256, 16384, 1.0f, fa, 256, fb, 16384, 0.0f, fc, 256) == SBAMD_OK);
  EXPECT(sbamd_last_splitk(h1, &depth, &chunk) == SBAMD_OK && depth > 1);
  EXPECT(hip_stub_live_allocations() == live);
  EXPECT(sbamd_set_kernel(h1, SBAMD_KERNEL_SPLITK, 200 | (1 << 16)) == SBAMD_OK);
  const long launches = hip_stub_launches();
  EXPECT(sbamd_sgemm_noalloc(h1, 'n', 'n', 1024, 1024, 16384 * 2, 1.0f, fa, 1024, fb, 16384 * 2, 0.0f, fc, 1024) == SBAMD_SCRATCH_TOO_SMALL);
  EXPECT(hip_stub_launches() == launches && hip_stub_live_allocations() == live);
  EXPECT(sbamd_set_kernel(h1, SBAMD_KERNEL_AUTO, 0) == SBAMD_OK);

  // ---- 16-bit pairs, symm, trsm
  const uint16_t* ha = (const uint16_t*)a;
  const uint16_t* hb = (const uint16_t*)b;
  for (int i = 0; i < 120; ++i) {
    const int64_t m = pick(1, 600), n = pick(1, 600), k = pick(1, 2000);
    const char ta = "nt"[rng() & 1], tb = "nt"[rng() & 1];
    const int64_t lda = (ta == 'n' ? m : k) + (i % 3 == 0) * 8, ldb = (tb == 'n' ? k : n) + (i % 4 == 0), ldc = m;
    int st = SBAMD_OK;
    switch (i % 5) {
      case 0: st = sbamd_hgemm(h, ta, tb, m, n, k, 1.0f, ha, lda, hb, ldb, 0.5f, (uint16_t*)c, ldc); break;
      case 1: st = sbamd_hsgemm(h, ta, tb, m, n, k, 1.0f, ha, lda, hb, ldb, 0.0f, fc, ldc); break;
      case 2: st = sbamd_bsgemm(h, ta, tb, m, n, k, 1.0f, ha, lda, hb, ldb, 0.5f, fc, ldc); break;
      case 3: st = sbamd_bsgemm_strided_batched(h, ta, tb, m, n, k, 1.0f, ha, lda, lda * std::max(m, k), hb, ldb, 0, 0.5f, fc, ldc, ldc * n, 3); break;
      case 4: st = sbamd_hgemm_batched(h, ta, tb, std::min<int64_t>(m, 40), std::min<int64_t>(n, 40), std::min<int64_t>(k, 40), 1.0f, ha, 48, hb, 48, 0.5f, (uint16_t*)c, 48, 5, SBAMD_BATCH_INTERLEAVED); break;
    }
    EXPECT(st == SBAMD_OK);
  }
  for (char side : {'l', 'r'})
    for (char uplo : {'u', 'l'})
      for (char trans : {'n', 't'}) {
        EXPECT(sbamd_ssymm(h, side, uplo, 300, 200, 1.5f, fa, 300, fb, 300, 0.5f, fc, 300) == SBAMD_OK);
        EXPECT(sbamd_dsymm(h, side, uplo, 65, 130, 1.5, (const double*)a, 130, (const double*)b, 65, 0.0, (double*)c, 65) == SBAMD_OK);
        EXPECT(sbamd_strsm(h, side, uplo, trans, 'n', 500, 260, 1.0f, fa, 500, fc, 500) == SBAMD_OK);
        EXPECT(sbamd_dtrsm(h, side, uplo, trans, 'u', 37, 1000, 2.0, (const double*)a, 1000, (double*)c, 37) == SBAMD_OK);
        int64_t diag = -1, panel = -1;
        EXPECT(sbamd_last_trsm_launches(h, &diag, &panel) == SBAMD_OK && diag >= 0 && panel >= 0);
      }

  // ---- the multi-GPU layer: partition helpers, and a one-rank communicator over the mesh transport
  for (int world : {1, 2, 3, 4, 6, 8, 16}) {
    int pr = 0, pc = 0;
    sbamd_dist_grid(world, &pr, &pc);
    EXPECT(pr * pc == world);
    int64_t covered = 0;
    for (int r = 0; r < world; ++r) {
      int64_t begin = 0, end = 0;
      sbamd_dist_split_range(1000, world, r, &begin, &end);
      EXPECT(begin == covered && end >= begin);
      covered = end;
      int64_t r0, r1, c0, c1;
      EXPECT(sbamd_dist_block_of(r, pr, pc, 777, 1001, &r0, &r1, &c0, &c1) == SBAMD_OK && r0 <= r1 && r1 <= 777 && c0 <= c1 && c1 <= 1001);
    }
    EXPECT(covered == 1000);
    int64_t widths[SBAMD_DIST_MAX_BANDS];
    for (int bands : {0, 1, 3, 4, 16}) {
      const int nb = sbamd_dist_band_widths(8192, bands, world, widths);
      int64_t sum = 0;
      for (int i = 0; i < nb; ++i) sum += widths[i];
      EXPECT(nb >= 1 && nb <= SBAMD_DIST_MAX_BANDS && sum == 8192);
    }
  }
  sbamd_comm_t comm = nullptr;
  EXPECT(sbamd_dist_create(&comm, h, 0, 1, one_rank_allgather, nullptr, SBAMD_DIST_MESH, size_t(8) << 20) == SBAMD_OK);
  if (comm) {
    int rank = -1, world = -1, transports = 0, window = -1;
    EXPECT(sbamd_dist_rank(comm, &rank, &world, &transports) == SBAMD_OK && rank == 0 && world == 1);
    EXPECT(sbamd_dist_register(comm, c, cap * 8, &window) == SBAMD_OK && window >= 0);
    EXPECT(sbamd_dist_barrier(comm) == SBAMD_OK);
    EXPECT(sbamd_dist_gathered_sgemm(comm, 'n', 'n', 512, 512, 512, 1.0f, fa, 512, fb, 512, 0.0f, fc, 512, window, 0) == SBAMD_OK);
    int64_t c0[SBAMD_DIST_MAX_BANDS], c1[SBAMD_DIST_MAX_BANDS], dp[SBAMD_DIST_MAX_BANDS], kc[SBAMD_DIST_MAX_BANDS];
    EXPECT(sbamd_dist_last_bands(comm, c0, c1, dp, kc) >= 1);
    EXPECT(sbamd_dist_splitk_dgemm(comm, 't', 'n', 128, 128, 2048, 1.0, (const double*)a, 2048, (const double*)b, 2048, 0.5, (double*)c, 128, window, 1) == SBAMD_OK);
    EXPECT(sbamd_dist_block_sgemm(comm, 1, 1, 'n', 'n', 512, 512, 512, 1.0f, fa, 512, fb, 512, 0.0f, fc, 512) == SBAMD_OK);
    EXPECT(sbamd_dist_all_gather_c(comm, 1, 1, 512, 512, c, 512, 4, window) == SBAMD_OK);
    EXPECT(sbamd_dist_status(comm) == SBAMD_OK);
    EXPECT(sbamd_dist_unregister(comm, window) == SBAMD_OK);
    EXPECT(sbamd_dist_destroy(comm) == SBAMD_OK);
  }

  EXPECT(sbamd_wait(h) == SBAMD_OK && sbamd_wait(h1) == SBAMD_OK);
  EXPECT(sbamd_free(h, a) == SBAMD_OK && sbamd_free(h, b) == SBAMD_OK);
  EXPECT(sbamd_device_free(0, c) == SBAMD_OK);
  EXPECT(sbamd_destroy(h) == SBAMD_OK && sbamd_destroy(h1) == SBAMD_OK);
  EXPECT(sbamd_destroy(nullptr) != SBAMD_OK);
  const long left = hip_stub_live_allocations();
  if (left) std::fprintf(stderr, "%ld device allocations outlive their handles\n", left);
  EXPECT(left == 0);
  std::printf("host logic walk: %ld launches checked, %d expectations failed\n", hip_stub_launches(), g_failed);
  return g_failed ? 1 : 0;
}
