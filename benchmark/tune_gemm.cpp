// GEMM kernel-family tuner for libsbamd, with the command line of the reference's
// auto tuner (tools/auto_tuner/README.md:51-68):
//
//   tune_gemm M K N bs rep [batch_type] [--dtype=float|double] [--trans=nn|nt|tn|tt|all] [--csv=FILE]
//
// The reference tries every tile configuration compiled into the tuner binary
// (tools/auto_tuner/gen/*.json) and prints them from worst to best.  Here the
// candidates are the kernel families of the library -- 256x256 / 256x128 for float / double ("huge"), 128x128
// ("large"), 64x64 ("small"), split-K over each tile at several depths, the naive kernel -- each
// pinned through sbamd_set_kernel, plus what the dispatcher picks by itself
// ("auto"), so a shape where "auto" is not last in the list is a dispatch-table bug.
// Every candidate's result is compared with the first candidate's: families other
// than split-K must agree bit for bit.  Each candidate is timed after 15 ms of its own launches (the clock the chip
// holds depends on what ran just before); with SBAMD_TUNE_AUTO_TWICE set in the environment "auto" is timed a
// second time as the last candidate, as a check on that.
#include <algorithm>
#include <cstdint>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "sbamd_gemm.h"

namespace {

FILE* g_csv = nullptr;  // --csv=FILE: one line per candidate (dtype,ta,tb,m,k,n,bs,kernel,tile,depth,ms,fused, then as launched: kernel,tile_m,tile_n,depth,fused)

void check(int st, sbamd_handle_t h, const char* what) {
  if (st == SBAMD_OK) return;
  std::string msg = std::string(what) + ": " + sbamd_status_string(st);
  if (h && st >= SBAMD_HIP_ERROR) msg += std::string(" (") + sbamd_last_error(h) + ")";
  throw std::runtime_error(msg);
}

struct Candidate {
  std::string name;
  int kernel;
  int param;
  bool exact;  // result must match the other exact candidates bit for bit
};

struct Outcome {
  Candidate cand;
  double ms = 0, gflops = 0;
  bool ok = false, same_bits = true;
  std::string note;
  int ran_kernel = -1, ran_tile_m = 0, ran_tile_n = 0, ran_fused = 0;  // what the library reports it launched
  int64_t ran_depth = 1, ran_chunk = 0;
};

template <typename T>
int gemm_call(sbamd_handle_t h, char ta, char tb, int64_t m, int64_t n, int64_t k, const T* a, int64_t lda, const T* b,
              int64_t ldb, T* c, int64_t ldc, int64_t bs, int batch_type);
template <>
int gemm_call<float>(sbamd_handle_t h, char ta, char tb, int64_t m, int64_t n, int64_t k, const float* a, int64_t lda,
                     const float* b, int64_t ldb, float* c, int64_t ldc, int64_t bs, int batch_type) {
  return sbamd_sgemm_batched(h, ta, tb, m, n, k, 1.0f, a, lda, b, ldb, 0.0f, c, ldc, bs, batch_type);
}
template <>
int gemm_call<double>(sbamd_handle_t h, char ta, char tb, int64_t m, int64_t n, int64_t k, const double* a, int64_t lda,
                      const double* b, int64_t ldb, double* c, int64_t ldc, int64_t bs, int batch_type) {
  return sbamd_dgemm_batched(h, ta, tb, m, n, k, 1.0, a, lda, b, ldb, 0.0, c, ldc, bs, batch_type);
}

template <typename T>
void tune(sbamd_handle_t h, char ta, char tb, int64_t m, int64_t k, int64_t n, int64_t bs, int rep, int batch_type) {
  const int64_t lda = ta == 'n' ? m : k, ldb = tb == 'n' ? k : n, ldc = m;
  const int64_t na = m * k * bs, nb = k * n * bs, nc = m * n * bs;
  void *a = nullptr, *b = nullptr, *c = nullptr;
  check(sbamd_malloc(h, &a, size_t(na) * sizeof(T)), h, "malloc");
  check(sbamd_malloc(h, &b, size_t(nb) * sizeof(T)), h, "malloc");
  check(sbamd_malloc(h, &c, size_t(nc) * sizeof(T)), h, "malloc");
  const int code = sizeof(T) == 4 ? 0 : 1;
  check(sbamd_fill_uniform(h, a, na, code, 11, -1.0, 1.0), h, "fill");
  check(sbamd_fill_uniform(h, b, nb, code, 12, -1.0, 1.0), h, "fill");

  std::vector<Candidate> cands = {{"auto", SBAMD_KERNEL_AUTO, 0, false}};
  if (batch_type == SBAMD_BATCH_INTERLEAVED && bs > 1) {
    cands.push_back({"interleaved 4x4 register tile", SBAMD_KERNEL_INTERLEAVED, 0, true});
  } else {
    cands.push_back({sizeof(T) == 4 ? "huge   256x256 tile, 8 waves" : "huge   256x128 tile, 8 waves", SBAMD_KERNEL_HUGE, 0, true});
    cands.push_back({"medium  96x96  tile, 4 waves (16x16x4 instruction)", SBAMD_KERNEL_MEDIUM, 0, true});
    cands.push_back({"medium 192x192 tile, 8 waves (16x16x4 instruction)", SBAMD_KERNEL_MEDIUM, 1, true});
    cands.push_back({sizeof(T) == 4 ? "large  128x128 tile, 8 waves" : "large  128x128 tile, 4 waves", SBAMD_KERNEL_LARGE, 0, true});
    cands.push_back({"small   64x64  tile, 4 waves", SBAMD_KERNEL_SMALL, 0, true});
    cands.push_back({"small   64x64  tile, register-staged only", SBAMD_KERNEL_SMALL, 1, true});
    if (m * n * bs <= 1024 * 1024) cands.push_back({"tiny    32x32  tile, 1 wave", SBAMD_KERNEL_TINY, 0, true});
    if (bs == 1) {
      static const char* tile_names[] = {"", "64x64", "128x128", sizeof(T) == 4 ? "256x256" : "256x128", "", "", "96x96", "192x192"};
      for (int tile : {1, 2, 3, 6, 7})
        for (int depth : {2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256}) {
          if (k / depth < 64) break;
          // more slabs than it takes to give every compute unit four workgroups only add traffic
          const int64_t bm = tile == 3 ? 256 : tile == 2 ? 128 : tile == 6 ? 96 : tile == 7 ? 192 : 64;
          const int64_t bn = tile == 3 ? (sizeof(T) == 4 ? 256 : 128) : bm;
          const int64_t tiles = ((m + bm - 1) / bm) * ((n + bn - 1) / bn);
          if (tiles * depth > 4096 && depth > 2) break;
          cands.push_back({std::string("splitk ") + tile_names[tile] + " depth " + std::to_string(depth),
                           SBAMD_KERNEL_SPLITK, depth | (tile << 16), false});
        }
      // both halves of K in one workgroup (falls back to the cube route when K does not split into whole K steps)
      cands.push_back({"quarters 32x32 (four K quarters per workgroup)", SBAMD_KERNEL_SPLITK, 4 | (4 << 16) | (1 << 24), false});
      for (int tile : {1, 2, 6})
        cands.push_back({std::string("halves ") + tile_names[tile] + " (two K halves per workgroup)", SBAMD_KERNEL_SPLITK,
                         2 | (tile << 16) | (1 << 24), false});
    }
    if (double(m) * double(n) * double(k) * double(bs) <= 8e9)  // the scalar kernel is for checking, keep it short
      cands.push_back({"naive  one element per lane", SBAMD_KERNEL_NAIVE, 0, true});
  }
  if (std::getenv("SBAMD_TUNE_AUTO_TWICE")) cands.push_back({"auto (again, last)", SBAMD_KERNEL_AUTO, 0, false});

  // The first launches of a process run at a lower clock, and the ramp takes tens of milliseconds: keep the device
  // busy for that long before timing anything, or the first candidate ("auto") reads 1.2x to 1.5x slow.
  {
    check(sbamd_set_kernel(h, SBAMD_KERNEL_AUTO, 0), h, "set_kernel");
    const auto t0 = std::chrono::steady_clock::now();
    do {
      for (int i = 0; i < 4; ++i)
        check(gemm_call<T>(h, ta, tb, m, n, k, (const T*)a, lda, (const T*)b, ldb, (T*)c, ldc, bs, batch_type), h, "gemm");
      check(sbamd_wait(h), h, "wait");
    } while (std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(80));
  }
  std::vector<T> first, got(static_cast<size_t>(nc));
  std::vector<Outcome> outcomes;
  const double flops = 2.0 * double(m) * double(n) * double(k) * double(bs);
  for (const Candidate& cand : cands) {
    Outcome o;
    o.cand = cand;
    try {
      check(sbamd_set_kernel(h, cand.kernel, cand.param), h, "set_kernel");
      auto call = [&]() {
        return gemm_call<T>(h, ta, tb, m, n, k, (const T*)a, lda, (const T*)b, ldb, (T*)c, ldc, bs, batch_type);
      };
      check(call(), h, "gemm");  // warm-up and the run whose result is compared
      check(sbamd_copy_to_host(h, got.data(), c, got.size() * sizeof(T)), h, "copy");
      check(sbamd_wait(h), h, "wait");
      o.ran_kernel = sbamd_last_kernel(h);
      sbamd_last_splitk(h, &o.ran_depth, &o.ran_chunk);
      sbamd_last_tile(h, &o.ran_tile_m, &o.ran_tile_n, &o.ran_fused);
      if (cand.kernel == SBAMD_KERNEL_AUTO) {
        int64_t depth = 1, chunk = 0;
        sbamd_last_splitk(h, &depth, &chunk);
        static const char* names[] = {"auto", "large", "small", "splitk", "naive", "huge", "interleaved", "tiny", "stream", "medium"};
        const int picked = sbamd_last_kernel(h);
        o.note = std::string("-> ") + (picked >= 0 && picked <= 9 ? names[picked] : "?");
        if (depth > 1) o.note += " depth " + std::to_string(depth);
        o.cand.exact = depth <= 1;
      }
      if (o.cand.exact) {
        if (first.empty()) first = got;
        else o.same_bits = std::memcmp(first.data(), got.data(), got.size() * sizeof(T)) == 0;
      }
      // The clock the chip holds depends on what it ran in the last milliseconds (a candidate timed right after a
      // run of short, light launches read up to 9 % faster than the same kernel timed after a heavy one): give
      // every candidate 15 ms of its own launches before its timed loop.
      {
        const auto t0 = std::chrono::steady_clock::now();
        do {
          for (int i = 0; i < 4; ++i) check(call(), h, "gemm");
          check(sbamd_wait(h), h, "wait");
        } while (std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(15));
      }
      // A timed loop of `rep` calls, then three loops long enough to last a millisecond each (a loop of ten
      // 10-microsecond launches reads 10 % off from run to run); the shortest loop counts.
      auto timed = [&](int calls) {
        sbamd_event_t e0 = nullptr, e1 = nullptr;
        check(sbamd_event_record(h, &e0), h, "event");
        for (int i = 0; i < calls; ++i) check(call(), h, "gemm");
        check(sbamd_event_record(h, &e1), h, "event");
        check(sbamd_wait(h), h, "wait");
        float ms = 0;
        check(sbamd_event_elapsed_ms(e0, e1, &ms), h, "elapsed");
        sbamd_event_destroy(e0);
        sbamd_event_destroy(e1);
        return double(ms) / calls;
      };
      o.ms = timed(rep);
      const int longer = int(std::min(400.0, std::ceil(1.0 / std::max(o.ms, 1e-4))));
      if (longer > rep)
        for (int round = 0; round < 3; ++round) o.ms = round == 0 ? timed(longer) : std::min(o.ms, timed(longer));
      o.gflops = flops / (o.ms * 1e6);
      o.ok = true;
    } catch (const std::exception& e) {
      o.note = e.what();
    }
    outcomes.push_back(o);
  }
  sbamd_set_kernel(h, SBAMD_KERNEL_AUTO, 0);
  sbamd_free(h, a), sbamd_free(h, b), sbamd_free(h, c);

  std::stable_sort(outcomes.begin(), outcomes.end(), [](const Outcome& x, const Outcome& y) {
    if (x.ok != y.ok) return !x.ok;  // failures first, then worst to best
    return x.gflops < y.gflops;
  });
  std::printf("== %s GEMM %c%c  M=%lld K=%lld N=%lld  batch=%lld (%s), %d repetitions\n", sizeof(T) == 4 ? "float" : "double",
              ta, tb, (long long)m, (long long)k, (long long)n, (long long)bs,
              batch_type == SBAMD_BATCH_INTERLEAVED ? "interleaved" : "strided", rep);
  for (const Outcome& o : outcomes) {
    if (!o.ok) {
      std::printf("  %-32s  not run: %s\n", o.cand.name.c_str(), o.note.c_str());
      continue;
    }
    std::printf("  %-32s %10.4f ms %12.1f GFLOP/s  %s%s\n", o.cand.name.c_str(), o.ms, o.gflops, o.note.c_str(),
                o.same_bits ? "" : "  RESULT DIFFERS");
    if (g_csv)
      // the candidate as pinned (kernel, tile, depth, fused), then what the library reports it ran
      std::fprintf(g_csv, "%s,%c,%c,%lld,%lld,%lld,%lld,%d,%d,%d,%.6f,%d,%d,%d,%d,%lld,%d\n", sizeof(T) == 4 ? "float" : "double", ta, tb,
                   (long long)m, (long long)k, (long long)n, (long long)bs, o.cand.kernel, (o.cand.param >> 16) & 0xff,
                   o.cand.param & 0xffff, o.ms, (o.cand.param >> 24) & 1, o.ran_kernel, o.ran_tile_m, o.ran_tile_n,
                   (long long)o.ran_depth, o.ran_fused);
  }
  for (const Outcome& o : outcomes)
    if (o.ok && !o.same_bits) throw std::runtime_error("kernel families disagree on this shape");
}

}  // namespace

int main(int argc, char** argv) {
  std::vector<std::string> pos;
  std::string dtype = "float", trans = "all";
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a.compare(0, 8, "--dtype=") == 0) dtype = a.substr(8);
    else if (a.compare(0, 8, "--trans=") == 0) trans = a.substr(8);
    else if (a.compare(0, 6, "--csv=") == 0) g_csv = std::fopen(a.substr(6).c_str(), "a");
    else pos.push_back(a);
  }
  if (pos.size() < 5 || pos.size() > 6) {
    std::fprintf(stderr, "usage: tune_gemm M K N bs rep [strided|interleaved] [--dtype=float|double] [--trans=nn|nt|tn|tt|all]\n");
    return 2;
  }
  try {
    const int64_t m = std::stoll(pos[0]), k = std::stoll(pos[1]), n = std::stoll(pos[2]), bs = std::stoll(pos[3]);
    const int rep = std::max(1, std::stoi(pos[4]));
    int batch_type = SBAMD_BATCH_STRIDED;
    if (pos.size() == 6) {
      if (pos[5] == "interleaved") batch_type = SBAMD_BATCH_INTERLEAVED;
      else if (pos[5] != "strided") throw std::runtime_error("batch_type must be strided or interleaved");
    }
    std::vector<std::string> combos = {"nn", "tn", "nt", "tt"};
    if (trans != "all") combos = {trans};
    sbamd_handle_t h = nullptr;
    check(sbamd_create(&h, 0, nullptr), nullptr, "sbamd_create");
    for (const std::string& t : combos) {
      if (t.size() != 2) throw std::runtime_error("bad --trans");
      if (dtype == "double") tune<double>(h, t[0], t[1], m, k, n, bs, rep, batch_type);
      else tune<float>(h, t[0], t[1], m, k, n, bs, rep, batch_type);
    }
    sbamd_destroy(h);
    if (g_csv) std::fclose(g_csv);
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "tune_gemm: %s\n", e.what());
    return 1;
  }
}
