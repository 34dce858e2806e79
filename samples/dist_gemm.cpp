// Multi-GPU GEMM through the C ABI of include/sbamd_dist.h, one process per GPU, and a self-check of
// everything that layer does.  The reference has no counterpart (one sycl::queue per SB_Handle,
// include/sb_handle/portblas_handle.h:46-60); this is what a C++ caller of the single-GPU interface
// (samples/gemm.cpp) writes to put one GEMM on all GPUs of a node.
//
//   dist_gemm_sample [--ranks N] [--share-gpu] [--rccl | --prefer-rccl] [--m M --n N --k K] [--dtype float|double]
//
// --rccl adds the RCCL transport (the mesh stays the first choice), --prefer-rccl makes RCCL the first
// choice -- also in a one-rank group, which exercises every collective call on a one-GPU box.
//
// The launcher forks N rank processes BEFORE anything touches the GPU; the ranks exchange their
// set-up records (IPC handles, the RCCL id) through a shared-memory segment -- the role MPI_Allgather
// or a torch.distributed gloo group plays in a real launcher.  --share-gpu puts every rank on GPU 0
// (a rehearsal of the rank logic, the IPC mappings and the flag protocol on a one-GPU box; RCCL
// refuses two ranks on one GPU, so only the mesh transport runs then).
//
// Checks, every one bit for bit against this rank's own recomputation with the single-GPU entry
// points on the same inputs:
//   1. gathered GEMM (strong scaling, whole C on every rank), automatic and fixed band counts;
//   2. C blocks on the Pr x Pc grid + all-gather of the blocks (config "65536^3 on 2 x 4" in small);
//   3. K cut across the ranks + rank-ordered sum, with and without the gather of the slabs.
#include <pthread.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "sbamd_dist.h"

namespace {

struct Board {  // shared between the rank processes
  pthread_barrier_t barrier;
  int failed;
  char slot[SBAMD_DIST_MAX_RANKS][4096];
};

struct Ctx {
  Board* board;
  int rank, world;
};

int board_allgather(void* ctx_, const void* send, void* recv, size_t bytes) {
  Ctx* ctx = static_cast<Ctx*>(ctx_);
  if (bytes > sizeof(ctx->board->slot[0])) return 1;
  std::memcpy(ctx->board->slot[ctx->rank], send, bytes);
  pthread_barrier_wait(&ctx->board->barrier);
  for (int r = 0; r < ctx->world; ++r) std::memcpy(static_cast<char*>(recv) + size_t(r) * bytes, ctx->board->slot[r], bytes);
  pthread_barrier_wait(&ctx->board->barrier);
  return 0;
}

#define CHECK(call)                                                                                    \
  do {                                                                                                 \
    const int st_ = (call);                                                                            \
    if (st_ != SBAMD_OK) {                                                                             \
      std::fprintf(stderr, "rank %d: %s -> %d (%s) %s\n", rank, #call, st_, sbamd_status_string(st_),  \
                   h ? sbamd_last_error(h) : "");                                                      \
      return 1;                                                                                        \
    }                                                                                                  \
  } while (0)

template <typename T>
struct Api;
template <>
struct Api<float> {
  static constexpr int fill = 0;
  static constexpr auto gemm = sbamd_sgemm;
  static constexpr auto gathered = sbamd_dist_gathered_sgemm;
  static constexpr auto block = sbamd_dist_block_sgemm;
  static constexpr auto splitk = sbamd_dist_splitk_sgemm;
};
template <>
struct Api<double> {
  static constexpr int fill = 1;
  static constexpr auto gemm = sbamd_dgemm;
  static constexpr auto gathered = sbamd_dist_gathered_dgemm;
  static constexpr auto block = sbamd_dist_block_dgemm;
  static constexpr auto splitk = sbamd_dist_splitk_dgemm;
};

// The column ranges every rank multiplies in a gathered GEMM, in the order it multiplies them.
// Mesh: rank r owns split_range(n, world, r), cut into bands by sbamd_dist_band_widths.  RCCL: band i is
// columns [i*W, (i+1)*W) and rank r takes the r-th part of it (the band count is what the call just used).
std::vector<std::pair<int64_t, int64_t>> columns_of_all(int world, int64_t n, int bands, bool rccl_layout, sbamd_comm_t comm) {
  std::vector<std::pair<int64_t, int64_t>> out;
  if (rccl_layout) {
    int64_t c0[SBAMD_DIST_MAX_BANDS], c1[SBAMD_DIST_MAX_BANDS], d[SBAMD_DIST_MAX_BANDS], kc[SBAMD_DIST_MAX_BANDS];
    const int nb = sbamd_dist_last_bands(comm, c0, c1, d, kc);
    const int64_t width = n / nb, part = width / world;
    for (int r = 0; r < world; ++r)
      for (int i = 0; i < nb; ++i) out.emplace_back(i * width + r * part, i * width + (r + 1) * part);
    return out;
  }
  for (int r = 0; r < world; ++r) {
    int64_t c0, c1, widths[SBAMD_DIST_MAX_BANDS];
    sbamd_dist_split_range(n, world, r, &c0, &c1);
    int nb = sbamd_dist_band_widths(c1 - c0, bands, world, widths);
    if (world == 1) nb = 1, widths[0] = n;
    int64_t col = c0;
    for (int i = 0; i < nb; ++i) {
      out.emplace_back(col, col + widths[i]);
      col += widths[i];
    }
  }
  return out;
}

struct Options {
  int ranks = 2;
  bool share_gpu = false, rccl = false, prefer_rccl = false;
  int64_t m = 640, n = 1536, k = 520;
  bool dbl = false;
};

template <typename T>
int run_rank(const Options& opt, Board* board, int rank) {
  using A = Api<T>;
  const int world = opt.ranks;
  Ctx ctx{board, rank, world};
  sbamd_handle_t h = nullptr;
  const int device = opt.share_gpu ? 0 : rank;
  CHECK(sbamd_create(&h, device, nullptr));
  const int64_t m = opt.m, n = opt.n, k = opt.k;
  const size_t ws_bytes = size_t(m) * size_t(n) * sizeof(T) * 2;
  const int transports = SBAMD_DIST_MESH | (opt.rccl ? SBAMD_DIST_RCCL : 0);
  sbamd_comm_t comm = nullptr;
  CHECK(sbamd_dist_create(&comm, h, rank, world, board_allgather, &ctx, transports, ws_bytes));

  // replicated operands: the same seeds on every rank
  const int64_t lda = m + 8, ldb = k, ldc = m;
  T *a = nullptr, *b = nullptr, *c = nullptr, *ref = nullptr, *tmp = nullptr;
  CHECK(sbamd_malloc(h, reinterpret_cast<void**>(&a), size_t(lda * k) * sizeof(T)));
  CHECK(sbamd_malloc(h, reinterpret_cast<void**>(&b), size_t(ldb * n) * sizeof(T)));
  CHECK(sbamd_malloc(h, reinterpret_cast<void**>(&c), size_t(ldc * n) * sizeof(T)));
  CHECK(sbamd_malloc(h, reinterpret_cast<void**>(&ref), size_t(ldc * n) * sizeof(T)));
  CHECK(sbamd_malloc(h, reinterpret_cast<void**>(&tmp), size_t(m * n) * sizeof(T)));
  CHECK(sbamd_fill_uniform(h, a, lda * k, A::fill, 101, -2.0, 5.0));
  CHECK(sbamd_fill_uniform(h, b, ldb * n, A::fill, 202, -2.0, 5.0));
  int window = -1;
  CHECK(sbamd_dist_register(comm, c, size_t(ldc * n) * sizeof(T), &window));

  std::vector<T> got(size_t(ldc * n)), want(size_t(ldc * n));
  auto same = [&](const char* what) {
    if (sbamd_copy_to_host(h, got.data(), c, got.size() * sizeof(T)) != SBAMD_OK ||
        sbamd_copy_to_host(h, want.data(), ref, want.size() * sizeof(T)) != SBAMD_OK || sbamd_wait(h) != SBAMD_OK)
      return false;
    if (std::memcmp(got.data(), want.data(), got.size() * sizeof(T)) != 0) {
      size_t bad = 0;
      while (bad < got.size() && std::memcmp(&got[bad], &want[bad], sizeof(T)) == 0) ++bad;
      std::fprintf(stderr, "rank %d: %s differs at element %zu (column %zu): %g vs %g\n", rank, what, bad, bad / size_t(ldc),
                   double(got[bad]), double(want[bad]));
      return false;
    }
    return true;
  };
  const T alpha = T(1.5), beta = T(0.5);

  // ---- 1. gathered GEMM: automatic bands, then three equal bands, beta != 0 on a known C
  for (int bands : {0, 3}) {
    CHECK(sbamd_fill_uniform(h, c, ldc * n, A::fill, 303, -2.0, 5.0));
    CHECK(sbamd_fill_uniform(h, ref, ldc * n, A::fill, 303, -2.0, 5.0));
    CHECK(A::gathered(comm, 'n', 'n', m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, window, bands));
    // this rank's own recomputation: every rank's columns, band by band, same shapes -> same kernels
    for (const auto& cr : columns_of_all(world, n, bands, opt.prefer_rccl, comm))
      CHECK(A::gemm(h, 'n', 'n', m, cr.second - cr.first, k, alpha, a, lda, b + cr.first * ldb, ldb, beta, ref + cr.first * ldc, ldc));
    if (!same(bands ? "gathered GEMM (3 bands)" : "gathered GEMM (automatic bands)")) return 1;
  }
  // transposed operands through the same schedule (A^T stored k x m with lda = k ... here: A is m x k
  // with lda, so op(A) = A^T needs m' = k; use B transposed instead: op(B) = B^T, B stored n x k)
  {
    T* bt = nullptr;
    const int64_t ldbt = n + 4;
    CHECK(sbamd_malloc(h, reinterpret_cast<void**>(&bt), size_t(ldbt * k) * sizeof(T)));
    CHECK(sbamd_fill_uniform(h, bt, ldbt * k, A::fill, 404, -2.0, 5.0));
    CHECK(A::gathered(comm, 'n', 't', m, n, k, alpha, a, lda, bt, ldbt, T(0), c, ldc, window, 2));
    for (const auto& cr : columns_of_all(world, n, 2, opt.prefer_rccl, comm))
      CHECK(A::gemm(h, 'n', 't', m, cr.second - cr.first, k, alpha, a, lda, bt + cr.first, ldbt, T(0), ref + cr.first * ldc, ldc));
    if (!same("gathered GEMM (B transposed)")) return 1;
    CHECK(sbamd_wait(h));
    CHECK(sbamd_free(h, bt));
  }

  // ---- 2. C blocks on the Pr x Pc grid, then the all-gather of the blocks
  {
    int pr, pc;
    sbamd_dist_grid(world, &pr, &pc);
    CHECK(sbamd_memset(h, c, 0, size_t(ldc * n) * sizeof(T)));
    CHECK(sbamd_memset(h, ref, 0, size_t(ldc * n) * sizeof(T)));
    CHECK(A::block(comm, pr, pc, 'n', 'n', m, n, k, alpha, a, lda, b, ldb, T(0), c, ldc));
    CHECK(sbamd_dist_all_gather_c(comm, pr, pc, m, n, c, ldc, int(sizeof(T)), window));
    for (int r = 0; r < world; ++r) {
      int64_t r0, r1, c0, c1;
      CHECK(sbamd_dist_block_of(r, pr, pc, m, n, &r0, &r1, &c0, &c1));
      CHECK(A::gemm(h, 'n', 'n', r1 - r0, c1 - c0, k, alpha, a + r0, lda, b + c0 * ldb, ldb, T(0), ref + r0 + c0 * ldc, ldc));
    }
    if (!same("C blocks + all-gather")) return 1;
    if (opt.rccl) {  // the same gather through RCCL (packed stripes)
      CHECK(sbamd_memset(h, c, 0, size_t(ldc * n) * sizeof(T)));
      CHECK(A::block(comm, pr, pc, 'n', 'n', m, n, k, alpha, a, lda, b, ldb, T(0), c, ldc));
      CHECK(sbamd_dist_all_gather_c(comm, pr, pc, m, n, c, ldc, int(sizeof(T)), -1));
      if (!same("C blocks + all-gather (RCCL)")) return 1;
    }
  }

  // ---- 3. K cut across the ranks, slabs gathered: every partial recomputed here, summed in rank order
  {
    CHECK(sbamd_fill_uniform(h, c, ldc * n, A::fill, 505, -2.0, 5.0));
    CHECK(sbamd_fill_uniform(h, ref, ldc * n, A::fill, 505, -2.0, 5.0));
    CHECK(sbamd_dist_barrier(comm));
    CHECK(A::splitk(comm, 'n', 'n', m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, window, 1));
    std::vector<T> sum(size_t(m * n)), part(size_t(m * n)), cin(size_t(ldc * n));
    CHECK(sbamd_copy_to_host(h, cin.data(), ref, cin.size() * sizeof(T)));
    const int64_t chunk = (k + world - 1) / world;
    for (int r = 0; r < world; ++r) {
      const int64_t k0 = std::min<int64_t>(k, r * chunk), k1 = std::min<int64_t>(k, k0 + chunk);
      if (k1 > k0) {
        CHECK(A::gemm(h, 'n', 'n', m, n, k1 - k0, alpha, a + k0 * lda, lda, b + k0, ldb, T(0), tmp, m));
      } else {
        CHECK(sbamd_memset(h, tmp, 0, size_t(m * n) * sizeof(T)));
      }
      CHECK(sbamd_copy_to_host(h, part.data(), tmp, part.size() * sizeof(T)));
      CHECK(sbamd_wait(h));
      for (size_t i = 0; i < sum.size(); ++i) {
        volatile T s = r == 0 ? part[i] : T(sum[i] + part[i]);
        sum[i] = s;
      }
    }
    for (int64_t j = 0; j < n; ++j)
      for (int64_t i = 0; i < m; ++i) {
        volatile T bc = beta * cin[size_t(i + j * ldc)];
        volatile T s = sum[size_t(i + j * m)] + bc;
        cin[size_t(i + j * ldc)] = s;
      }
    CHECK(sbamd_copy_to_device(h, ref, cin.data(), cin.size() * sizeof(T)));
    if (!same("K cut across ranks + gather")) return 1;
  }

  CHECK(sbamd_dist_status(comm));
  CHECK(sbamd_wait(h));
  CHECK(sbamd_dist_barrier(comm));
  CHECK(sbamd_wait(h));
  CHECK(sbamd_dist_unregister(comm, window));
  CHECK(sbamd_dist_destroy(comm));
  for (T* p : {a, b, c, ref, tmp}) CHECK(sbamd_free(h, p));
  CHECK(sbamd_destroy(h));
  if (rank == 0)
    std::printf("dist sample ok: %d rank(s)%s, %s %ldx%ldx%ld, transports mesh%s\n", world, opt.share_gpu ? " sharing GPU 0" : "",
                std::is_same<T, float>::value ? "float" : "double", long(m), long(n), long(k), opt.rccl ? " + rccl" : "");
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  Options opt;
  for (int i = 1; i < argc; ++i) {
    const std::string arg = argv[i];
    auto next = [&]() { return i + 1 < argc ? std::atoll(argv[++i]) : 0LL; };
    if (arg == "--ranks") opt.ranks = int(next());
    else if (arg == "--share-gpu") opt.share_gpu = true;
    else if (arg == "--rccl") opt.rccl = true;
    else if (arg == "--prefer-rccl") opt.rccl = opt.prefer_rccl = true;
    else if (arg == "--m") opt.m = next();
    else if (arg == "--n") opt.n = next();
    else if (arg == "--k") opt.k = next();
    else if (arg == "--dtype" && i + 1 < argc) opt.dbl = std::string(argv[++i]) == "double";
    else {
      std::fprintf(stderr, "unknown argument %s\n", arg.c_str());
      return 2;
    }
  }
  if (opt.ranks < 1 || opt.ranks > SBAMD_DIST_MAX_RANKS) return 2;
  if (opt.rccl && opt.share_gpu && opt.ranks > 1) {
    std::fprintf(stderr, "RCCL does not take two ranks on one GPU\n");
    return 2;
  }
  if (opt.prefer_rccl) setenv("SBAMD_DIST_PREFER", "rccl", 1);  // read by sbamd_dist_create
  // shared board, created before the fork; nothing has touched the GPU yet
  Board* board = static_cast<Board*>(mmap(nullptr, sizeof(Board), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0));
  if (board == MAP_FAILED) return 3;
  pthread_barrierattr_t attr;
  pthread_barrierattr_init(&attr);
  pthread_barrierattr_setpshared(&attr, PTHREAD_PROCESS_SHARED);
  pthread_barrier_init(&board->barrier, &attr, unsigned(opt.ranks));
  board->failed = 0;
  std::vector<pid_t> kids;
  for (int r = 0; r < opt.ranks; ++r) {
    const pid_t pid = fork();
    if (pid < 0) return 3;
    if (pid == 0) {
      alarm(240);  // a rank that waits for a dead peer ends by itself
      const int rc = opt.dbl ? run_rank<double>(opt, board, r) : run_rank<float>(opt, board, r);
      std::fflush(nullptr);
      _exit(rc);
    }
    kids.push_back(pid);
  }
  int worst = 0;
  bool killed = false;
  for (size_t done = 0; done < kids.size(); ++done) {
    int status = 0;
    const pid_t pid = wait(&status);
    const int rc = WIFEXITED(status) ? WEXITSTATUS(status) : 128 + (WIFSIGNALED(status) ? WTERMSIG(status) : 0);
    if (rc != 0 && !killed) {
      // a failed rank leaves its peers waiting at the next exchange: end them (exact PIDs of our own children)
      worst = rc;
      killed = true;
      for (pid_t k : kids)
        if (k != pid) kill(k, SIGTERM);
    }
  }
  return worst;
}
