// Multi-GPU layer of the GEMM engine (include/sbamd_dist.h): one process per GPU, the GPUs of
// one node.  The reference has no multi-device code (one sycl::queue per SB_Handle,
// include/sb_handle/portblas_handle.h:46-60), so everything here is new design, made for the
// xGMI full mesh of an MI355X node:
//
//   * a rank's output goes straight into the memory of each peer (buffers exchanged once as HIP
//     IPC handles), written by ONE copy kernel whose workgroups are dealt over the peers, so all
//     seven links carry data at once -- no ring, no staging buffer, no re-layout;
//   * ranks order themselves with flag words in uncached device memory: rank r stores an
//     increasing epoch number into slot r of every peer's flag array (system-scope release) and
//     a one-wavefront kernel waits until every slot of its own array has reached the epoch
//     (bounded: a peer that never arrives sets a status word instead of hanging the GPU);
//   * the multiplications run through the single-GPU entry points on the handle's stream, the
//     traffic on a second stream behind events, so band i travels while band i + 1 is multiplied;
//   * RCCL (ncclAllGather / ncclReduceScatter, loaded on demand) is the second transport.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "../../include/sbamd_dist.h"
#include "sbamd_internal.hpp"

namespace {

constexpr int MAXR = SBAMD_DIST_MAX_RANKS;
constexpr int MAXB = SBAMD_DIST_MAX_BANDS;

// One device pointer per rank, passed to kernels by value.
struct PeerTable {
  void* p[MAXR];
};

// What a rank tells its peers about a buffer it shares.
struct ShareRecord {
  hipIpcMemHandle_t handle;
  uint64_t offset;  // of the shared range inside the allocation the handle names
  uint64_t bytes;
};

// A peer's allocation mapped into this process.  One allocation may back several shared
// ranges (a caching allocator hands out pieces of big segments) and may be opened only once.
struct Mapping {
  void* base;
  int refs;
};
std::mutex g_map_mu;
std::map<std::string, Mapping> g_mappings;

hipError_t open_peer(const ShareRecord& rec, void** out, std::string* key_out) {
  std::lock_guard<std::mutex> lock(g_map_mu);
  const std::string key(reinterpret_cast<const char*>(&rec.handle), sizeof(rec.handle));
  auto it = g_mappings.find(key);
  if (it == g_mappings.end()) {
    void* base = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&base, rec.handle, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return e;
    it = g_mappings.emplace(key, Mapping{base, 0}).first;
  }
  it->second.refs++;
  *out = static_cast<char*>(it->second.base) + rec.offset;
  *key_out = key;
  return hipSuccess;
}

void close_peer(const std::string& key) {
  std::lock_guard<std::mutex> lock(g_map_mu);
  auto it = g_mappings.find(key);
  if (it == g_mappings.end()) return;
  if (--it->second.refs <= 0) {
    (void)hipIpcCloseMemHandle(it->second.base);
    g_mappings.erase(it);
  }
}

struct Window {
  void* local = nullptr;
  size_t bytes = 0;
  void* peer[MAXR] = {};
  std::string keys[MAXR];
  bool live = false;
};

// librccl, resolved at run time so that single-GPU users never load it.
struct Rccl {
  void* dl = nullptr;
  ncclComm_t comm = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

template <typename F>
bool resolve(void* dl, const char* name, F* out) {
  *out = reinterpret_cast<F>(dlsym(dl, name));
  return *out != nullptr;
}

}  // namespace

struct sbamd_comm_s {
  sbamd_handle_t h = nullptr;
  int rank = 0, world = 1, transports = 0;
  bool prefer_rccl = false;
  sbamd_allgather_fn ag = nullptr;
  void* ctx = nullptr;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_compute = nullptr, ev_comm = nullptr;
  // mesh
  uint64_t* flags = nullptr;  // [world] epochs this rank has been told, slot p written by rank p
  PeerTable peer_flags = {};
  std::string flag_keys[MAXR];
  uint64_t epoch = 0;
  uint32_t* status = nullptr;  // device word: bit 0 = a wait timed out
  int64_t timeout_ticks = 0;
  int push_blocks = 16;
  bool fault_mute = false;  // SBAMD_DIST_FAULT=mute: this rank never signals (fault injection for the callers' fallback paths)
  void* ws = nullptr;
  size_t ws_bytes = 0;
  PeerTable peer_ws = {};
  std::string ws_keys[MAXR];
  std::vector<Window> windows;
  Rccl rccl;
  // the multiplications of the last gathered call
  int last_bands = 0;
  int64_t last_c0[MAXB] = {}, last_c1[MAXB] = {}, last_depth[MAXB] = {}, last_chunk[MAXB] = {};
};

namespace {

using sbamd::DeviceGuard;

int hip_fail(sbamd_comm_t c, hipError_t e, const char* what) {
  c->h->impl.last_error = std::string(what) + ": " + hipGetErrorString(e);
  return e == hipErrorOutOfMemory ? SBAMD_ALLOC_FAILED : SBAMD_HIP_ERROR;
}

int rccl_fail(sbamd_comm_t c, ncclResult_t r, const char* what) {
  c->h->impl.last_error = std::string(what) + ": " + (c->rccl.GetErrorString ? c->rccl.GetErrorString(r) : "RCCL error");
  return SBAMD_RCCL_ERROR;
}

// ------------------------------------------------------------------ kernels

// Rank `rank` tells every peer that it has reached `epoch`.  The fence makes everything this rank
// wrote before (its own memory, and the peers' through the mesh) visible system-wide first.
__global__ void mesh_signal_kernel(PeerTable flags, int rank, int world, uint64_t epoch) {
  const int p = threadIdx.x;
  __threadfence_system();
  if (p < world && p != rank) {
    uint64_t* slot = static_cast<uint64_t*>(flags.p[p]) + rank;
    __hip_atomic_store(slot, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Wait until every peer has signalled `epoch` (epochs only grow, so a peer that is already one
// call ahead also satisfies it).  Bounded: after `timeout` ticks of the constant-rate clock the
// lane gives up and sets bit 0 of *status -- every lane reaches the end of the kernel.
__global__ void mesh_wait_kernel(const uint64_t* flags, int rank, int world, uint64_t epoch, int64_t timeout,
                                 uint32_t* status) {
  const int p = threadIdx.x;
  if (p < world && p != rank) {
    const int64_t t0 = wall_clock64();
    while (__hip_atomic_load(flags + p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
      if (wall_clock64() - t0 > timeout) {
        atomicOr(status, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(32);
    }
  }
  __threadfence_system();
}

// Copy a rows x cols piece of a column-major matrix (V-sized units along a column, ld_bytes between
// columns) from this rank's memory to the same place in every peer's.  blockIdx.y picks the peer,
// starting at rank + 1 so that at any moment the ranks write to different destinations; each thread
// keeps four independent transfers in flight.
template <typename V>
__global__ void __launch_bounds__(256)
    mesh_push_kernel(const char* src, PeerTable dst, int rank, int world, int64_t rows_v, int64_t cols, int64_t ld_bytes) {
  const int peer = (rank + 1 + int(blockIdx.y)) % world;
  char* d = static_cast<char*>(dst.p[peer]);
  const int64_t total = rows_v * cols;
  const int64_t stride = int64_t(gridDim.x) * blockDim.x;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += 4 * stride) {
    V v[4];
    int64_t off[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t idx = i + u * stride;
      const int64_t col = idx / rows_v;
      off[u] = idx < total ? col * ld_bytes + (idx - col * rows_v) * int64_t(sizeof(V)) : -1;
      if (off[u] >= 0) v[u] = *reinterpret_cast<const V*>(src + off[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (off[u] >= 0) *reinterpret_cast<V*>(d + off[u]) = v[u];
  }
  __threadfence_system();
}

// Sum the slab [c0, c0 + cols) of every rank's partial product (packed, leading dimension m) in
// rank order, add beta * C once, store into C.  The order of the additions is the same on every
// rank, and the products and sums round separately (the file is built with -ffp-contract=off):
// the result is that of a K cut of `world` slabs on one GPU.
template <typename T>
__global__ void __launch_bounds__(256)
    mesh_reduce_kernel(PeerTable partials, int world, int64_t m, int64_t c0, int64_t cols, T* c, int64_t ldc, T beta,
                       int beta_zero) {
  const int64_t total = m * cols;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t col = c0 + i / m, row = i % m;
    const int64_t off = row + col * m;
    T s = static_cast<const T*>(partials.p[0])[off];
    for (int r = 1; r < world; ++r) s = s + static_cast<const T*>(partials.p[r])[off];
    T* out = c + row + col * ldc;
    if (!beta_zero) {
      const T bc = beta * *out;
      s = s + bc;
    }
    *out = s;
  }
}

// RCCL staging: a rows x cols piece of a column-major matrix <-> a packed slot (leading dimension slot_ld).
template <typename V>
__global__ void __launch_bounds__(256) pack_kernel(const V* src, int64_t ld, V* slot, int64_t slot_ld, int64_t rows, int64_t cols) {
  const int64_t total = rows * cols;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t col = i / rows, row = i % rows;
    slot[row + col * slot_ld] = src[row + col * ld];
  }
}
template <typename V>
__global__ void __launch_bounds__(256) unpack_kernel(const V* slot, int64_t slot_ld, V* dst, int64_t ld, int64_t rows, int64_t cols) {
  const int64_t total = rows * cols;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t col = i / rows, row = i % rows;
    dst[row + col * ld] = slot[row + col * slot_ld];
  }
}

// C slab <- reduced + beta * C (epilogue of the RCCL reduce-scatter route)
template <typename T>
__global__ void __launch_bounds__(256) slab_epilogue_kernel(const T* reduced, int64_t m, int64_t cols, T* c, int64_t ldc, T beta, int beta_zero) {
  const int64_t total = m * cols;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t col = i / m, row = i % m;
    T s = reduced[i];
    T* out = c + row + col * ldc;
    if (!beta_zero) {
      const T bc = beta * *out;
      s = s + bc;
    }
    *out = s;
  }
}

// ------------------------------------------------------------- host helpers

// Share `bytes` at `ptr` with every peer: exchange IPC handles through the caller's all-gather and
// map the peers' ranges.  peers[rank] = ptr.
int share_buffer(sbamd_comm_t c, void* ptr, size_t bytes, void** peers, std::string* keys) {
  ShareRecord mine;
  std::memset(&mine, 0, sizeof(mine));
  void* base = nullptr;
  size_t span = 0;
  hipError_t e = hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t*>(&base), &span, ptr);
  if (e != hipSuccess) return hip_fail(c, e, "hipMemGetAddressRange");
  if ((e = hipIpcGetMemHandle(&mine.handle, base)) != hipSuccess) return hip_fail(c, e, "hipIpcGetMemHandle");
  mine.offset = uint64_t(static_cast<char*>(ptr) - static_cast<char*>(base));
  mine.bytes = bytes;
  std::vector<ShareRecord> all(size_t(c->world));
  if (c->ag(c->ctx, &mine, all.data(), sizeof(ShareRecord)) != 0) return SBAMD_DIST_BOOTSTRAP;
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank) {
      peers[p] = ptr;
      continue;
    }
    if (all[size_t(p)].bytes != bytes) {
      c->h->impl.last_error = "the ranks passed buffers of different sizes to a collective registration";
      return SBAMD_INVALID_VALUE;
    }
    if ((e = open_peer(all[size_t(p)], &peers[p], &keys[p])) != hipSuccess) return hip_fail(c, e, "hipIpcOpenMemHandle");
  }
  return SBAMD_OK;
}

// SBAMD_DIST_PREFER=rccl: take the RCCL route wherever the communicator has it -- also in a one-rank
// group, where nothing needs to travel (a self-test of the collective calls, streams and events on one GPU).
bool rccl_preferred(sbamd_comm_t c) { return c->prefer_rccl && (c->transports & SBAMD_DIST_RCCL); }

bool use_mesh(sbamd_comm_t c, int window) {
  return (c->transports & SBAMD_DIST_MESH) && window >= 0 && !rccl_preferred(c);
}

int check_window(sbamd_comm_t c, int window, const void* ptr, size_t bytes) {
  if (window < 0 || size_t(window) >= c->windows.size() || !c->windows[size_t(window)].live) return SBAMD_INVALID_VALUE;
  const Window& w = c->windows[size_t(window)];
  const char* lo = static_cast<const char*>(w.local);
  const char* p = static_cast<const char*>(ptr);
  if (p < lo || p + bytes > lo + w.bytes) {
    c->h->impl.last_error = "the matrix does not lie inside the registered window";
    return SBAMD_INVALID_VALUE;
  }
  return SBAMD_OK;
}

// signal + wait on `stream`: a barrier over all ranks in stream order
int mesh_barrier(sbamd_comm_t c, hipStream_t stream) {
  if (c->world == 1) return SBAMD_OK;
  const uint64_t e = ++c->epoch;
  if (!c->fault_mute) hipLaunchKernelGGL(mesh_signal_kernel, dim3(1), dim3(64), 0, stream, c->peer_flags, c->rank, c->world, e);
  hipLaunchKernelGGL(mesh_wait_kernel, dim3(1), dim3(64), 0, stream, c->flags, c->rank, c->world, e, c->timeout_ticks,
                     c->status);
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? SBAMD_OK : hip_fail(c, err, "mesh barrier launch");
}

// Push rows x cols elements at (row0, col0) of the matrix at `c_local` (window `w`) to every peer.
int mesh_push(sbamd_comm_t c, const Window& w, const void* c_local, int64_t ldc, int elem, int64_t row0, int64_t col0,
              int64_t rows, int64_t cols, hipStream_t stream) {
  if (c->world == 1 || rows <= 0 || cols <= 0) return SBAMD_OK;
  const int64_t origin = (row0 + col0 * ldc) * elem;
  const char* src = static_cast<const char*>(c_local) + origin;
  const int64_t in_window = src - static_cast<const char*>(w.local);
  PeerTable dst;
  for (int p = 0; p < MAXR; ++p) dst.p[p] = p < c->world ? static_cast<char*>(w.peer[p]) + in_window : nullptr;
  const int64_t ld_bytes = ldc * elem;
  // 16-byte units when every column segment starts and ends on one (the same offsets on every rank)
  const bool vec = (in_window % 16 == 0) && (reinterpret_cast<uintptr_t>(w.local) % 16 == 0) && ((rows * elem) % 16 == 0) &&
                   (ld_bytes % 16 == 0);
  bool peers_aligned = vec;
  for (int p = 0; p < c->world && peers_aligned; ++p) peers_aligned = reinterpret_cast<uintptr_t>(w.peer[p]) % 16 == 0;
  const int64_t bytes = rows * cols * elem;
  int blocks = c->push_blocks;
  blocks = int(std::max<int64_t>(1, std::min<int64_t>(blocks, bytes / (256 * 16 * 4) + 1)));
  const dim3 grid(unsigned(blocks), unsigned(c->world - 1));
  if (peers_aligned) {
    hipLaunchKernelGGL((mesh_push_kernel<uint4>), grid, dim3(256), 0, stream, src, dst, c->rank, c->world, rows * elem / 16, cols,
                       ld_bytes);
  } else if (elem == 8) {
    hipLaunchKernelGGL((mesh_push_kernel<uint64_t>), grid, dim3(256), 0, stream, src, dst, c->rank, c->world, rows, cols, ld_bytes);
  } else {
    hipLaunchKernelGGL((mesh_push_kernel<uint32_t>), grid, dim3(256), 0, stream, src, dst, c->rank, c->world, rows, cols, ld_bytes);
  }
  hipError_t err = hipGetLastError();
  return err == hipSuccess ? SBAMD_OK : hip_fail(c, err, "mesh push launch");
}

int load_rccl(sbamd_comm_t c) {
  Rccl& r = c->rccl;
  r.dl = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!r.dl) r.dl = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!r.dl) {
    c->h->impl.last_error = std::string("librccl could not be loaded: ") + dlerror();
    return SBAMD_RCCL_ERROR;
  }
  const bool ok = resolve(r.dl, "ncclGetUniqueId", &r.GetUniqueId) && resolve(r.dl, "ncclCommInitRank", &r.CommInitRank) &&
                  resolve(r.dl, "ncclCommDestroy", &r.CommDestroy) && resolve(r.dl, "ncclAllGather", &r.AllGather) &&
                  resolve(r.dl, "ncclReduceScatter", &r.ReduceScatter) && resolve(r.dl, "ncclAllReduce", &r.AllReduce) &&
                  resolve(r.dl, "ncclGetErrorString", &r.GetErrorString);
  if (!ok) {
    c->h->impl.last_error = "librccl lacks an expected entry point";
    return SBAMD_RCCL_ERROR;
  }
  ncclUniqueId id;
  std::memset(&id, 0, sizeof(id));
  if (c->rank == 0) {
    ncclResult_t rr = r.GetUniqueId(&id);
    if (rr != ncclSuccess) return rccl_fail(c, rr, "ncclGetUniqueId");
  }
  std::vector<ncclUniqueId> all(size_t(c->world));
  if (c->ag(c->ctx, &id, all.data(), sizeof(id)) != 0) return SBAMD_DIST_BOOTSTRAP;
  ncclResult_t rr = r.CommInitRank(&r.comm, c->world, all[0], c->rank);
  if (rr != ncclSuccess) return rccl_fail(c, rr, "ncclCommInitRank");
  return SBAMD_OK;
}

template <typename T>
struct Gemm;
template <>
struct Gemm<float> {
  static constexpr ncclDataType_t nccl = ncclFloat32;
  static int run(sbamd_handle_t h, char ta, char tb, int64_t m, int64_t n, int64_t k, float al, const float* a, int64_t lda,
                 const float* b, int64_t ldb, float be, float* c, int64_t ldc) {
    return sbamd_sgemm(h, ta, tb, m, n, k, al, a, lda, b, ldb, be, c, ldc);
  }
};
template <>
struct Gemm<double> {
  static constexpr ncclDataType_t nccl = ncclFloat64;
  static int run(sbamd_handle_t h, char ta, char tb, int64_t m, int64_t n, int64_t k, double al, const double* a, int64_t lda,
                 const double* b, int64_t ldb, double be, double* c, int64_t ldc) {
    return sbamd_dgemm(h, ta, tb, m, n, k, al, a, lda, b, ldb, be, c, ldc);
  }
};

inline bool is_n(char t) { return t == 'n' || t == 'N'; }

void split_range(int64_t extent, int parts, int index, int64_t* begin, int64_t* end) {
  const int64_t base = extent / parts, rem = extent % parts;
  *begin = index * base + std::min<int64_t>(index, rem);
  *end = *begin + base + (index < rem ? 1 : 0);
}

int band_widths(int64_t cols, int bands, int world, int64_t* widths) {
  if (cols <= 0) return 0;
  if (bands > MAXB) bands = MAXB;
  if (bands > 0) {
    int count = 0;
    for (int b = 0; b < bands; ++b) {
      int64_t b0, b1;
      split_range(cols, bands, b, &b0, &b1);
      if (b1 > b0) widths[count++] = b1 - b0;
    }
    return count;
  }
  if (world <= 1) {
    widths[0] = cols;
    return 1;
  }
  // Halves that shrink towards the end: a band travels while the next one is multiplied, and only the
  // last band's transfer has nothing to hide behind -- so it is made small (an eighth of the columns),
  // while the early bands stay big enough to fill the chip with large tiles.  Widths are multiples of 64.
  int count = 0;
  int64_t left = cols;
  while (count < 3 && left >= 256) {
    int64_t w = (left / 2 + 63) / 64 * 64;
    if (w >= left) break;
    widths[count++] = w;
    left -= w;
  }
  if (left > 0) widths[count++] = left;
  return count;
}

template <typename T>
int gathered_gemm(sbamd_comm_t cm, char transa, char transb, int64_t m, int64_t n, int64_t k, T alpha, const T* a, int64_t lda,
                  const T* b, int64_t ldb, T beta, T* c, int64_t ldc, int window, int bands) {
  if (!cm) return SBAMD_INVALID_HANDLE;
  if (m < 0 || n < 0 || k < 0 || ldc < m) return SBAMD_INVALID_VALUE;
  sbamd::Handle* h = &cm->h->impl;
  DeviceGuard guard(h->device);
  const int world = cm->world, rank = cm->rank;
  const bool mesh = use_mesh(cm, window);
  if (world > 1 && !mesh && !(cm->transports & SBAMD_DIST_RCCL)) return SBAMD_DIST_NO_TRANSPORT;
  cm->last_bands = 0;
  if (m == 0 || n == 0) return SBAMD_OK;
  hipError_t e = hipSuccess;
  int st = SBAMD_OK;
  auto record_band = [&](int64_t c0, int64_t c1) {
    if (cm->last_bands < MAXB) {
      const int i = cm->last_bands++;
      cm->last_c0[i] = c0, cm->last_c1[i] = c1;
      cm->last_depth[i] = h->last_depth, cm->last_chunk[i] = h->last_k_chunk;
    }
  };
  auto multiply = [&](int64_t c0, int64_t c1) {
    const T* bp = is_n(transb) ? b + c0 * ldb : b + c0;
    const int s = Gemm<T>::run(cm->h, transa, transb, m, c1 - c0, k, alpha, a, lda, bp, ldb, beta, c + c0 * ldc, ldc);
    if (s == SBAMD_OK) record_band(c0, c1);
    return s;
  };

  if (world == 1 && !rccl_preferred(cm)) return multiply(0, n);

  if (mesh) {
    if ((st = check_window(cm, window, c, size_t((n - 1) * ldc + m) * sizeof(T))) != SBAMD_OK) return st;
    const Window& w = cm->windows[size_t(window)];
    int64_t c0, c1, widths[MAXB];
    split_range(n, world, rank, &c0, &c1);
    const int nb = band_widths(c1 - c0, bands, world, widths);
    // The traffic stream joins the handle's stream here: what was queued on it before this call (the
    // last readers of C on this rank) comes first.  Then the ranks meet: nobody writes into a peer's C
    // before that peer has entered the call.  The first multiplication does not wait for any of it.
    if ((e = hipEventRecord(cm->ev_compute, h->stream)) != hipSuccess) return hip_fail(cm, e, "event record");
    if ((e = hipStreamWaitEvent(cm->comm_stream, cm->ev_compute, 0)) != hipSuccess) return hip_fail(cm, e, "stream wait");
    if ((st = mesh_barrier(cm, cm->comm_stream)) != SBAMD_OK) return st;
    int64_t col = c0;
    for (int bi = 0; bi < nb; ++bi) {
      if ((st = multiply(col, col + widths[bi])) != SBAMD_OK) return st;
      if ((e = hipEventRecord(cm->ev_compute, h->stream)) != hipSuccess) return hip_fail(cm, e, "event record");
      if ((e = hipStreamWaitEvent(cm->comm_stream, cm->ev_compute, 0)) != hipSuccess) return hip_fail(cm, e, "stream wait");
      if ((st = mesh_push(cm, w, c, ldc, int(sizeof(T)), 0, col, m, widths[bi], cm->comm_stream)) != SBAMD_OK) return st;
      col += widths[bi];
    }
    // all of this rank's columns are on their way; wait for everybody else's
    if ((st = mesh_barrier(cm, cm->comm_stream)) != SBAMD_OK) return st;
    if ((e = hipEventRecord(cm->ev_comm, cm->comm_stream)) != hipSuccess) return hip_fail(cm, e, "event record");
    if ((e = hipStreamWaitEvent(h->stream, cm->ev_comm, 0)) != hipSuccess) return hip_fail(cm, e, "stream wait");
    return SBAMD_OK;
  }

  // RCCL: band i = columns [i*W, (i+1)*W), rank r multiplies the r-th part of it in place and the band's
  // all-gather (in place: the send buffer is the rank's slot of the receive buffer) runs on the second stream
  if (bands <= 0) {
    bands = 1;
    for (int cand = 2; cand <= 4; ++cand)
      if (n % (int64_t(cand) * world) == 0 && n / (int64_t(cand) * world) >= 512) bands = cand;
  }
  if (ldc != m || n % (int64_t(bands) * world) != 0) {
    h->last_error = "the RCCL band gather needs ldc == m and n divisible by bands * ranks";
    return SBAMD_INVALID_VALUE;
  }
  const int64_t width = n / bands, part = width / world;
  for (int bi = 0; bi < bands; ++bi) {
    const int64_t col = bi * width + rank * part;
    if ((st = multiply(col, col + part)) != SBAMD_OK) return st;
    if ((e = hipEventRecord(cm->ev_compute, h->stream)) != hipSuccess) return hip_fail(cm, e, "event record");
    if ((e = hipStreamWaitEvent(cm->comm_stream, cm->ev_compute, 0)) != hipSuccess) return hip_fail(cm, e, "stream wait");
    ncclResult_t rr = cm->rccl.AllGather(c + col * m, c + bi * width * m, size_t(m * part), Gemm<T>::nccl, cm->rccl.comm,
                                         cm->comm_stream);
    if (rr != ncclSuccess) return rccl_fail(cm, rr, "ncclAllGather");
  }
  if ((e = hipEventRecord(cm->ev_comm, cm->comm_stream)) != hipSuccess) return hip_fail(cm, e, "event record");
  if ((e = hipStreamWaitEvent(h->stream, cm->ev_comm, 0)) != hipSuccess) return hip_fail(cm, e, "stream wait");
  return SBAMD_OK;
}

template <typename T>
int splitk_gemm(sbamd_comm_t cm, char transa, char transb, int64_t m, int64_t n, int64_t k, T alpha, const T* a, int64_t lda,
                const T* b, int64_t ldb, T beta, T* c, int64_t ldc, int window, int gather) {
  if (!cm) return SBAMD_INVALID_HANDLE;
  if (m < 0 || n < 0 || k < 0 || ldc < m) return SBAMD_INVALID_VALUE;
  sbamd::Handle* h = &cm->h->impl;
  DeviceGuard guard(h->device);
  const int world = cm->world, rank = cm->rank;
  if (m == 0 || n == 0) return SBAMD_OK;
  if (world == 1 && !rccl_preferred(cm)) return Gemm<T>::run(cm->h, transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc);
  const bool mesh = (cm->transports & SBAMD_DIST_MESH) && !rccl_preferred(cm);
  if (!mesh && !(cm->transports & SBAMD_DIST_RCCL)) return SBAMD_DIST_NO_TRANSPORT;
  const size_t need = size_t(m) * size_t(n) * sizeof(T) + (mesh ? 0 : size_t(m) * size_t(n / world) * sizeof(T));
  if (need > cm->ws_bytes) {
    h->last_error = "the communicator's workspace is too small for a full-size partial product";
    return SBAMD_DIST_NO_TRANSPORT;
  }
  // K range of this rank: slabs of ceil(k / world), the cut of a K-cut launch on one GPU
  const int64_t chunk = std::max<int64_t>(1, (k + world - 1) / world);
  const int64_t k0 = std::min<int64_t>(k, rank * chunk), k1 = std::min<int64_t>(k, k0 + chunk);
  T* part = static_cast<T*>(cm->ws);
  hipError_t e = hipSuccess;
  int st = SBAMD_OK;
  if (mesh) {
    // the peers may still be reading the partial products of the previous call out of this workspace
    if ((st = mesh_barrier(cm, h->stream)) != SBAMD_OK) return st;
  }
  if (k1 > k0) {
    const T* ap = is_n(transa) ? a + k0 * lda : a + k0;
    const T* bp = is_n(transb) ? b + k0 : b + k0 * ldb;
    if ((st = Gemm<T>::run(cm->h, transa, transb, m, n, k1 - k0, alpha, ap, lda, bp, ldb, T(0), part, m)) != SBAMD_OK) return st;
  } else if ((e = hipMemsetAsync(part, 0, size_t(m) * size_t(n) * sizeof(T), h->stream)) != hipSuccess) {
    return hip_fail(cm, e, "memset");
  }
  int64_t c0, c1;
  if (mesh) {
    split_range(n, world, rank, &c0, &c1);
    if ((st = mesh_barrier(cm, h->stream)) != SBAMD_OK) return st;  // every partial product is complete and visible
    if (c1 > c0) {
      const int64_t total = m * (c1 - c0);
      const unsigned grid = unsigned(std::min<int64_t>((total + 255) / 256, 4096));
      hipLaunchKernelGGL((mesh_reduce_kernel<T>), dim3(grid), dim3(256), 0, h->stream, cm->peer_ws, world, m, c0, c1 - c0, c, ldc,
                         beta, beta == T(0) ? 1 : 0);
      if ((e = hipGetLastError()) != hipSuccess) return hip_fail(cm, e, "reduce launch");
    }
    if (gather) {
      if (window < 0) return SBAMD_DIST_NO_TRANSPORT;
      if ((st = check_window(cm, window, c, size_t((n - 1) * ldc + m) * sizeof(T))) != SBAMD_OK) return st;
      if ((st = mesh_push(cm, cm->windows[size_t(window)], c, ldc, int(sizeof(T)), 0, c0, m, c1 - c0, h->stream)) != SBAMD_OK)
        return st;
      if ((st = mesh_barrier(cm, h->stream)) != SBAMD_OK) return st;
    }
    return SBAMD_OK;
  }
  if (n % world != 0) {
    h->last_error = "the RCCL reduce-scatter needs n divisible by the number of ranks";
    return SBAMD_INVALID_VALUE;
  }
  const int64_t slab = n / world;
  T* reduced = part + m * n;
  ncclResult_t rr = cm->rccl.ReduceScatter(part, reduced, size_t(m * slab), Gemm<T>::nccl, ncclSum, cm->rccl.comm, h->stream);
  if (rr != ncclSuccess) return rccl_fail(cm, rr, "ncclReduceScatter");
  const int64_t total = m * slab;
  const unsigned grid = unsigned(std::min<int64_t>((total + 255) / 256, 4096));
  hipLaunchKernelGGL((slab_epilogue_kernel<T>), dim3(grid), dim3(256), 0, h->stream, reduced, m, slab, c + rank * slab * ldc, ldc,
                     beta, beta == T(0) ? 1 : 0);
  if ((e = hipGetLastError()) != hipSuccess) return hip_fail(cm, e, "epilogue launch");
  if (gather) {  // the slabs are whole column panels: one in-place all-gather when C is packed
    if (ldc != m) {
      h->last_error = "gathering the slabs through RCCL needs ldc == m";
      return SBAMD_INVALID_VALUE;
    }
    rr = cm->rccl.AllGather(c + rank * slab * m, c, size_t(m * slab), Gemm<T>::nccl, cm->rccl.comm, h->stream);
    if (rr != ncclSuccess) return rccl_fail(cm, rr, "ncclAllGather");
  }
  return SBAMD_OK;
}

template <typename V>
int gather_blocks_rccl(sbamd_comm_t cm, int pr, int pc, int64_t m, int64_t n, V* c, int64_t ldc) {
  sbamd::Handle* h = &cm->h->impl;
  const int world = cm->world, rank = cm->rank;
  // largest block extents (ragged blocks travel padded)
  int64_t bm_max = 0, bn_max = 0;
  for (int r = 0; r < world; ++r) {
    int64_t r0, r1, c0, c1;
    sbamd_dist_block_of(r, pr, pc, m, n, &r0, &r1, &c0, &c1);
    bm_max = std::max(bm_max, r1 - r0), bn_max = std::max(bn_max, c1 - c0);
  }
  if (bm_max == 0 || bn_max == 0) return SBAMD_OK;
  // column stripes of the blocks, as wide as the workspace allows: [send slot | world receive slots]
  const int64_t stripe = std::min<int64_t>(bn_max, int64_t(cm->ws_bytes / (size_t(world + 1) * size_t(bm_max) * sizeof(V))));
  if (stripe <= 0) {
    h->last_error = "the communicator's workspace is too small to stage one column of a C block";
    return SBAMD_DIST_NO_TRANSPORT;
  }
  V* send = static_cast<V*>(cm->ws);
  V* recv = send + bm_max * stripe;
  int64_t my_r0, my_r1, my_c0, my_c1;
  sbamd_dist_block_of(rank, pr, pc, m, n, &my_r0, &my_r1, &my_c0, &my_c1);
  hipError_t e = hipSuccess;
  for (int64_t s0 = 0; s0 < bn_max; s0 += stripe) {
    const int64_t sw = std::min(stripe, bn_max - s0);
    const int64_t mine = std::max<int64_t>(0, std::min(sw, (my_c1 - my_c0) - s0));
    if (mine > 0 && my_r1 > my_r0) {
      const int64_t total = (my_r1 - my_r0) * mine;
      hipLaunchKernelGGL((pack_kernel<V>), dim3(unsigned(std::min<int64_t>((total + 255) / 256, 4096))), dim3(256), 0, h->stream,
                         c + my_r0 + (my_c0 + s0) * ldc, ldc, send, bm_max, my_r1 - my_r0, mine);
    }
    ncclResult_t rr = cm->rccl.AllGather(send, recv, size_t(bm_max * stripe) * sizeof(V), ncclChar, cm->rccl.comm, h->stream);
    if (rr != ncclSuccess) return rccl_fail(cm, rr, "ncclAllGather");
    for (int r = 0; r < world; ++r) {
      if (r == rank) continue;
      int64_t r0, r1, c0, c1;
      sbamd_dist_block_of(r, pr, pc, m, n, &r0, &r1, &c0, &c1);
      const int64_t theirs = std::max<int64_t>(0, std::min(sw, (c1 - c0) - s0));
      if (theirs <= 0 || r1 <= r0) continue;
      const int64_t total = (r1 - r0) * theirs;
      hipLaunchKernelGGL((unpack_kernel<V>), dim3(unsigned(std::min<int64_t>((total + 255) / 256, 4096))), dim3(256), 0, h->stream,
                         recv + int64_t(r) * bm_max * stripe, bm_max, c + r0 + (c0 + s0) * ldc, ldc, r1 - r0, theirs);
    }
    if ((e = hipGetLastError()) != hipSuccess) return hip_fail(cm, e, "pack / unpack launch");
  }
  return SBAMD_OK;
}

}  // namespace

extern "C" {

void sbamd_dist_grid(int world, int* pr, int* pc) {
  int r = 1;
  switch (world) {
    case 1: r = 1; break;
    case 2: r = 1; break;
    case 4: r = 2; break;
    case 8: r = 2; break;
    default:
      for (r = 1; (r + 1) * (r + 1) <= world; ++r) {}
      while (r > 1 && world % r) --r;
  }
  if (pr) *pr = r;
  if (pc) *pc = world / r;
}

void sbamd_dist_split_range(int64_t extent, int parts, int index, int64_t* begin, int64_t* end) {
  int64_t b = 0, e = 0;
  if (parts > 0 && index >= 0 && index < parts && extent >= 0) split_range(extent, parts, index, &b, &e);
  if (begin) *begin = b;
  if (end) *end = e;
}

int sbamd_dist_block_of(int rank, int pr, int pc, int64_t m, int64_t n, int64_t* r0, int64_t* r1, int64_t* c0, int64_t* c1) {
  if (pr <= 0 || pc <= 0 || rank < 0 || rank >= pr * pc || m < 0 || n < 0 || !r0 || !r1 || !c0 || !c1) return SBAMD_INVALID_VALUE;
  split_range(m, pr, rank / pc, r0, r1);
  split_range(n, pc, rank % pc, c0, c1);
  return SBAMD_OK;
}

int sbamd_dist_band_widths(int64_t cols, int bands, int world, int64_t* widths) {
  if (!widths || cols < 0 || bands < 0) return 0;
  return band_widths(cols, bands, world, widths);
}

int sbamd_dist_create(sbamd_comm_t* out, sbamd_handle_t h, int rank, int world, sbamd_allgather_fn allgather, void* ctx,
                      int transports, size_t workspace_bytes) {
  if (!out) return SBAMD_INVALID_VALUE;
  *out = nullptr;
  if (!h) return SBAMD_INVALID_HANDLE;
  if (world < 1 || world > MAXR || rank < 0 || rank >= world || (world > 1 && !allgather) ||
      (transports & ~(SBAMD_DIST_MESH | SBAMD_DIST_RCCL)))
    return SBAMD_INVALID_VALUE;
  sbamd_comm_s* c = new (std::nothrow) sbamd_comm_s;
  if (!c) return SBAMD_ALLOC_FAILED;
  c->h = h;
  c->rank = rank;
  c->world = world;
  c->transports = transports;
  c->ag = allgather;
  c->ctx = ctx;
  const char* prefer = std::getenv("SBAMD_DIST_PREFER");
  c->prefer_rccl = prefer && std::strcmp(prefer, "rccl") == 0;
  const char* tmo = std::getenv("SBAMD_DIST_TIMEOUT_MS");
  const int64_t ms = tmo ? std::max<int64_t>(1, std::atoll(tmo)) : 20000;
  c->timeout_ticks = ms * 100000;  // the constant-rate clock runs at 100 MHz
  const char* fault = std::getenv("SBAMD_DIST_FAULT");
  c->fault_mute = fault && std::strcmp(fault, "mute") == 0;
  const char* pb = std::getenv("SBAMD_DIST_PUSH_BLOCKS");
  if (pb && std::atoi(pb) > 0) c->push_blocks = std::min(std::atoi(pb), 256);
  DeviceGuard guard(h->impl.device);
  int st = SBAMD_OK;
  hipError_t e = hipSuccess;
  auto bail = [&](int status) {
    sbamd_dist_destroy(c);
    return status;
  };
  if ((e = hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking)) != hipSuccess) return bail(hip_fail(c, e, "stream create"));
  if ((e = hipEventCreateWithFlags(&c->ev_compute, hipEventDisableTiming)) != hipSuccess) return bail(hip_fail(c, e, "event create"));
  if ((e = hipEventCreateWithFlags(&c->ev_comm, hipEventDisableTiming)) != hipSuccess) return bail(hip_fail(c, e, "event create"));
  if ((e = hipMalloc(reinterpret_cast<void**>(&c->status), 256)) != hipSuccess) return bail(hip_fail(c, e, "hipMalloc"));
  if ((e = hipMemset(c->status, 0, 256)) != hipSuccess) return bail(hip_fail(c, e, "memset"));
  if (workspace_bytes > 0) {
    if ((e = hipMalloc(&c->ws, workspace_bytes)) != hipSuccess) return bail(hip_fail(c, e, "workspace hipMalloc"));
    c->ws_bytes = workspace_bytes;
  }
  for (int p = 0; p < MAXR; ++p) c->peer_flags.p[p] = nullptr, c->peer_ws.p[p] = nullptr;
  if (transports & SBAMD_DIST_MESH) {
    // flag words: uncached, so that a waiting lane sees a peer's store without any cache maintenance
    const size_t flag_bytes = 4096;
    if ((e = hipExtMallocWithFlags(reinterpret_cast<void**>(&c->flags), flag_bytes, hipDeviceMallocUncached)) != hipSuccess)
      return bail(hip_fail(c, e, "flag allocation"));
    if ((e = hipMemset(c->flags, 0, flag_bytes)) != hipSuccess) return bail(hip_fail(c, e, "memset"));
    if ((e = hipDeviceSynchronize()) != hipSuccess) return bail(hip_fail(c, e, "synchronize"));
    c->peer_flags.p[rank] = c->flags;
    c->peer_ws.p[rank] = c->ws;
    if (world > 1) {
      if ((st = share_buffer(c, c->flags, flag_bytes, c->peer_flags.p, c->flag_keys)) != SBAMD_OK) return bail(st);
      if (c->ws && (st = share_buffer(c, c->ws, c->ws_bytes, c->peer_ws.p, c->ws_keys)) != SBAMD_OK) return bail(st);
    }
  }
  if ((transports & SBAMD_DIST_RCCL) && (st = load_rccl(c)) != SBAMD_OK) return bail(st);
  *out = c;
  return SBAMD_OK;
}

int sbamd_dist_destroy(sbamd_comm_t c) {
  if (!c) return SBAMD_INVALID_HANDLE;
  DeviceGuard guard(c->h->impl.device);
  (void)hipStreamSynchronize(c->h->impl.stream);
  if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
  if (c->rccl.comm && c->rccl.CommDestroy) (void)c->rccl.CommDestroy(c->rccl.comm);
  if (c->rccl.dl) dlclose(c->rccl.dl);
  for (Window& w : c->windows)
    if (w.live)
      for (int p = 0; p < c->world; ++p)
        if (p != c->rank && !w.keys[p].empty()) close_peer(w.keys[p]);
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank) continue;
    if (!c->flag_keys[p].empty()) close_peer(c->flag_keys[p]);
    if (!c->ws_keys[p].empty()) close_peer(c->ws_keys[p]);
  }
  if (c->flags) (void)hipFree(c->flags);
  if (c->ws) (void)hipFree(c->ws);
  if (c->status) (void)hipFree(c->status);
  if (c->ev_compute) (void)hipEventDestroy(c->ev_compute);
  if (c->ev_comm) (void)hipEventDestroy(c->ev_comm);
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  delete c;
  return SBAMD_OK;
}

int sbamd_dist_rank(sbamd_comm_t c, int* rank, int* world, int* transports) {
  if (!c) return SBAMD_INVALID_HANDLE;
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (transports) *transports = c->transports;
  return SBAMD_OK;
}

int sbamd_dist_register(sbamd_comm_t c, void* ptr, size_t bytes, int* window) {
  if (!c) return SBAMD_INVALID_HANDLE;
  if (!ptr || bytes == 0 || !window) return SBAMD_INVALID_VALUE;
  if (!(c->transports & SBAMD_DIST_MESH)) return SBAMD_DIST_NO_TRANSPORT;
  DeviceGuard guard(c->h->impl.device);
  Window w;
  w.local = ptr;
  w.bytes = bytes;
  w.live = true;
  w.peer[c->rank] = ptr;
  if (c->world > 1) {
    const int st = share_buffer(c, ptr, bytes, w.peer, w.keys);
    if (st != SBAMD_OK) return st;
  }
  c->windows.push_back(w);
  *window = int(c->windows.size()) - 1;
  return SBAMD_OK;
}

int sbamd_dist_unregister(sbamd_comm_t c, int window) {
  if (!c) return SBAMD_INVALID_HANDLE;
  if (window < 0 || size_t(window) >= c->windows.size() || !c->windows[size_t(window)].live) return SBAMD_INVALID_VALUE;
  DeviceGuard guard(c->h->impl.device);
  // nobody may still be writing through the mapping
  (void)hipStreamSynchronize(c->h->impl.stream);
  (void)hipStreamSynchronize(c->comm_stream);
  Window& w = c->windows[size_t(window)];
  for (int p = 0; p < c->world; ++p)
    if (p != c->rank && !w.keys[p].empty()) close_peer(w.keys[p]);
  w = Window();
  return SBAMD_OK;
}

int sbamd_dist_barrier(sbamd_comm_t c) {
  if (!c) return SBAMD_INVALID_HANDLE;
  if (c->world == 1 && !rccl_preferred(c)) return SBAMD_OK;
  DeviceGuard guard(c->h->impl.device);
  if ((c->transports & SBAMD_DIST_MESH) && !rccl_preferred(c)) return mesh_barrier(c, c->h->impl.stream);
  if (c->transports & SBAMD_DIST_RCCL) {
    float* word = reinterpret_cast<float*>(c->status) + 16;  // scratch inside the status block
    ncclResult_t rr = c->rccl.AllReduce(word, word, 1, ncclFloat32, ncclSum, c->rccl.comm, c->h->impl.stream);
    return rr == ncclSuccess ? SBAMD_OK : rccl_fail(c, rr, "ncclAllReduce");
  }
  return SBAMD_DIST_NO_TRANSPORT;
}

int sbamd_dist_status(sbamd_comm_t c) {
  if (!c) return SBAMD_INVALID_HANDLE;
  DeviceGuard guard(c->h->impl.device);
  uint32_t word = 0;
  hipError_t e = hipStreamSynchronize(c->comm_stream);
  if (e == hipSuccess) e = hipMemcpyAsync(&word, c->status, sizeof(word), hipMemcpyDeviceToHost, c->h->impl.stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->h->impl.stream);
  if (e != hipSuccess) return hip_fail(c, e, "status read");
  if (word & 1u) {
    c->h->impl.last_error = "a peer did not reach a barrier of the mesh transport in time";
    return SBAMD_DIST_TIMEOUT;
  }
  return SBAMD_OK;
}

#define SBAMD_DIST_BLOCK(NAME, TIN, TOUT, TSCAL, CALL)                                                                        \
  int NAME(sbamd_comm_t c, int pr, int pc, char transa, char transb, int64_t m, int64_t n, int64_t k, TSCAL alpha,            \
           const TIN* a, int64_t lda, const TIN* b, int64_t ldb, TSCAL beta, TOUT* cm, int64_t ldc) {                          \
    if (!c) return SBAMD_INVALID_HANDLE;                                                                                      \
    int64_t r0, r1, c0, c1;                                                                                                   \
    if (pr * pc != c->world || sbamd_dist_block_of(c->rank, pr, pc, m, n, &r0, &r1, &c0, &c1) != SBAMD_OK)                    \
      return SBAMD_INVALID_VALUE;                                                                                             \
    if (r1 == r0 || c1 == c0) return SBAMD_OK;                                                                                \
    /* op(A) row r0 starts at A + r0 ('n') or A + r0*lda; op(B) column c0 at B + c0*ldb ('n') or B + c0 */                     \
    const TIN* ap = is_n(transa) ? a + r0 : a + r0 * lda;                                                                     \
    const TIN* bp = is_n(transb) ? b + c0 * ldb : b + c0;                                                                     \
    return CALL(c->h, transa, transb, r1 - r0, c1 - c0, k, alpha, ap, lda, bp, ldb, beta, cm + r0 + c0 * ldc, ldc);           \
  }
SBAMD_DIST_BLOCK(sbamd_dist_block_sgemm, float, float, float, sbamd_sgemm)
SBAMD_DIST_BLOCK(sbamd_dist_block_dgemm, double, double, double, sbamd_dgemm)
SBAMD_DIST_BLOCK(sbamd_dist_block_bsgemm, uint16_t, float, float, sbamd_bsgemm)

int sbamd_dist_all_gather_c(sbamd_comm_t c, int pr, int pc, int64_t m, int64_t n, void* cm, int64_t ldc, int elem_bytes,
                            int window) {
  if (!c) return SBAMD_INVALID_HANDLE;
  if (pr * pc != c->world || m < 0 || n < 0 || ldc < m || !cm || (elem_bytes != 4 && elem_bytes != 8)) return SBAMD_INVALID_VALUE;
  if ((c->world == 1 && !rccl_preferred(c)) || m == 0 || n == 0) return SBAMD_OK;
  DeviceGuard guard(c->h->impl.device);
  if (use_mesh(c, window)) {
    int st = check_window(c, window, cm, size_t((n - 1) * ldc + m) * size_t(elem_bytes));
    if (st != SBAMD_OK) return st;
    int64_t r0, r1, c0, c1;
    sbamd_dist_block_of(c->rank, pr, pc, m, n, &r0, &r1, &c0, &c1);
    hipStream_t s = c->h->impl.stream;
    if ((st = mesh_barrier(c, s)) != SBAMD_OK) return st;  // every peer's C is ready to be written into
    if ((st = mesh_push(c, c->windows[size_t(window)], cm, ldc, elem_bytes, r0, c0, r1 - r0, c1 - c0, s)) != SBAMD_OK) return st;
    return mesh_barrier(c, s);  // every block has arrived
  }
  if (!(c->transports & SBAMD_DIST_RCCL)) return SBAMD_DIST_NO_TRANSPORT;
  return elem_bytes == 4 ? gather_blocks_rccl<uint32_t>(c, pr, pc, m, n, static_cast<uint32_t*>(cm), ldc)
                         : gather_blocks_rccl<uint64_t>(c, pr, pc, m, n, static_cast<uint64_t*>(cm), ldc);
}

int sbamd_dist_gathered_sgemm(sbamd_comm_t c, char transa, char transb, int64_t m, int64_t n, int64_t k, float alpha,
                              const float* a, int64_t lda, const float* b, int64_t ldb, float beta, float* cm, int64_t ldc,
                              int window, int bands) {
  return gathered_gemm<float>(c, transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, cm, ldc, window, bands);
}
int sbamd_dist_gathered_dgemm(sbamd_comm_t c, char transa, char transb, int64_t m, int64_t n, int64_t k, double alpha,
                              const double* a, int64_t lda, const double* b, int64_t ldb, double beta, double* cm, int64_t ldc,
                              int window, int bands) {
  return gathered_gemm<double>(c, transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, cm, ldc, window, bands);
}

int sbamd_dist_last_bands(sbamd_comm_t c, int64_t* c0, int64_t* c1, int64_t* depth, int64_t* k_chunk) {
  if (!c || !c0 || !c1 || !depth || !k_chunk) return 0;
  for (int i = 0; i < c->last_bands; ++i) c0[i] = c->last_c0[i], c1[i] = c->last_c1[i], depth[i] = c->last_depth[i], k_chunk[i] = c->last_chunk[i];
  return c->last_bands;
}

int sbamd_dist_splitk_sgemm(sbamd_comm_t c, char transa, char transb, int64_t m, int64_t n, int64_t k, float alpha,
                            const float* a, int64_t lda, const float* b, int64_t ldb, float beta, float* cm, int64_t ldc,
                            int window, int gather) {
  return splitk_gemm<float>(c, transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, cm, ldc, window, gather);
}
int sbamd_dist_splitk_dgemm(sbamd_comm_t c, char transa, char transb, int64_t m, int64_t n, int64_t k, double alpha,
                            const double* a, int64_t lda, const double* b, int64_t ldb, double beta, double* cm, int64_t ldc,
                            int window, int gather) {
  return splitk_gemm<double>(c, transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, cm, ldc, window, gather);
}

}  // extern "C"
