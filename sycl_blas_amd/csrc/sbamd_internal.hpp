// Internal declarations shared by the C-ABI translation unit and the kernel
// translation units (one per storage type, so they compile in parallel).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sbamd_gemm.h"

namespace sbamd {

struct Handle {
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  int compute_units = 0;
  int kernel_override = SBAMD_KERNEL_AUTO;
  int kernel_param = 0;
  int last_kernel = -1;
  int64_t last_depth = 1;    // split-K cut of the last call
  int last_tile_m = 0, last_tile_n = 0;  // macro tile of the last fp32 / fp64 call's matrix-pipe kernel (0: none)
  bool last_fused = false;   // both K halves inside one workgroup
  int64_t last_k_chunk = 0;
  int64_t last_trsm_diag = 0;   // how the last _trsm call was carried out
  int64_t last_trsm_panel = 0;
  // Scratch for the split-K cube (the reference draws it from
  // Temp_Mem_Pool / acquire_temp_mem, src/sb_handle/portblas_handle.hpp:42-101).
  // Grown on demand, reused across calls, freed with the handle.  Work on one
  // stream is ordered, so consecutive calls may share it.
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  // Second scratch area for the GEMM callers: the expanded symmetric matrix of
  // _symm, the inverted diagonal blocks and the X copy of _trsm.  Kept apart from
  // the cube because those drivers issue GEMM calls that may use the cube.
  void* aux = nullptr;
  size_t aux_bytes = 0;
  // Buffers that were outgrown while work queued on the stream might still use
  // them: each waits here with an event recorded at the moment it was replaced
  // and is freed once that event has completed (no wait in the middle of a stream).
  struct Retired {
    void* ptr;
    hipEvent_t done;
  };
  std::vector<Retired> retired;
  // This device's address of the fixed array that the entry point which allocates
  // nothing keeps its split-K cube in (looked up once, in sbamd_create).
  void* fixed_scratch = nullptr;
  std::mutex mu;
  std::string last_error;
};

// Makes `dev` the current device for the lifetime of the object and restores
// the previous one afterwards.
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// A kernel attribute (the dynamic LDS limit) belongs to one device's copy of the
// kernel: remember per kernel which devices have had it set, so a process that
// holds handles on several GPUs sets it on each.
struct PerDeviceOnce {
  std::atomic<uint64_t> done{0};  // bit d: set on device d
  template <typename F>
  hipError_t run(F&& set_attribute) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 64) return set_attribute();
    const uint64_t bit = uint64_t(1) << dev;
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = set_attribute();
    if (e != hipSuccess) return e;
    done.fetch_or(bit, std::memory_order_release);
    return hipSuccess;
  }
};

}  // namespace sbamd

// The opaque handle of the C ABI is a thin shell around sbamd::Handle.
struct sbamd_handle_s {
  sbamd::Handle impl;
};

namespace sbamd {

// Everything the front end has already validated and normalised.
template <typename TIn, typename TOut>
struct Problem {
  bool ta, tb;  // op(A) / op(B) transposed
  int64_t m, n, k;
  TOut alpha, beta;
  const TIn* a;
  int64_t lda, stride_a;
  const TIn* b;
  int64_t ldb, stride_b;
  TOut* c;
  int64_t ldc, stride_c;
  int64_t batch;
  bool beta_zero;
  bool interleaved;  // batch index contiguous (gemm_batch_type_t::interleaved)
  // Caller-provided split-K cube (the entry point that allocates nothing): when set, the
  // partial products go there instead of the handle's on-demand buffer, the planner only
  // considers cuts whose cube fits, and a pinned cut that does not fit fails with
  // hipErrorOutOfMemory before anything is launched.  Carried per call, so other calls on
  // the same handle are not affected.
  void* cube = nullptr;
  size_t cube_bytes = 0;
};

// Per-type back ends (defined in gemm_f32.hip, gemm_f64.hip, gemm_half.hip).
// Each picks a kernel family (or obeys h->kernel_override), launches on
// h->stream, records h->last_kernel and returns a HIP error code.
hipError_t run_gemm_f32(Handle* h, const Problem<float, float>& p);
hipError_t run_gemm_f64(Handle* h, const Problem<double, double>& p);

// 16-bit storage (bit patterns): half -> half, half -> float, bfloat16 -> float.
// alpha / beta arrive as float (already rounded to the output type by the caller).
hipError_t run_gemm_h(Handle* h, const Problem<uint16_t, uint16_t>& p, float alpha, float beta);
hipError_t run_gemm_hs(Handle* h, const Problem<uint16_t, float>& p, float alpha, float beta);
hipError_t run_gemm_bs(Handle* h, const Problem<uint16_t, float>& p, float alpha, float beta);
hipError_t run_scale_f16(Handle* h, uint16_t* c, int64_t rows, int64_t cols, int64_t ld, int64_t stride,
                         int64_t batch, float beta, bool assign_zero);

// C = beta * C over `batch` windows of rows x cols (alpha == 0 path).
hipError_t run_scale_f32(Handle* h, float* c, int64_t rows, int64_t cols, int64_t ld,
                         int64_t stride, int64_t batch, float beta, bool assign_zero);
hipError_t run_scale_f64(Handle* h, double* c, int64_t rows, int64_t cols, int64_t ld,
                         int64_t stride, int64_t batch, double beta, bool assign_zero);

// Device-side uniform fill (dtype codes as in sbamd_fill_uniform).
hipError_t run_fill_uniform(Handle* h, void* dst, int64_t count, int dtype, uint64_t seed,
                            double lo, double hi);

// Grow the handle's scratch buffer (split-K cube) / aux buffer to at least `bytes`.
hipError_t ensure_scratch(Handle* h, size_t bytes);
hipError_t ensure_aux(Handle* h, size_t bytes);

}  // namespace sbamd
