// C ABI of the GEMM engine: handle, events, memory helpers and the GEMM front
// end.  The front end restates, rule for rule, what the reference does between
// its public entry points and the kernel launch
// (src/interface/gemm_interface.hpp:102-240).
#include <cctype>
#include <cstring>
#include <new>

#include "sbamd_internal.hpp"

struct sbamd_event_s {
  hipEvent_t ev;
};

// Split-K cube of the entry point that allocates nothing (sbamd_sgemm_noalloc):
// a fixed array in every GPU's memory, held from the moment the library's code
// is loaded on that GPU until the process ends.  In host code the name is only
// a stand-in for the symbol, never a device address: the address on a given GPU
// comes from hipGetSymbolAddress with that GPU current (fixed_scratch_of below,
// called from sbamd_create) and is what every kernel argument uses.
static __device__ __align__(256) unsigned char scratch[SBAMD_NOALLOC_SCRATCH_BYTES];

namespace sbamd {

// Frees the outgrown buffers whose last possible reader has finished
// (`all`: the stream has been drained, free every one).  Caller holds h->mu.
static void release_retired(Handle* h, bool all) {
  size_t keep = 0;
  for (size_t i = 0; i < h->retired.size(); ++i) {
    Handle::Retired r = h->retired[i];
    if (all || hipEventQuery(r.done) == hipSuccess) {
      (void)hipEventDestroy(r.done);
      (void)hipFree(r.ptr);
    } else {
      h->retired[keep++] = r;
    }
  }
  h->retired.resize(keep);
}

// High-water-mark arena: a request that fits is served from what is there; a
// bigger one gets a new allocation at once, and the old buffer -- which work
// already queued on the stream may still read -- is parked with an event and
// freed later, once that event has completed.  The handle's stream is not waited
// for here (hipMalloc, and the hipFree of a parked buffer whose event has already
// completed, are the only runtime calls that may block).
static hipError_t grow(Handle* h, void** buf, size_t* have, size_t bytes) {
  std::lock_guard<std::mutex> lock(h->mu);
  if (bytes <= *have) return hipSuccess;
  release_retired(h, false);
  const size_t rounded = (bytes + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
  void* fresh = nullptr;
  hipError_t e = hipMalloc(&fresh, rounded);
  if (e != hipSuccess) return e;
  if (*buf) {
    hipEvent_t done;
    e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(done, h->stream);
    if (e != hipSuccess) {
      (void)hipFree(fresh);
      return e;
    }
    h->retired.push_back({*buf, done});
  }
  *buf = fresh;
  *have = rounded;
  return hipSuccess;
}

// The address of `scratch` on the current device, looked up once per device.
static hipError_t fixed_scratch_of(int device, void** out) {
  static std::mutex mu;
  static void* per_device[64] = {};
  std::lock_guard<std::mutex> lock(mu);
  if (device >= 0 && device < 64 && per_device[device]) {
    *out = per_device[device];
    return hipSuccess;
  }
  void* ptr = nullptr;
  hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
  if (e != hipSuccess) return e;
  if (!ptr) return hipErrorInvalidSymbol;
  if (device >= 0 && device < 64) per_device[device] = ptr;
  *out = ptr;
  return hipSuccess;
}

hipError_t ensure_scratch(Handle* h, size_t bytes) { return grow(h, &h->scratch, &h->scratch_bytes, bytes); }
hipError_t ensure_aux(Handle* h, size_t bytes) { return grow(h, &h->aux, &h->aux_bytes, bytes); }

namespace {

int fail_hip(Handle* h, hipError_t e, const char* what) {
  h->last_error = std::string(what) + ": " + hipGetErrorString(e);
  return e == hipErrorOutOfMemory ? SBAMD_ALLOC_FAILED : SBAMD_HIP_ERROR;
}

// One policy per instantiated type pair (src/interface/blas3/gemm.cpp.in:33-137
// plus bfloat16 -> float): storage types, the scalar type alpha/beta are held
// in, and the back end that serves it.
struct PolicyS {
  using in_t = float;
  using out_t = float;
  using scal_t = float;
  static hipError_t run(Handle* h, const Problem<float, float>& p, float, float) { return run_gemm_f32(h, p); }
  static hipError_t scale(Handle* h, float* c, int64_t r, int64_t n, int64_t ld, int64_t st, int64_t b, float beta,
                          bool z) {
    return run_scale_f32(h, c, r, n, ld, st, b, beta, z);
  }
};
struct PolicyD {
  using in_t = double;
  using out_t = double;
  using scal_t = double;
  static hipError_t run(Handle* h, const Problem<double, double>& p, double, double) { return run_gemm_f64(h, p); }
  static hipError_t scale(Handle* h, double* c, int64_t r, int64_t n, int64_t ld, int64_t st, int64_t b,
                          double beta, bool z) {
    return run_scale_f64(h, c, r, n, ld, st, b, beta, z);
  }
};
struct PolicyH {  // half -> half
  using in_t = uint16_t;
  using out_t = uint16_t;
  using scal_t = float;
  static hipError_t run(Handle* h, const Problem<uint16_t, uint16_t>& p, float al, float be) {
    return run_gemm_h(h, p, al, be);
  }
  static hipError_t scale(Handle* h, uint16_t* c, int64_t r, int64_t n, int64_t ld, int64_t st, int64_t b,
                          float beta, bool z) {
    return run_scale_f16(h, c, r, n, ld, st, b, beta, z);
  }
};
struct PolicyHS {  // half -> float
  using in_t = uint16_t;
  using out_t = float;
  using scal_t = float;
  static hipError_t run(Handle* h, const Problem<uint16_t, float>& p, float al, float be) {
    return run_gemm_hs(h, p, al, be);
  }
  static hipError_t scale(Handle* h, float* c, int64_t r, int64_t n, int64_t ld, int64_t st, int64_t b, float beta,
                          bool z) {
    return run_scale_f32(h, c, r, n, ld, st, b, beta, z);
  }
};
struct PolicyBS : PolicyHS {  // bfloat16 -> float
  static hipError_t run(Handle* h, const Problem<uint16_t, float>& p, float al, float be) {
    return run_gemm_bs(h, p, al, be);
  }
};

// Problem<> carries alpha/beta in the output type for the float/double back
// ends; the 16-bit back ends take them as separate floats.
template <typename TOut, typename TScal>
struct scalar_slot {
  static TOut put(TScal v) { return TOut(v); }
};
template <>
struct scalar_slot<uint16_t, float> {
  static uint16_t put(float) { return 0; }
};

// The reference's _gemm_backend (src/interface/gemm_interface.hpp:102-185),
// in the same order: alpha == 0 first, then the trans flags, then the strides.
template <typename P>
int gemm_backend(sbamd_handle_t hh, char transa, char transb, int64_t m, int64_t n, int64_t k,
                 typename P::scal_t alpha, const typename P::in_t* a, int64_t lda, int64_t stride_a,
                 const typename P::in_t* b, int64_t ldb, int64_t stride_b, typename P::scal_t beta,
                 typename P::out_t* c, int64_t ldc, int64_t stride_c, int64_t batch, int batch_type,
                 void* fixed_cube = nullptr, size_t fixed_cube_bytes = 0) {
  using TIn = typename P::in_t;
  using TOut = typename P::out_t;
  using TScal = typename P::scal_t;
  if (!hh) return SBAMD_INVALID_HANDLE;
  Handle* h = &hh->impl;
  if (m < 0 || n < 0 || k < 0 || batch < 0 || lda < 0 || ldb < 0 || ldc < 0) return SBAMD_INVALID_VALUE;
  if (batch > 0x7fffffffLL) return SBAMD_INVALID_VALUE;  // the kernels carry the batch count in 32 bits
  if (batch_type != SBAMD_BATCH_STRIDED && batch_type != SBAMD_BATCH_INTERLEAVED) return SBAMD_INVALID_VALUE;
  DeviceGuard guard(h->device);
  if (!guard.ok) return SBAMD_HIP_ERROR;
  h->last_kernel = -1;
  h->last_depth = 1;
  h->last_k_chunk = k;
  h->last_tile_m = h->last_tile_n = 0;
  h->last_fused = false;

  if (alpha == TScal(0)) {
    // gemm_interface.hpp:112-139.  size_c == stride_c picks a whole-buffer
    // scale; with ldc == m it is the vector _scal, whose zero factor stores
    // zeros (blas1_interface.hpp:444-448); every other route is _scal_matrix,
    // which multiplies (a factor of one launches nothing, :498-499).
    if (m == 0 || n == 0 || batch == 0) return SBAMD_OK;
    if (!c) return SBAMD_INVALID_VALUE;
    const int64_t size_c = ldc * n;
    hipError_t e = hipSuccess;
    if (batch_type == SBAMD_BATCH_INTERLEAVED && batch > 1) {
      // Interleaved items share every column: element (r, c) of item b sits
      // at c*ldc*batch + r*batch + b, so the m x n windows of all items form
      // one (m*batch) x n window with leading dimension ldc*batch.
      if (!(beta == TScal(1))) e = P::scale(h, c, m * batch, n, ldc * batch, 0, 1, beta, false);
    } else if (size_c == stride_c) {
      if (ldc == m) {
        e = P::scale(h, c, size_c * batch, 1, size_c * batch, 0, 1, beta, beta == TScal(0));
      } else if (!(beta == TScal(1))) {
        e = P::scale(h, c, m, n * batch, ldc, 0, 1, beta, false);
      }
    } else if (!(beta == TScal(1))) {
      e = P::scale(h, c, m, n, ldc, stride_c, batch, beta, false);
    }
    return e == hipSuccess ? SBAMD_OK : fail_hip(h, e, "scale launch");
  }

  const char ta_c = char(std::tolower(static_cast<unsigned char>(transa)));
  const char tb_c = char(std::tolower(static_cast<unsigned char>(transb)));
  if (ta_c != 'n' && ta_c != 't' && ta_c != 'c') return SBAMD_INVALID_TRANSA;  // :144-145
  if (tb_c != 'n' && tb_c != 't' && tb_c != 'c') return SBAMD_INVALID_TRANSB;  // :146-147
  const bool ta = ta_c != 'n';
  const bool tb = tb_c != 'n';

  if (batch > 1 && batch_type == SBAMD_BATCH_STRIDED) {  // :158-166
    const int64_t c_size = ldc * n;
    if (stride_c < c_size || stride_c < 0) return SBAMD_INVALID_STRIDEC;
    if (stride_a < 0) return SBAMD_INVALID_STRIDEA;
    if (stride_b < 0) return SBAMD_INVALID_STRIDEB;
  }

  // Nothing to write: the reference launches work-groups that exit at once
  // (src/operations/blas3/gemm_local.hpp:302, :596-598).
  if (m == 0 || n == 0 || batch == 0) return SBAMD_OK;
  if (!c || (k > 0 && (!a || !b))) return SBAMD_INVALID_VALUE;

  Problem<TIn, TOut> p;
  p.ta = ta;
  p.tb = tb;
  p.m = m;
  p.n = n;
  p.k = k;
  p.alpha = scalar_slot<TOut, TScal>::put(alpha);
  p.beta = scalar_slot<TOut, TScal>::put(beta);
  p.a = a;
  p.lda = lda;
  p.stride_a = stride_a;
  p.b = b;
  p.ldb = ldb;
  p.stride_b = stride_b;
  p.c = c;
  p.ldc = ldc;
  p.stride_c = stride_c;
  p.batch = batch;
  p.beta_zero = (beta == TScal(0));  // gemm_interface.hpp:92 (isZero)
  // One item has the same addresses in both layouts, so it takes the strided kernels.
  p.interleaved = batch_type == SBAMD_BATCH_INTERLEAVED && batch > 1;
  p.cube = fixed_cube;
  p.cube_bytes = fixed_cube_bytes;
  hipError_t e = P::run(h, p, alpha, beta);
  if (e == hipErrorNotSupported) return SBAMD_NOT_SUPPORTED;
  if (fixed_cube && e == hipErrorOutOfMemory) {
    h->last_error = "the pinned K cut needs more scratch than the fixed array of the no-allocation entry point holds";
    return SBAMD_SCRATCH_TOO_SMALL;
  }
  return e == hipSuccess ? SBAMD_OK : fail_hip(h, e, "gemm launch");
}

// _gemm_batched's default strides (gemm_interface.hpp:213-219).
template <typename P>
int gemm_batched(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n, int64_t k,
                 typename P::scal_t alpha, const typename P::in_t* a, int64_t lda, const typename P::in_t* b,
                 int64_t ldb, typename P::scal_t beta, typename P::out_t* c, int64_t ldc, int64_t batch,
                 int batch_type) {
  int64_t sa = 0, sb = 0, sc = 0;
  if (batch_type == SBAMD_BATCH_STRIDED) {
    sa = (std::tolower(static_cast<unsigned char>(transa)) != 'n') ? m * lda : k * lda;
    sb = (std::tolower(static_cast<unsigned char>(transb)) != 'n') ? ldb * k : n * ldb;
    sc = ldc * n;
  }
  return gemm_backend<P>(h, transa, transb, m, n, k, alpha, a, lda, sa, b, ldb, sb, beta, c, ldc, sc, batch,
                         batch_type);
}

}  // namespace
}  // namespace sbamd

using namespace sbamd;

extern "C" {

const char* sbamd_status_string(int status) {
  switch (status) {
    case SBAMD_OK: return "success";
    case SBAMD_INVALID_TRANSA: return "invalid _TransA";
    case SBAMD_INVALID_TRANSB: return "invalid _TransB";
    case SBAMD_INVALID_STRIDEC: return "invalid _stridec";
    case SBAMD_INVALID_STRIDEA: return "invalid _stridea";
    case SBAMD_INVALID_STRIDEB: return "invalid _strideb";
    case SBAMD_INVALID_HANDLE: return "invalid handle";
    case SBAMD_INVALID_VALUE: return "invalid value";
    case SBAMD_HIP_ERROR: return "HIP runtime error";
    case SBAMD_ALLOC_FAILED: return "device allocation failed";
    case SBAMD_NOT_SUPPORTED: return "not supported";
    case SBAMD_INVALID_UPLO: return "invalid _uplo";
    case SBAMD_INVALID_SIDE: return "invalid _side";
    case SBAMD_TRSM_INVALID_SIZE: return "invalid matrix size argument";
    case SBAMD_TRSM_INVALID_SIDE: return "invalid Side argument";
    case SBAMD_TRSM_INVALID_UPLO: return "invalid Triangle argument";
    case SBAMD_TRSM_INVALID_TRANS: return "invalid Transpose argument";
    case SBAMD_TRSM_INVALID_DIAG: return "invalid Diagonal argument";
    case SBAMD_SCRATCH_TOO_SMALL: return "scratch demand exceeds the fixed array of the no-allocation entry point";
    // the multi-GPU layer's codes (include/sbamd_dist.h)
    case 19: return "RCCL call failed or librccl could not be loaded";
    case 20: return "a peer did not arrive in time";
    case 21: return "the caller's all-gather callback failed";
    case 22: return "the communicator lacks the transport or window the operation needs";
    default: return "unknown status";
  }
}

int sbamd_create(sbamd_handle_t* out, int device, void* stream) {
  if (!out) return SBAMD_INVALID_VALUE;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return SBAMD_HIP_ERROR;
  sbamd_handle_s* h = new (std::nothrow) sbamd_handle_s;
  if (!h) return SBAMD_ALLOC_FAILED;
  h->impl.device = device;
  DeviceGuard guard(device);
  hipDeviceProp_t prop;
  if (!guard.ok || hipGetDeviceProperties(&prop, device) != hipSuccess) {
    delete h;
    return SBAMD_HIP_ERROR;
  }
  h->impl.compute_units = prop.multiProcessorCount;
  // this device's copy of the fixed scratch array (device `device` is current here)
  if (fixed_scratch_of(device, &h->impl.fixed_scratch) != hipSuccess) {
    delete h;
    return SBAMD_HIP_ERROR;
  }
  if (stream) {
    h->impl.stream = static_cast<hipStream_t>(stream);
  } else {
    if (hipStreamCreateWithFlags(&h->impl.stream, hipStreamNonBlocking) != hipSuccess) {
      delete h;
      return SBAMD_HIP_ERROR;
    }
    h->impl.owns_stream = true;
  }
  *out = h;
  return SBAMD_OK;
}

int sbamd_destroy(sbamd_handle_t h) {
  if (!h) return SBAMD_INVALID_HANDLE;
  DeviceGuard guard(h->impl.device);
  (void)hipStreamSynchronize(h->impl.stream);
  {
    std::lock_guard<std::mutex> lock(h->impl.mu);
    release_retired(&h->impl, true);
  }
  if (h->impl.scratch) (void)hipFree(h->impl.scratch);
  if (h->impl.aux) (void)hipFree(h->impl.aux);
  if (h->impl.owns_stream) (void)hipStreamDestroy(h->impl.stream);
  delete h;
  return SBAMD_OK;
}

int sbamd_wait(sbamd_handle_t h) {
  if (!h) return SBAMD_INVALID_HANDLE;
  hipError_t e = hipStreamSynchronize(h->impl.stream);
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lock(h->impl.mu);
    if (!h->impl.retired.empty()) {
      DeviceGuard guard(h->impl.device);
      release_retired(&h->impl, true);
    }
  }
  return e == hipSuccess ? SBAMD_OK : fail_hip(&h->impl, e, "stream synchronize");
}

int sbamd_get_num_compute_units(sbamd_handle_t h, int* out) {
  if (!h) return SBAMD_INVALID_HANDLE;
  if (!out) return SBAMD_INVALID_VALUE;
  *out = h->impl.compute_units;
  return SBAMD_OK;
}

int sbamd_get_stream(sbamd_handle_t h, void** out_stream) {
  if (!h) return SBAMD_INVALID_HANDLE;
  if (!out_stream) return SBAMD_INVALID_VALUE;
  *out_stream = h->impl.stream;
  return SBAMD_OK;
}

const char* sbamd_last_error(sbamd_handle_t h) { return h ? h->impl.last_error.c_str() : "invalid handle"; }

int sbamd_set_kernel(sbamd_handle_t h, int kernel, int param) {
  if (!h) return SBAMD_INVALID_HANDLE;
  if (kernel < SBAMD_KERNEL_AUTO || (kernel > SBAMD_KERNEL_TINY && kernel != SBAMD_KERNEL_MEDIUM) || kernel == SBAMD_KERNEL_INTERLEAVED ||
      param < 0)
    return SBAMD_INVALID_VALUE;
  h->impl.kernel_override = kernel;
  h->impl.kernel_param = param;
  return SBAMD_OK;
}

int sbamd_last_kernel(sbamd_handle_t h) { return h ? h->impl.last_kernel : -1; }

int sbamd_last_splitk(sbamd_handle_t h, int64_t* depth, int64_t* k_chunk) {
  if (!h) return SBAMD_INVALID_HANDLE;
  if (!depth || !k_chunk) return SBAMD_INVALID_VALUE;
  *depth = h->impl.last_depth;
  *k_chunk = h->impl.last_k_chunk;
  return SBAMD_OK;
}

int sbamd_last_tile(sbamd_handle_t h, int* tile_m, int* tile_n, int* fused) {
  if (!h) return SBAMD_INVALID_HANDLE;
  if (!tile_m || !tile_n || !fused) return SBAMD_INVALID_VALUE;
  *tile_m = h->impl.last_tile_m;
  *tile_n = h->impl.last_tile_n;
  *fused = h->impl.last_fused ? 1 : 0;
  return SBAMD_OK;
}

int sbamd_last_trsm_launches(sbamd_handle_t h, int64_t* diag, int64_t* panel) {
  if (!h) return SBAMD_INVALID_HANDLE;
  if (!diag || !panel) return SBAMD_INVALID_VALUE;
  *diag = h->impl.last_trsm_diag;
  *panel = h->impl.last_trsm_panel;
  return SBAMD_OK;
}

int sbamd_event_record(sbamd_handle_t h, sbamd_event_t* out) {
  if (!h) return SBAMD_INVALID_HANDLE;
  if (!out) return SBAMD_INVALID_VALUE;
  DeviceGuard guard(h->impl.device);
  sbamd_event_s* ev = new (std::nothrow) sbamd_event_s;
  if (!ev) return SBAMD_ALLOC_FAILED;
  hipError_t e = hipEventCreate(&ev->ev);
  if (e == hipSuccess) e = hipEventRecord(ev->ev, h->impl.stream);
  if (e != hipSuccess) {
    delete ev;
    return fail_hip(&h->impl, e, "event record");
  }
  *out = ev;
  return SBAMD_OK;
}

int sbamd_stream_wait_event(sbamd_handle_t h, sbamd_event_t ev) {
  if (!h) return SBAMD_INVALID_HANDLE;
  if (!ev) return SBAMD_INVALID_VALUE;
  DeviceGuard guard(h->impl.device);
  hipError_t e = hipStreamWaitEvent(h->impl.stream, ev->ev, 0);
  return e == hipSuccess ? SBAMD_OK : fail_hip(&h->impl, e, "stream wait event");
}

int sbamd_event_synchronize(sbamd_event_t ev) {
  if (!ev) return SBAMD_INVALID_VALUE;
  return hipEventSynchronize(ev->ev) == hipSuccess ? SBAMD_OK : SBAMD_HIP_ERROR;
}

int sbamd_event_elapsed_ms(sbamd_event_t start, sbamd_event_t stop, float* ms) {
  if (!start || !stop || !ms) return SBAMD_INVALID_VALUE;
  return hipEventElapsedTime(ms, start->ev, stop->ev) == hipSuccess ? SBAMD_OK : SBAMD_HIP_ERROR;
}

int sbamd_event_destroy(sbamd_event_t ev) {
  if (!ev) return SBAMD_INVALID_VALUE;
  (void)hipEventDestroy(ev->ev);
  delete ev;
  return SBAMD_OK;
}

int sbamd_malloc(sbamd_handle_t h, void** out, size_t bytes) {
  if (!h) return SBAMD_INVALID_HANDLE;
  if (!out) return SBAMD_INVALID_VALUE;
  DeviceGuard guard(h->impl.device);
  *out = nullptr;
  hipError_t e = hipMalloc(out, bytes ? bytes : 1);
  return e == hipSuccess ? SBAMD_OK : fail_hip(&h->impl, e, "hipMalloc");
}

int sbamd_free(sbamd_handle_t h, void* ptr) {
  if (!h) return SBAMD_INVALID_HANDLE;
  if (!ptr) return SBAMD_OK;
  DeviceGuard guard(h->impl.device);
  hipError_t e = hipFree(ptr);  // waits for outstanding work on the device by itself
  return e == hipSuccess ? SBAMD_OK : fail_hip(&h->impl, e, "hipFree");
}

int sbamd_device_free(int device, void* ptr) {
  if (!ptr) return SBAMD_OK;
  DeviceGuard guard(device);
  if (!guard.ok) return SBAMD_HIP_ERROR;
  return hipFree(ptr) == hipSuccess ? SBAMD_OK : SBAMD_HIP_ERROR;
}

int sbamd_copy_to_device(sbamd_handle_t h, void* dst_dev, const void* src_host, size_t bytes) {
  if (!h) return SBAMD_INVALID_HANDLE;
  if (bytes == 0) return SBAMD_OK;
  if (!dst_dev || !src_host) return SBAMD_INVALID_VALUE;
  DeviceGuard guard(h->impl.device);
  hipError_t e = hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, h->impl.stream);
  return e == hipSuccess ? SBAMD_OK : fail_hip(&h->impl, e, "copy to device");
}

int sbamd_copy_to_host(sbamd_handle_t h, void* dst_host, const void* src_dev, size_t bytes) {
  if (!h) return SBAMD_INVALID_HANDLE;
  if (bytes == 0) return SBAMD_OK;
  if (!dst_host || !src_dev) return SBAMD_INVALID_VALUE;
  DeviceGuard guard(h->impl.device);
  hipError_t e = hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, h->impl.stream);
  return e == hipSuccess ? SBAMD_OK : fail_hip(&h->impl, e, "copy to host");
}

int sbamd_memset(sbamd_handle_t h, void* dst_dev, int byte_value, size_t bytes) {
  if (!h) return SBAMD_INVALID_HANDLE;
  if (bytes == 0) return SBAMD_OK;
  if (!dst_dev) return SBAMD_INVALID_VALUE;
  DeviceGuard guard(h->impl.device);
  hipError_t e = hipMemsetAsync(dst_dev, byte_value, bytes, h->impl.stream);
  return e == hipSuccess ? SBAMD_OK : fail_hip(&h->impl, e, "memset");
}

int sbamd_fill_uniform(sbamd_handle_t h, void* dst_dev, int64_t count, int dtype, uint64_t seed,
                       double lo, double hi) {
  if (!h) return SBAMD_INVALID_HANDLE;
  if (count < 0 || dtype < 0 || dtype > 3 || (count > 0 && !dst_dev)) return SBAMD_INVALID_VALUE;
  DeviceGuard guard(h->impl.device);
  hipError_t e = run_fill_uniform(&h->impl, dst_dev, count, dtype, seed, lo, hi);
  return e == hipSuccess ? SBAMD_OK : fail_hip(&h->impl, e, "fill launch");
}

/* ------------------------------------------------------------ fp32 / fp64 */

int sbamd_sgemm(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n, int64_t k,
                float alpha, const float* a, int64_t lda, const float* b, int64_t ldb, float beta,
                float* c, int64_t ldc) {
  // blas::_gemm: strides 0, batch 1, strided (gemm_interface.hpp:194-197)
  return gemm_backend<PolicyS>(h, transa, transb, m, n, k, alpha, a, lda, 0, b, ldb, 0, beta, c,
                                    ldc, 0, 1, SBAMD_BATCH_STRIDED);
}

int sbamd_sgemm_noalloc(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n, int64_t k,
                        float alpha, const float* a, int64_t lda, const float* b, int64_t ldb, float beta,
                        float* c, int64_t ldc) {
  if (!h) return SBAMD_INVALID_HANDLE;
  Handle* hp = &h->impl;
  if (!hp->fixed_scratch) return SBAMD_HIP_ERROR;
  // The cube of this call is the device's fixed array; it travels with the call (the handle's own
  // buffers are neither used nor touched, so other calls on the handle are not affected).
  return gemm_backend<PolicyS>(h, transa, transb, m, n, k, alpha, a, lda, 0, b, ldb, 0, beta, c, ldc, 0, 1,
                               SBAMD_BATCH_STRIDED, hp->fixed_scratch, SBAMD_NOALLOC_SCRATCH_BYTES);
}

int sbamd_dgemm(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n, int64_t k,
                double alpha, const double* a, int64_t lda, const double* b, int64_t ldb,
                double beta, double* c, int64_t ldc) {
  return gemm_backend<PolicyD>(h, transa, transb, m, n, k, alpha, a, lda, 0, b, ldb, 0, beta,
                                      c, ldc, 0, 1, SBAMD_BATCH_STRIDED);
}

int sbamd_sgemm_batched(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n, int64_t k,
                        float alpha, const float* a, int64_t lda, const float* b, int64_t ldb,
                        float beta, float* c, int64_t ldc, int64_t batch, int batch_type) {
  return gemm_batched<PolicyS>(h, transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc,
                                    batch, batch_type);
}

int sbamd_dgemm_batched(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n, int64_t k,
                        double alpha, const double* a, int64_t lda, const double* b, int64_t ldb,
                        double beta, double* c, int64_t ldc, int64_t batch, int batch_type) {
  return gemm_batched<PolicyD>(h, transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c,
                                      ldc, batch, batch_type);
}

int sbamd_sgemm_strided_batched(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n,
                                int64_t k, float alpha, const float* a, int64_t lda,
                                int64_t stride_a, const float* b, int64_t ldb, int64_t stride_b,
                                float beta, float* c, int64_t ldc, int64_t stride_c,
                                int64_t batch) {
  return gemm_backend<PolicyS>(h, transa, transb, m, n, k, alpha, a, lda, stride_a, b, ldb,
                                    stride_b, beta, c, ldc, stride_c, batch, SBAMD_BATCH_STRIDED);
}

int sbamd_dgemm_strided_batched(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n,
                                int64_t k, double alpha, const double* a, int64_t lda,
                                int64_t stride_a, const double* b, int64_t ldb, int64_t stride_b,
                                double beta, double* c, int64_t ldc, int64_t stride_c,
                                int64_t batch) {
  return gemm_backend<PolicyD>(h, transa, transb, m, n, k, alpha, a, lda, stride_a, b, ldb,
                                      stride_b, beta, c, ldc, stride_c, batch, SBAMD_BATCH_STRIDED);
}

/* ------------------------------------------------- half / bfloat16 storage */

// alpha / beta of the half -> half pair have the output type in the reference
// (gemm.cpp.in:33-39): round them to binary16 before use.
static inline float to_half_precision(float v) { return float(_Float16(v)); }

int sbamd_hgemm(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n, int64_t k, float alpha,
                const uint16_t* a, int64_t lda, const uint16_t* b, int64_t ldb, float beta, uint16_t* c,
                int64_t ldc) {
  return gemm_backend<PolicyH>(h, transa, transb, m, n, k, to_half_precision(alpha), a, lda, 0, b, ldb, 0,
                               to_half_precision(beta), c, ldc, 0, 1, SBAMD_BATCH_STRIDED);
}
int sbamd_hsgemm(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n, int64_t k, float alpha,
                 const uint16_t* a, int64_t lda, const uint16_t* b, int64_t ldb, float beta, float* c,
                 int64_t ldc) {
  return gemm_backend<PolicyHS>(h, transa, transb, m, n, k, alpha, a, lda, 0, b, ldb, 0, beta, c, ldc, 0, 1,
                                SBAMD_BATCH_STRIDED);
}
int sbamd_bsgemm(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n, int64_t k, float alpha,
                 const uint16_t* a, int64_t lda, const uint16_t* b, int64_t ldb, float beta, float* c,
                 int64_t ldc) {
  return gemm_backend<PolicyBS>(h, transa, transb, m, n, k, alpha, a, lda, 0, b, ldb, 0, beta, c, ldc, 0, 1,
                                SBAMD_BATCH_STRIDED);
}
int sbamd_hgemm_batched(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n, int64_t k, float alpha,
                        const uint16_t* a, int64_t lda, const uint16_t* b, int64_t ldb, float beta, uint16_t* c,
                        int64_t ldc, int64_t batch, int batch_type) {
  return gemm_batched<PolicyH>(h, transa, transb, m, n, k, to_half_precision(alpha), a, lda, b, ldb,
                               to_half_precision(beta), c, ldc, batch, batch_type);
}
int sbamd_hsgemm_batched(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n, int64_t k, float alpha,
                         const uint16_t* a, int64_t lda, const uint16_t* b, int64_t ldb, float beta, float* c,
                         int64_t ldc, int64_t batch, int batch_type) {
  return gemm_batched<PolicyHS>(h, transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, batch, batch_type);
}
int sbamd_bsgemm_batched(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n, int64_t k, float alpha,
                         const uint16_t* a, int64_t lda, const uint16_t* b, int64_t ldb, float beta, float* c,
                         int64_t ldc, int64_t batch, int batch_type) {
  return gemm_batched<PolicyBS>(h, transa, transb, m, n, k, alpha, a, lda, b, ldb, beta, c, ldc, batch, batch_type);
}
int sbamd_hgemm_strided_batched(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n, int64_t k,
                                float alpha, const uint16_t* a, int64_t lda, int64_t stride_a, const uint16_t* b,
                                int64_t ldb, int64_t stride_b, float beta, uint16_t* c, int64_t ldc,
                                int64_t stride_c, int64_t batch) {
  return gemm_backend<PolicyH>(h, transa, transb, m, n, k, to_half_precision(alpha), a, lda, stride_a, b, ldb,
                               stride_b, to_half_precision(beta), c, ldc, stride_c, batch, SBAMD_BATCH_STRIDED);
}
int sbamd_hsgemm_strided_batched(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n, int64_t k,
                                 float alpha, const uint16_t* a, int64_t lda, int64_t stride_a, const uint16_t* b,
                                 int64_t ldb, int64_t stride_b, float beta, float* c, int64_t ldc,
                                 int64_t stride_c, int64_t batch) {
  return gemm_backend<PolicyHS>(h, transa, transb, m, n, k, alpha, a, lda, stride_a, b, ldb, stride_b, beta, c, ldc,
                                stride_c, batch, SBAMD_BATCH_STRIDED);
}
int sbamd_bsgemm_strided_batched(sbamd_handle_t h, char transa, char transb, int64_t m, int64_t n, int64_t k,
                                 float alpha, const uint16_t* a, int64_t lda, int64_t stride_a, const uint16_t* b,
                                 int64_t ldb, int64_t stride_b, float beta, float* c, int64_t ldc,
                                 int64_t stride_c, int64_t batch) {
  return gemm_backend<PolicyBS>(h, transa, transb, m, n, k, alpha, a, lda, stride_a, b, ldb, stride_b, beta, c, ldc,
                                stride_c, batch, SBAMD_BATCH_STRIDED);
}

}  // extern "C"
