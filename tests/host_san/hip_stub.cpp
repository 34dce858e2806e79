// Stand-in for the HIP runtime, for running the C ABI's HOST logic (argument rules, planner, buffer bookkeeping,
// copies, the no-allocation path, handle lifetime) on a machine without a GPU under AddressSanitizer and
// UndefinedBehaviorSanitizer.  "Device" memory is host memory, so every copy the library issues is bounds-checked
// by the sanitizer and every buffer it forgets to free is reported as a leak; kernel launches are checked for a
// legal grid / block / LDS size and otherwise do nothing (no result is computed: this is not a CPU fallback and the
// product never links it).  Test infrastructure only; built by tests/host_san/Makefile.
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

namespace {
std::mutex g_mu;
std::map<uintptr_t, size_t> g_allocs;          // base -> bytes
std::map<const void*, size_t> g_symbols;       // host stand-in of a device variable -> bytes
std::map<const void*, void*> g_symbol_memory;  // ... -> its "device" memory
long g_launches = 0;
int g_device = 0;
struct Config {
  dim3 grid, block;
  size_t lds;
  hipStream_t stream;
};
thread_local Config g_config;

struct StubStream {
  int alive = 1;
};
struct StubEvent {
  std::chrono::steady_clock::time_point t;
  bool recorded = false;
};

hipError_t alloc(void** p, size_t bytes) {
  if (!p) return hipErrorInvalidValue;
  if (bytes > (size_t(8) << 30)) return hipErrorOutOfMemory;  // the machine this runs on is small
  *p = std::malloc(bytes ? bytes : 1);
  if (!*p) return hipErrorOutOfMemory;
  std::lock_guard<std::mutex> lock(g_mu);
  g_allocs[uintptr_t(*p)] = bytes;
  return hipSuccess;
}
}  // namespace

extern "C" long hip_stub_launches() { return g_launches; }
extern "C" long hip_stub_live_allocations() {
  std::lock_guard<std::mutex> lock(g_mu);
  return long(g_allocs.size());
}

extern "C" {
void** __hipRegisterFatBinary(const void*) {
  static void* module = nullptr;
  return &module;
}
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void* var, char*, char*, int, size_t size, int, int) {
  std::lock_guard<std::mutex> lock(g_mu);
  g_symbols[var] = size;
}
void __hipRegisterManagedVar(void*, void**, void*, const char*, size_t, unsigned) {}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t lds, hipStream_t stream) {
  g_config = {grid, block, lds, stream};
  return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* lds, hipStream_t* stream) {
  *grid = g_config.grid, *block = g_config.block, *lds = g_config.lds, *stream = g_config.stream;
  return hipSuccess;
}
}

hipError_t hipLaunchKernel(const void* f, dim3 grid, dim3 block, void** args, size_t lds, hipStream_t) {
  const uint64_t threads = uint64_t(block.x) * block.y * block.z;
  if (!f || !args || grid.x == 0 || grid.y == 0 || grid.z == 0 || threads == 0 || threads > 1024 || lds > 160 * 1024 ||
      grid.y > 65535 || grid.z > 65535) {
    std::fprintf(stderr, "hip_stub: illegal launch grid (%u, %u, %u) block (%u, %u, %u) lds %zu\n", grid.x, grid.y, grid.z,
                 block.x, block.y, block.z, lds);
    std::abort();
  }
  ++g_launches;
  return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "error (stub runtime)"; }
hipError_t hipGetDeviceCount(int* n) { return *n = 2, hipSuccess; }
hipError_t hipGetDevice(int* d) { return *d = g_device, hipSuccess; }
hipError_t hipSetDevice(int d) {
  if (d < 0 || d >= 2) return hipErrorInvalidDevice;
  return g_device = d, hipSuccess;
}
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int d) {
  if (d < 0 || d >= 2) return hipErrorInvalidDevice;
  std::memset(p, 0, sizeof(*p));
  std::snprintf(p->name, sizeof(p->name), "stub gfx950");
  std::snprintf(p->gcnArchName, sizeof(p->gcnArchName), "gfx950:sramecc+:xnack-");
  p->multiProcessorCount = 256;
  p->totalGlobalMem = size_t(288) << 30;
  p->warpSize = 64;
  p->maxThreadsPerBlock = 1024;
  p->sharedMemPerBlock = 160 * 1024;
  return hipSuccess;
}
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipFuncSetAttribute(const void* f, hipFuncAttribute, int value) { return f && value >= 0 ? hipSuccess : hipErrorInvalidValue; }

hipError_t hipMalloc(void** p, size_t bytes) { return alloc(p, bytes); }
hipError_t hipExtMallocWithFlags(void** p, size_t bytes, unsigned) { return alloc(p, bytes); }
hipError_t hipFree(void* p) {
  if (!p) return hipSuccess;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_allocs.erase(uintptr_t(p))) {
      std::fprintf(stderr, "hip_stub: hipFree of a pointer hipMalloc did not return\n");
      std::abort();
    }
  }
  std::free(p);
  return hipSuccess;
}
hipError_t hipMemGetAddressRange(hipDeviceptr_t* base, size_t* size, hipDeviceptr_t p) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_allocs.upper_bound(uintptr_t(p));
  if (it == g_allocs.begin()) return hipErrorInvalidValue;
  --it;
  if (uintptr_t(p) >= it->first + (it->second ? it->second : 1)) return hipErrorInvalidValue;
  if (base) *base = hipDeviceptr_t(it->first);
  if (size) *size = it->second;
  return hipSuccess;
}
hipError_t hipGetSymbolAddress(void** p, const void* symbol) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_symbols.find(symbol);
  if (it == g_symbols.end()) return hipErrorInvalidSymbol;
  void*& mem = g_symbol_memory[symbol];
  if (!mem) mem = std::calloc(1, it->second);  // lives as long as the process, like the device variable
  *p = mem;
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t) {
  if (bytes) std::memmove(dst, src, bytes);
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind,
                            hipStream_t) {
  if (width > dpitch || width > spitch) return hipErrorInvalidPitchValue;
  for (size_t r = 0; r < height; ++r) std::memmove((char*)dst + r * dpitch, (const char*)src + r * spitch, width);
  return hipSuccess;
}
hipError_t hipMemset(void* dst, int v, size_t bytes) { return std::memset(dst, v, bytes), hipSuccess; }
hipError_t hipMemsetAsync(void* dst, int v, size_t bytes, hipStream_t) { return std::memset(dst, v, bytes), hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return *s = reinterpret_cast<hipStream_t>(new StubStream), hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) {
  delete reinterpret_cast<StubStream*>(s);
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t e, unsigned) { return e ? hipSuccess : hipErrorInvalidHandle; }
hipError_t hipEventCreate(hipEvent_t* e) { return *e = reinterpret_cast<hipEvent_t>(new StubEvent), hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) {
  delete reinterpret_cast<StubEvent*>(e);
  return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
  auto* ev = reinterpret_cast<StubEvent*>(e);
  ev->t = std::chrono::steady_clock::now();
  ev->recorded = true;
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) { return reinterpret_cast<StubEvent*>(e)->recorded ? hipSuccess : hipErrorInvalidHandle; }
hipError_t hipEventQuery(hipEvent_t e) { return reinterpret_cast<StubEvent*>(e)->recorded ? hipSuccess : hipErrorInvalidHandle; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  auto *ea = reinterpret_cast<StubEvent*>(a), *eb = reinterpret_cast<StubEvent*>(b);
  if (!ea->recorded || !eb->recorded) return hipErrorInvalidHandle;
  *ms = std::chrono::duration<float, std::milli>(eb->t - ea->t).count();
  return hipSuccess;
}
// one process, one "device": a handle is the pointer itself
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* h, void* p) {
  std::memset(h, 0, sizeof(*h));
  std::memcpy(h, &p, sizeof(p));
  return hipSuccess;
}
hipError_t hipIpcOpenMemHandle(void** p, hipIpcMemHandle_t h, unsigned) {
  std::memcpy(p, &h, sizeof(*p));
  return hipSuccess;
}
hipError_t hipIpcCloseMemHandle(void*) { return hipSuccess; }
