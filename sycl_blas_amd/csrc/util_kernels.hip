// Device-side helpers that are not GEMM kernels: the counter-based uniform
// fill used to create multi-GiB benchmark operands in place.
#include <hip/hip_fp16.h>

#include "sbamd_internal.hpp"

namespace sbamd {

// splitmix64 finaliser: element i of a buffer depends only on (seed, i), so
// the same values can be regenerated on the host for spot checks.
__host__ __device__ inline uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__device__ inline double uniform01(uint64_t seed, uint64_t i) {
  return double(mix64(seed * 0xD1342543DE82EF95ull + i) >> 11) * (1.0 / 9007199254740992.0);
}

__device__ inline uint16_t f32_to_bf16_rne(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);  // finite inputs only (the fill never makes NaN)
  return uint16_t(u >> 16);
}

template <int DTYPE>
__global__ void __launch_bounds__(256)
    fill_uniform_kernel(void* dst, int64_t count, uint64_t seed, double lo, double hi) {
  const int64_t stride = int64_t(gridDim.x) * 256;
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < count; i += stride) {
    const double v = lo + (hi - lo) * uniform01(seed, uint64_t(i));
    if (DTYPE == 0) static_cast<float*>(dst)[i] = float(v);
    if (DTYPE == 1) static_cast<double*>(dst)[i] = v;
    if (DTYPE == 2) static_cast<__half*>(dst)[i] = __float2half_rn(float(v));
    if (DTYPE == 3) static_cast<uint16_t*>(dst)[i] = f32_to_bf16_rne(float(v));
  }
}

hipError_t run_fill_uniform(Handle* h, void* dst, int64_t count, int dtype, uint64_t seed,
                            double lo, double hi) {
  if (count <= 0) return hipSuccess;
  const int64_t want = (count + 255) / 256;
  const unsigned grid = unsigned(want < 8192 ? want : 8192);
  switch (dtype) {
    case 0: hipLaunchKernelGGL(fill_uniform_kernel<0>, dim3(grid), dim3(256), 0, h->stream, dst, count, seed, lo, hi); break;
    case 1: hipLaunchKernelGGL(fill_uniform_kernel<1>, dim3(grid), dim3(256), 0, h->stream, dst, count, seed, lo, hi); break;
    case 2: hipLaunchKernelGGL(fill_uniform_kernel<2>, dim3(grid), dim3(256), 0, h->stream, dst, count, seed, lo, hi); break;
    case 3: hipLaunchKernelGGL(fill_uniform_kernel<3>, dim3(grid), dim3(256), 0, h->stream, dst, count, seed, lo, hi); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace sbamd
