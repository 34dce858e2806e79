// Lab: what the chip sustains when 1024 resident workgroups write a column-major n x n fp32 matrix tile by tile,
// for several shapes of the per-wavefront store (no arithmetic, no reads): which epilogue layout a write-bound
// GEMM (K <= 64) should use.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/store_lab.hip -o tools/bin/store_lab
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                         \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: the MFMA accumulator layout of a 64 x 64 tile on 2 x 2 wavefronts -- a lane holds one row, a store
//         instruction writes 2 columns x 128 bytes
// MODE 1: 64 x 64 tile, a lane holds 4 consecutive rows of one column: an instruction writes 4 columns x 256 bytes
// MODE 2: 256 x 16 tile, a lane holds 4 consecutive rows: an instruction writes 1 column x 1 KiB
// MODE 3: 64 x 64 tile as MODE 1 with non-temporal stores
// MODE 4: linear fill in 16-byte pieces (the yardstick)
template <int MODE>
__global__ void __launch_bounds__(256) store_kernel(float* c, int n, int tiles_m, int tiles_n) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long ntiles = long(tiles_m) * tiles_n;
  for (long z = blockIdx.x; z < ntiles; z += gridDim.x) {
    const long tm = z % tiles_m, tn = z / tiles_m;
    const float v = float(z);
    if (MODE == 0) {
      const int wm = wave & 1, wn = wave >> 1;
      float* p = c + (tm * 64 + wm * 32 + (lane & 31)) + (tn * 64 + wn * 32 + 4 * (lane >> 5)) * long(n);
#pragma unroll
      for (int r = 0; r < 16; ++r) p[((r & 3) + 8 * (r >> 2)) * long(n)] = v;
    } else if (MODE == 1 || MODE == 3) {
      float* p = c + (tm * 64 + 4 * (lane & 15)) + (tn * 64 + wave * 16 + (lane >> 4)) * long(n);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        f32x4 vv = {v, v, v, v};
        if (MODE == 3) __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(p + 4 * r * long(n)));
        else *reinterpret_cast<f32x4*>(p + 4 * r * long(n)) = vv;
      }
    } else if (MODE == 2) {
      float* p = c + (tm * 256 + 4 * lane) + (tn * 16 + wave) * long(n);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        f32x4 vv = {v, v, v, v};
        *reinterpret_cast<f32x4*>(p + 4 * r * long(n)) = vv;
      }
    } else {
      f32x4 vv = {v, v, v, v};
#pragma unroll
      for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(c + z * 4096 + r * 1024 + tid * 4) = vv;
    }
  }
}

template <int MODE>
int run(const char* name, float* c, int n, int grid) {
  const int tm = MODE == 2 ? n / 256 : n / 64, tn = MODE == 2 ? n / 16 : n / 64;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(grid), dim3(256), 0, 0, c, n, tm, tn);
  CHECK(hipDeviceSynchronize());
  const int reps = 20;
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(grid), dim3(256), 0, 0, c, n, tm, tn);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms / reps * 1e3;
  std::printf("%-70s n=%d grid=%d: %7.1f us  %.2f TB/s\n", name, n, grid, us, double(n) * n * 4 / us / 1e6);
  return 0;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 8192;
  float* c;
  CHECK(hipMalloc(&c, size_t(n) * n * 4));
  for (int grid : {1024, 2048, 16384}) {
    if (run<0>("64x64 tile, accumulator layout (2 columns x 128 B per instruction)", c, n, grid)) return 1;
    if (run<1>("64x64 tile, 4 rows per lane (4 columns x 256 B per instruction)", c, n, grid)) return 1;
    if (run<3>("64x64 tile, 4 rows per lane, non-temporal", c, n, grid)) return 1;
    if (run<2>("256x16 tile, 4 rows per lane (1 column x 1 KiB per instruction)", c, n, grid)) return 1;
    if (run<4>("linear fill", c, n, grid)) return 1;
  }
  return 0;
}
