// Probe: does v_mfma_f32_16x16x4_f32 add its four products in k order with one rounding each (a sequential fused
// multiply-add chain, as v_mfma_f32_32x32x2_f32 and v_mfma_f64_16x16x4_f64 do)?  Compares a K = 64 product of
// random operands bit for bit with fmaf chains on the host.
//   hipcc -O2 --offload-arch=gfx950 -ffp-contract=off tools/mfma16_order.hip -o tools/bin/mfma16_order
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// lane l: op0[i = l & 15][k = l >> 4], op1[k = l >> 4][j = l & 15]; D[i = 4 * (l >> 4) + r][j = l & 15], r = 0..3
__global__ void probe(const float* a, const float* b, float* d, int k) {
  const int lane = threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  for (int k0 = 0; k0 < k; k0 += 4) {
    const float av = a[(lane & 15) * k + k0 + (lane >> 4)];   // A row-major 16 x k
    const float bv = b[(k0 + (lane >> 4)) * 16 + (lane & 15)];  // B row-major k x 16
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) d[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
}

int main() {
  const int k = 64;
  std::vector<float> a(16 * k), b(k * 16), d(256);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return float(int(s >> 8) % 20001 - 10000) / 3000.0f; };
  for (auto& x : a) x = rnd();
  for (auto& x : b) x = rnd();
  float *da, *db, *dd;
  hipMalloc(&da, a.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dd, 256 * 4);
  hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd, k);
  hipMemcpy(d.data(), dd, 256 * 4, hipMemcpyDeviceToHost);
  int seq = 0, pair = 0, unfused = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      float c1 = 0, c3 = 0;
      for (int kk = 0; kk < k; ++kk) c1 = fmaf(a[i * k + kk], b[kk * 16 + j], c1);
      for (int kk = 0; kk < k; ++kk) { volatile float p = a[i * k + kk] * b[kk * 16 + j]; c3 = c3 + p; }
      float c2 = 0;  // groups of four: ((p0 + p1) + (p2 + p3)) in higher precision, then added
      for (int k0 = 0; k0 < k; k0 += 4) {
        double t = 0;
        for (int q = 0; q < 4; ++q) t += double(a[i * k + k0 + q]) * double(b[(k0 + q) * 16 + j]);
        c2 = float(double(c2) + t);
      }
      const float g = d[i * 16 + j];
      seq += std::memcmp(&g, &c1, 4) == 0;
      pair += std::memcmp(&g, &c2, 4) == 0;
      unfused += std::memcmp(&g, &c3, 4) == 0;
    }
  std::printf("v_mfma_f32_16x16x4_f32, K = 64: %d / 256 elements equal the sequential fmaf chain, %d the four-at-a-time exact sum, %d the unfused chain\n",
              seq, pair, unfused);
  return 0;
}
