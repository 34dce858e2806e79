// Drop-in demo for the two callers of the GEMM path, through portblas_compat.hpp:
// blas::_symm with the arguments of the reference's sample (samples/symm.cpp:18-26,
// :67-68 in the reference tree: m = 5, n = 7, leading dimensions 32, side 'l',
// uplo 'l', alpha 1.5, beta 0.5) and blas::_trsm on a system built from the same
// matrix.  Both results are checked on the host: _symm against the mirrored
// triangle multiplied out with a sequential-k fma loop (the same chain the library
// builds, so the match is exact), _trsm by the residual of op(A) * X - alpha * B.
// With an argument N both run N x N instead of the sample's sizes.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "portblas_compat.hpp"

int main(int argc, char** argv) {
  blas::SB_Handle sb_handle(0);

  const bool big = argc > 1;
  const size_t m = big ? size_t(std::atoi(argv[1])) : 5;
  const size_t n = big ? m : 7;
  const size_t lda = big ? m : 32, ldb = big ? m : 32, ldc = big ? m : 32;
  const float alpha = 1.5f, beta = 0.5f;
  const char side = 'l', uplo = 'l';
  const size_t k = side == 'l' ? m : n;

  std::vector<float> A(lda * k), B(ldb * n), C(ldc * n);
  std::mt19937 gen(2024);
  std::uniform_real_distribution<float> dist(-1.0f, 1.0f);
  for (auto& v : A) v = dist(gen) / float(k);
  for (size_t i = 0; i < k; ++i) A[i + i * lda] = 2.0f + 0.01f * float(i % 7);  // a well-conditioned triangle
  for (auto& v : B) v = dist(gen);
  for (auto& v : C) v = dist(gen);
  const std::vector<float> C0 = C;

  auto a_gpu = blas::make_sycl_iterator_buffer<float>(sb_handle, lda * k);
  auto b_gpu = blas::make_sycl_iterator_buffer<float>(sb_handle, ldb * n);
  auto c_gpu = blas::make_sycl_iterator_buffer<float>(sb_handle, ldc * n);
  blas::helper::copy_to_device(sb_handle, A.data(), a_gpu, lda * k);
  blas::helper::copy_to_device(sb_handle, B.data(), b_gpu, ldb * n);
  blas::helper::copy_to_device(sb_handle, C.data(), c_gpu, ldc * n);

  /* C = alpha * sym(A) * B + beta * C, only the lower triangle of A is read */
  std::printf("Executing C = %g*A*B + %g*C  (symmetric A, %zu x %zu)\n", alpha, beta, m, n);
  blas::_symm(sb_handle, side, uplo, m, n, alpha, a_gpu, lda, b_gpu, ldb, beta, c_gpu, ldc);
  sb_handle.wait(blas::helper::copy_to_host(sb_handle, c_gpu, C.data(), ldc * n));

  size_t bad = 0;
  for (size_t j = 0; j < n; ++j)
    for (size_t i = 0; i < m; ++i) {
      float acc = 0.0f;
      for (size_t l = 0; l < k; ++l) {
        const float a_il = (i >= l) ? A[i + l * lda] : A[l + i * lda];  // mirror the stored triangle
        acc = std::fmaf(a_il, B[l + j * ldb], acc);
      }
      const float t_a = alpha * acc, t_b = beta * C0[i + j * ldc];
      const float want = t_a + t_b;
      // Large K may be cut into slabs that are summed separately: exact for the sample, a bound otherwise.
      const bool ok = big ? std::fabs(C[i + j * ldc] - want) <= 1e-5f * (1.0f + std::fabs(want)) : C[i + j * ldc] == want;
      if (!ok) ++bad;
    }
  std::printf("_symm: %zu mismatching elements against the sequential fma loop\n", bad);

  /* Solve L * X = alpha * B in place (left, lower, no transpose, non-unit diagonal) */
  std::printf("Executing B = %g * inv(L) * B  (%zu x %zu)\n", alpha, m, n);
  blas::_trsm(sb_handle, 'l', 'l', 'n', 'n', m, n, alpha, a_gpu, lda, b_gpu, ldb);
  std::vector<float> X(ldb * n);
  sb_handle.wait(blas::helper::copy_to_host(sb_handle, b_gpu, X.data(), ldb * n));

  double worst = 0.0;
  for (size_t j = 0; j < n; ++j)
    for (size_t i = 0; i < m; ++i) {
      double s = 0.0;
      for (size_t l = 0; l <= i; ++l) s += double(A[i + l * lda]) * double(X[l + j * ldb]);
      const double want = double(alpha) * double(B[i + j * ldb]);
      worst = std::fmax(worst, std::fabs(s - want) / (1.0 + std::fabs(want)));
    }
  std::printf("_trsm: largest relative residual %.3g\n", worst);

  /* Argument errors carry the reference's texts */
  bool threw = false;
  try {
    blas::_trsm(sb_handle, 'l', 'l', 'x', 'n', m, n, alpha, a_gpu, lda, b_gpu, ldb);
  } catch (const std::invalid_argument& e) {
    threw = std::string(e.what()) == "invalid Transpose argument";
  }

  if (bad == 0 && worst < 1e-4 && threw) {
    std::printf("PASS\n");
    return 0;
  }
  std::printf("FAIL (exception check %s)\n", threw ? "ok" : "failed");
  return 1;
}
