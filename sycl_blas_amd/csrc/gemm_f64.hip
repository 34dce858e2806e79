// fp64 back end: instantiates the MFMA tile kernels for double storage.
#include "gemm_dispatch.hpp"

namespace sbamd {

hipError_t run_gemm_f64(Handle* h, const Problem<double, double>& p) { return run_gemm_real<double>(h, p); }

hipError_t run_scale_f64(Handle* h, double* c, int64_t rows, int64_t cols, int64_t ld,
                         int64_t stride, int64_t batch, double beta, bool assign_zero) {
  return run_scale_real<double>(h, c, rows, cols, ld, stride, batch, beta, assign_zero);
}

}  // namespace sbamd
