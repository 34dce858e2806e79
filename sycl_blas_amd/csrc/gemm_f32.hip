// fp32 back end: instantiates the MFMA tile kernels for float storage.
#include "gemm_dispatch.hpp"

namespace sbamd {

hipError_t run_gemm_f32(Handle* h, const Problem<float, float>& p) { return run_gemm_real<float>(h, p); }

hipError_t run_scale_f32(Handle* h, float* c, int64_t rows, int64_t cols, int64_t ld,
                         int64_t stride, int64_t batch, float beta, bool assign_zero) {
  return run_scale_real<float>(h, c, rows, cols, ld, stride, batch, beta, assign_zero);
}

}  // namespace sbamd
