// portblas_compat.hpp -- header-only C++ face of the GEMM engine with the
// reference's names and argument order, so code written against portBLAS'
// GEMM entry points compiles against libsbamd.so with only its buffer creation
// changed.  Everything forwards to the C ABI of sbamd_gemm.h.
//
//   blas::SB_Handle                 include/sb_handle/portblas_handle.h:46-200
//   blas::_gemm, _gemm_batched,
//         _gemm_strided_batched     include/interface/blas3_interface.h:86-123
//   blas::gemm_batch_type_t         include/operations/blas3_trees.h:59
//   blas::BufferIterator<T>, `+`    include/container/sycl_iterator.h:33-182
//   blas::make_sycl_iterator_buffer include/container/sycl_iterator.h:227-272
//   blas::helper::allocate / deallocate / copy_to_device / copy_to_host
//                                   include/portblas_helper.h:54-67, :139-191
//
// Differences a caller sees: there is no sycl::queue -- SB_Handle is built from
// a HIP device ordinal (and optionally a hipStream_t); buffers are created
// through the handle; events wrap HIP events.  Argument errors throw the same
// std::invalid_argument texts as src/interface/gemm_interface.hpp:144-165.
#ifndef PORTBLAS_COMPAT_HPP
#define PORTBLAS_COMPAT_HPP

#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "sbamd_gemm.h"

namespace blas {

enum class gemm_batch_type_t : int { strided = 0, interleaved = 1 };

// 16-bit storage types (bit patterns; conversion is the caller's business).
struct half_t {
  uint16_t bits;
};
struct bfloat16_t {
  uint16_t bits;
};

namespace detail {
inline void check(int status, sbamd_handle_t h = nullptr) {
  if (status == SBAMD_OK) return;
  if ((status >= SBAMD_INVALID_TRANSA && status <= SBAMD_INVALID_STRIDEB) ||
      (status >= SBAMD_INVALID_UPLO && status <= SBAMD_TRSM_INVALID_DIAG))
    throw std::invalid_argument(sbamd_status_string(status));
  if (status == SBAMD_INVALID_VALUE) throw std::invalid_argument("invalid value");
  std::string msg = sbamd_status_string(status);
  if (h) msg += std::string(": ") + sbamd_last_error(h);
  throw std::runtime_error(msg);
}
struct EventBox {
  sbamd_event_t ev = nullptr;
  ~EventBox() {
    if (ev) sbamd_event_destroy(ev);
  }
};
}  // namespace detail

// Stands in for sycl::event; copies share one HIP event.
class Event {
 public:
  Event() = default;
  explicit Event(sbamd_event_t raw) : box_(std::make_shared<detail::EventBox>()) { box_->ev = raw; }
  void wait() const {
    if (box_ && box_->ev) detail::check(sbamd_event_synchronize(box_->ev));
  }
  sbamd_event_t raw() const { return box_ ? box_->ev : nullptr; }

 private:
  std::shared_ptr<detail::EventBox> box_;
};

// Device memory + element offset with pointer arithmetic; copies share the
// allocation (the reference's copies share one ref-counted sycl::buffer).
template <typename T>
class BufferIterator {
 public:
  using scalar_t = T;
  BufferIterator() = default;
  BufferIterator(std::shared_ptr<void> owner, T* base, std::ptrdiff_t offset = 0)
      : owner_(std::move(owner)), base_(base), offset_(offset) {}
  T* get() const { return base_ + offset_; }
  std::ptrdiff_t get_offset() const { return offset_; }
  BufferIterator operator+(std::ptrdiff_t n) const { return BufferIterator(owner_, base_, offset_ + n); }
  BufferIterator operator-(std::ptrdiff_t n) const { return BufferIterator(owner_, base_, offset_ - n); }
  BufferIterator& operator+=(std::ptrdiff_t n) {
    offset_ += n;
    return *this;
  }
  // BufferIterator<T> -> BufferIterator<const T>
  template <typename U = T, typename = typename std::enable_if<!std::is_const<U>::value>::type>
  operator BufferIterator<const U>() const {
    return BufferIterator<const U>(owner_, base_, offset_);
  }

 private:
  std::shared_ptr<void> owner_;
  T* base_ = nullptr;
  std::ptrdiff_t offset_ = 0;
};

class SB_Handle {
 public:
  using event_t = std::vector<Event>;

  explicit SB_Handle(int device = 0, void* hip_stream = nullptr) : device_(device) {
    detail::check(sbamd_create(&h_, device, hip_stream));
  }
  ~SB_Handle() {
    if (h_) sbamd_destroy(h_);
  }
  SB_Handle(const SB_Handle&) = delete;
  SB_Handle& operator=(const SB_Handle&) = delete;

  void wait() { detail::check(sbamd_wait(h_), h_); }
  void wait(const event_t& evs) {
    for (const Event& e : evs) e.wait();
  }
  void wait(const Event& ev) { ev.wait(); }
  template <typename first_event_t, typename... next_event_t>
  void wait(first_event_t first, next_event_t... rest) {
    wait(first);
    wait(rest...);
  }

  size_t get_num_compute_units() const {
    int cu = 0;
    detail::check(sbamd_get_num_compute_units(h_, &cu), h_);
    return size_t(cu);
  }
  size_t get_work_group_size() const { return 256; }  // min(256, max work-group size), portblas_helper.h:121-131
  bool has_local_memory() const { return true; }
  void* get_queue() const {  // the HIP stream plays the queue's part
    void* s = nullptr;
    detail::check(sbamd_get_stream(h_, &s), h_);
    return s;
  }
  sbamd_handle_t native() const { return h_; }
  int device() const { return device_; }

  // helpers used by the entry points below
  void depends_on(const event_t& deps) {
    for (const Event& e : deps)
      if (e.raw()) detail::check(sbamd_stream_wait_event(h_, e.raw()), h_);
  }
  event_t record() {
    sbamd_event_t raw = nullptr;
    detail::check(sbamd_event_record(h_, &raw), h_);
    return {Event(raw)};
  }

 private:
  sbamd_handle_t h_ = nullptr;
  int device_ = 0;
};

template <typename T>
BufferIterator<T> make_sycl_iterator_buffer(SB_Handle& sb_handle, size_t count) {
  void* raw = nullptr;
  detail::check(sbamd_malloc(sb_handle.native(), &raw, count * sizeof(T)), sb_handle.native());
  // The deleter knows the device, not the handle: like the reference's BufferIterator, whose
  // sycl::buffer lives on independently of the SB_Handle, a buffer may outlive the handle.
  const int device = sb_handle.device();
  std::shared_ptr<void> owner(raw, [device](void* p) { sbamd_device_free(device, p); });
  return BufferIterator<T>(owner, static_cast<T*>(raw));
}

namespace helper {
template <typename T>
BufferIterator<T> allocate(size_t count, SB_Handle& sb_handle) {
  return make_sycl_iterator_buffer<T>(sb_handle, count);
}
template <typename T>
void deallocate(BufferIterator<T>&, SB_Handle&) {}  // released with the last copy

template <typename T>
SB_Handle::event_t copy_to_device(SB_Handle& sb_handle, const T* src, BufferIterator<T> dst, size_t count) {
  detail::check(sbamd_copy_to_device(sb_handle.native(), dst.get(), src, count * sizeof(T)), sb_handle.native());
  return sb_handle.record();
}
template <typename T>
SB_Handle::event_t copy_to_device(SB_Handle& sb_handle, const T* src, T* dst, size_t count) {
  detail::check(sbamd_copy_to_device(sb_handle.native(), dst, src, count * sizeof(T)), sb_handle.native());
  return sb_handle.record();
}
template <typename T>
SB_Handle::event_t copy_to_host(SB_Handle& sb_handle, BufferIterator<T> src, T* dst, size_t count) {
  detail::check(sbamd_copy_to_host(sb_handle.native(), dst, src.get(), count * sizeof(T)), sb_handle.native());
  return sb_handle.record();
}
template <typename T>
SB_Handle::event_t copy_to_host(SB_Handle& sb_handle, const T* src, T* dst, size_t count) {
  detail::check(sbamd_copy_to_host(sb_handle.native(), dst, src, count * sizeof(T)), sb_handle.native());
  return sb_handle.record();
}
}  // namespace helper

namespace detail {
// containers: BufferIterator<T> or a raw device pointer (the reference's USM form)
template <typename T>
T* raw(const BufferIterator<T>& b) {
  return b.get();
}
template <typename T>
T* raw(T* p) {
  return p;
}
template <typename C>
struct elem;
template <typename T>
struct elem<BufferIterator<T>> {
  using type = typename std::remove_const<T>::type;
};
template <typename T>
struct elem<T*> {
  using type = typename std::remove_const<T>::type;
};

// (input element, output element) -> C-ABI entry points
template <typename TIn, typename TOut>
struct abi;
template <>
struct abi<float, float> {
  using scalar = float;
  static constexpr auto gemm = sbamd_sgemm;
  static constexpr auto symm = sbamd_ssymm;
  static constexpr auto trsm = sbamd_strsm;
  static constexpr auto strided = sbamd_sgemm_strided_batched;
  static int batched(sbamd_handle_t h, char ta, char tb, int64_t m, int64_t n, int64_t k, float al, const float* a,
                     int64_t lda, const float* b, int64_t ldb, float be, float* c, int64_t ldc, int64_t batch, int bt) {
    return sbamd_sgemm_batched(h, ta, tb, m, n, k, al, a, lda, b, ldb, be, c, ldc, batch, bt);
  }
};
template <>
struct abi<double, double> {
  using scalar = double;
  static constexpr auto gemm = sbamd_dgemm;
  static constexpr auto symm = sbamd_dsymm;
  static constexpr auto trsm = sbamd_dtrsm;
  static constexpr auto strided = sbamd_dgemm_strided_batched;
  static int batched(sbamd_handle_t h, char ta, char tb, int64_t m, int64_t n, int64_t k, double al, const double* a,
                     int64_t lda, const double* b, int64_t ldb, double be, double* c, int64_t ldc, int64_t batch,
                     int bt) {
    return sbamd_dgemm_batched(h, ta, tb, m, n, k, al, a, lda, b, ldb, be, c, ldc, batch, bt);
  }
};
#define SBAMD_ABI16(TIN, TOUT, PFX, COUT)                                                                         \
  template <>                                                                                                     \
  struct abi<TIN, TOUT> {                                                                                         \
    using scalar = float;                                                                                         \
    static int gemm(sbamd_handle_t h, char ta, char tb, int64_t m, int64_t n, int64_t k, float al, const TIN* a,  \
                    int64_t lda, const TIN* b, int64_t ldb, float be, TOUT* c, int64_t ldc) {                     \
      return PFX##gemm(h, ta, tb, m, n, k, al, reinterpret_cast<const uint16_t*>(a), lda,                         \
                       reinterpret_cast<const uint16_t*>(b), ldb, be, reinterpret_cast<COUT*>(c), ldc);           \
    }                                                                                                             \
    static int strided(sbamd_handle_t h, char ta, char tb, int64_t m, int64_t n, int64_t k, float al,             \
                       const TIN* a, int64_t lda, int64_t sa, const TIN* b, int64_t ldb, int64_t sb, float be,    \
                       TOUT* c, int64_t ldc, int64_t sc, int64_t batch) {                                         \
      return PFX##gemm_strided_batched(h, ta, tb, m, n, k, al, reinterpret_cast<const uint16_t*>(a), lda, sa,     \
                                       reinterpret_cast<const uint16_t*>(b), ldb, sb, be,                         \
                                       reinterpret_cast<COUT*>(c), ldc, sc, batch);                               \
    }                                                                                                             \
    static int batched(sbamd_handle_t h, char ta, char tb, int64_t m, int64_t n, int64_t k, float al,             \
                       const TIN* a, int64_t lda, const TIN* b, int64_t ldb, float be, TOUT* c, int64_t ldc,      \
                       int64_t batch, int bt) {                                                                   \
      return PFX##gemm_batched(h, ta, tb, m, n, k, al, reinterpret_cast<const uint16_t*>(a), lda,                 \
                               reinterpret_cast<const uint16_t*>(b), ldb, be, reinterpret_cast<COUT*>(c), ldc,    \
                               batch, bt);                                                                        \
    }                                                                                                             \
  };
SBAMD_ABI16(half_t, half_t, sbamd_h, uint16_t)
SBAMD_ABI16(half_t, float, sbamd_hs, float)
SBAMD_ABI16(bfloat16_t, float, sbamd_bs, float)
#undef SBAMD_ABI16
}  // namespace detail

// blas::_gemm (include/interface/blas3_interface.h:86-95)
template <typename sb_handle_t, typename container_0_t, typename container_1_t, typename container_2_t,
          typename element_t, typename index_t>
typename sb_handle_t::event_t _gemm(sb_handle_t& sb_handle, char _TransA, char _TransB, index_t _M, index_t _N,
                                    index_t _K, element_t _alpha, container_0_t a_, index_t _lda,
                                    container_1_t b_, index_t _ldb, element_t _beta, container_2_t _C,
                                    index_t _ldc, const typename sb_handle_t::event_t& _dependencies = {}) {
  using in_t = typename detail::elem<container_0_t>::type;
  using out_t = typename detail::elem<container_2_t>::type;
  using api = detail::abi<in_t, out_t>;
  sb_handle.depends_on(_dependencies);
  detail::check(api::gemm(sb_handle.native(), _TransA, _TransB, int64_t(_M), int64_t(_N), int64_t(_K),
                          typename api::scalar(_alpha), detail::raw(a_), int64_t(_lda), detail::raw(b_),
                          int64_t(_ldb), typename api::scalar(_beta), detail::raw(_C), int64_t(_ldc)),
                sb_handle.native());
  return sb_handle.record();
}

// blas::_gemm_batched (include/interface/blas3_interface.h:97-109)
template <typename sb_handle_t, typename container_0_t, typename container_1_t, typename container_2_t,
          typename element_t, typename index_t>
typename sb_handle_t::event_t _gemm_batched(sb_handle_t& sb_handle, char _TransA, char _TransB, index_t _M,
                                            index_t _N, index_t _K, element_t _alpha, container_0_t a_,
                                            index_t _lda, container_1_t b_, index_t _ldb, element_t _beta,
                                            container_2_t _C, index_t _ldc, index_t batch_size,
                                            gemm_batch_type_t batch_type = gemm_batch_type_t::strided,
                                            const typename sb_handle_t::event_t& _dependencies = {}) {
  using in_t = typename detail::elem<container_0_t>::type;
  using out_t = typename detail::elem<container_2_t>::type;
  using api = detail::abi<in_t, out_t>;
  sb_handle.depends_on(_dependencies);
  detail::check(api::batched(sb_handle.native(), _TransA, _TransB, int64_t(_M), int64_t(_N), int64_t(_K),
                             typename api::scalar(_alpha), detail::raw(a_), int64_t(_lda), detail::raw(b_),
                             int64_t(_ldb), typename api::scalar(_beta), detail::raw(_C), int64_t(_ldc),
                             int64_t(batch_size), int(batch_type)),
                sb_handle.native());
  return sb_handle.record();
}

// blas::_gemm_strided_batched (include/interface/blas3_interface.h:111-123)
template <typename sb_handle_t, typename container_0_t, typename container_1_t, typename container_2_t,
          typename element_t, typename index_t>
typename sb_handle_t::event_t _gemm_strided_batched(sb_handle_t& sb_handle, char _TransA, char _TransB, index_t _M,
                                                    index_t _N, index_t _K, element_t _alpha, container_0_t a_,
                                                    index_t _lda, index_t _stridea, container_1_t b_,
                                                    index_t _ldb, index_t _strideb, element_t _beta,
                                                    container_2_t _C, index_t _ldc, index_t _stridec,
                                                    index_t batch_size,
                                                    const typename sb_handle_t::event_t& _dependencies = {}) {
  using in_t = typename detail::elem<container_0_t>::type;
  using out_t = typename detail::elem<container_2_t>::type;
  using api = detail::abi<in_t, out_t>;
  sb_handle.depends_on(_dependencies);
  detail::check(api::strided(sb_handle.native(), _TransA, _TransB, int64_t(_M), int64_t(_N), int64_t(_K),
                             typename api::scalar(_alpha), detail::raw(a_), int64_t(_lda), int64_t(_stridea),
                             detail::raw(b_), int64_t(_ldb), int64_t(_strideb), typename api::scalar(_beta),
                             detail::raw(_C), int64_t(_ldc), int64_t(_stridec), int64_t(batch_size)),
                sb_handle.native());
  return sb_handle.record();
}

// blas::_symm (include/interface/blas3_interface.h:136-145), float and double
template <typename sb_handle_t, typename container_0_t, typename container_1_t, typename container_2_t,
          typename element_t, typename index_t>
typename sb_handle_t::event_t _symm(sb_handle_t& sb_handle, char _side, char _uplo, index_t _M, index_t _N,
                                    element_t _alpha, container_0_t a_, index_t _lda, container_1_t b_,
                                    index_t _ldb, element_t _beta, container_2_t _C, index_t _ldc,
                                    const typename sb_handle_t::event_t& _dependencies = {}) {
  using out_t = typename detail::elem<container_2_t>::type;
  using api = detail::abi<out_t, out_t>;
  sb_handle.depends_on(_dependencies);
  detail::check(api::symm(sb_handle.native(), _side, _uplo, int64_t(_M), int64_t(_N), out_t(_alpha),
                          detail::raw(a_), int64_t(_lda), detail::raw(b_), int64_t(_ldb), out_t(_beta),
                          detail::raw(_C), int64_t(_ldc)),
                sb_handle.native());
  return sb_handle.record();
}

// blas::_trsm (include/interface/blas3_interface.h:125-134), float and double
template <typename sb_handle_t, typename container_0_t, typename container_1_t, typename element_t,
          typename index_t>
typename sb_handle_t::event_t _trsm(sb_handle_t& sb_handle, char side, char uplo, char trans, char diag,
                                    index_t M, index_t N, element_t alpha, container_0_t A, index_t lda,
                                    container_1_t B, index_t ldb,
                                    const typename sb_handle_t::event_t& _dependencies = {}) {
  using out_t = typename detail::elem<container_1_t>::type;
  using api = detail::abi<out_t, out_t>;
  sb_handle.depends_on(_dependencies);
  detail::check(api::trsm(sb_handle.native(), side, uplo, trans, diag, int64_t(M), int64_t(N), out_t(alpha),
                          detail::raw(A), int64_t(lda), detail::raw(B), int64_t(ldb)),
                sb_handle.native());
  return sb_handle.record();
}

}  // namespace blas

#endif  // PORTBLAS_COMPAT_HPP
