"""Host-side mirror of the reference's GEMM interface over the C ABI.

Names, argument order and error behaviour follow the reference so that tests
read like the reference's own:

  blas::SB_Handle                      include/sb_handle/portblas_handle.h:46-200
  blas::_gemm / _gemm_batched /
        _gemm_strided_batched          include/interface/blas3_interface.h:86-123
  blas::gemm_batch_type_t              include/operations/blas3_trees.h:59
  blas::helper::allocate /
        copy_to_device / copy_to_host  include/portblas_helper.h:54-67, :139-191
  BufferIterator `+ offset`            include/container/sycl_iterator.h:33-182

std::invalid_argument("invalid _TransA") etc. (src/interface/gemm_interface.hpp:144-165)
become ValueError with the same text.  Every call enqueues on the handle's HIP
stream and returns a list with one event, as the reference returns a vector of
sycl events.
"""
import ctypes as C
import enum

import numpy as np

from . import _lib


class gemm_batch_type_t(enum.IntEnum):
    strided = 0
    interleaved = 1


class SbamdError(RuntimeError):
    """A HIP or allocation failure reported by the library (never an argument error)."""


_ARG_ERRORS = {
    _lib.INVALID_UPLO: "invalid _uplo",
    _lib.INVALID_SIDE: "invalid _side",
    _lib.TRSM_INVALID_SIZE: "invalid matrix size argument",
    _lib.TRSM_INVALID_SIDE: "invalid Side argument",
    _lib.TRSM_INVALID_UPLO: "invalid Triangle argument",
    _lib.TRSM_INVALID_TRANS: "invalid Transpose argument",
    _lib.TRSM_INVALID_DIAG: "invalid Diagonal argument",
    _lib.INVALID_TRANSA: "invalid _TransA",
    _lib.INVALID_TRANSB: "invalid _TransB",
    _lib.INVALID_STRIDEC: "invalid _stridec",
    _lib.INVALID_STRIDEA: "invalid _stridea",
    _lib.INVALID_STRIDEB: "invalid _strideb",
}

# storage dtype name -> (itemsize, sbamd_fill_uniform code)
_DTYPES = {"float32": (4, 0), "float64": (8, 1), "float16": (2, 2), "bfloat16": (2, 3)}


def _dtype_name(dtype):
    name = dtype if isinstance(dtype, str) else np.dtype(dtype).name
    if name not in _DTYPES:
        raise TypeError(f"unsupported element type {name}")
    return name


def _check(handle, status):
    if status == _lib.OK:
        return
    if status in _ARG_ERRORS:
        raise ValueError(_ARG_ERRORS[status])
    if status == _lib.INVALID_VALUE:
        raise ValueError("invalid value")
    if status == _lib.NOT_SUPPORTED:
        raise NotImplementedError("not supported by the HIP back end")
    detail = ""
    if handle is not None and handle._h:
        detail = ": " + _lib.load().sbamd_last_error(handle._h).decode()
    raise SbamdError(_lib.status_string(status) + detail)


class Event:
    """One recorded point on the handle's stream (stands in for sycl::event)."""

    def __init__(self, raw):
        self._e = raw

    def wait(self):
        st = _lib.load().sbamd_event_synchronize(self._e)
        if st != _lib.OK:
            raise SbamdError(_lib.status_string(st))

    def elapsed_ms(self, later):
        ms = C.c_float()
        st = _lib.load().sbamd_event_elapsed_ms(self._e, later._e, C.byref(ms))
        if st != _lib.OK:
            raise SbamdError(_lib.status_string(st))
        return ms.value

    def __del__(self):
        try:
            if self._e:
                _lib.load().sbamd_event_destroy(self._e)
                self._e = None
        except Exception:
            pass


class DeviceBuffer:
    """Non-owning view: device address + element type (+ element offset via `+`).

    Plays the part of BufferIterator<T> / a USM pointer: `buf + n` is a view
    that starts n elements later and shares the allocation.
    """

    def __init__(self, ptr, dtype, count, owner=None):
        self.ptr = int(ptr)
        self.dtype = _dtype_name(dtype)
        self.count = int(count)
        self._owner = owner  # keeps the allocation alive

    @property
    def itemsize(self):
        return _DTYPES[self.dtype][0]

    def __add__(self, n):
        n = int(n)
        return DeviceBuffer(self.ptr + n * self.itemsize, self.dtype, self.count - n,
                            self._owner if self._owner is not None else self)

    @classmethod
    def from_pointer(cls, ptr, dtype, count, owner=None):
        return cls(ptr, dtype, count, owner)


class _Allocation(DeviceBuffer):
    def __init__(self, handle, ptr, dtype, count):
        super().__init__(ptr, dtype, count)
        self._handle = handle

    def free(self):
        if self.ptr:
            # by device, not by handle: the allocation may outlive the handle it came from
            _lib.load().sbamd_device_free(self._handle.device, self.ptr)
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class SB_Handle:
    """Owns the HIP stream all calls are enqueued on (reference: one sycl::queue)."""

    def __init__(self, device=0, stream=None):
        self._h = None
        lib = _lib.load()
        raw = C.c_void_p()
        st = lib.sbamd_create(C.byref(raw), int(device), C.c_void_p(stream) if stream else None)
        if st != _lib.OK:
            raise SbamdError("sbamd_create: " + _lib.status_string(st))
        self._h = raw
        self.device = int(device)
        self._pin = []  # host arrays of uploads that may still be in flight

    def close(self):
        if self._h:
            _lib.load().sbamd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # SB_Handle::wait() / wait(events)  (portblas_handle.h:175-190)
    def wait(self, events=None):
        if events is None:
            _check(self, _lib.load().sbamd_wait(self._h))
            self._pin = []  # the stream has drained: every queued upload has read its host array
            return
        if isinstance(events, Event):
            events = [events]
        for ev in events:
            ev.wait()

    def get_num_compute_units(self):
        out = C.c_int()
        _check(self, _lib.load().sbamd_get_num_compute_units(self._h, C.byref(out)))
        return out.value

    def get_stream(self):
        out = C.c_void_p()
        _check(self, _lib.load().sbamd_get_stream(self._h, C.byref(out)))
        return out.value

    def record_event(self):
        raw = C.c_void_p()
        _check(self, _lib.load().sbamd_event_record(self._h, C.byref(raw)))
        return Event(raw)

    def depends_on(self, dependencies):
        for ev in dependencies or ():
            _check(self, _lib.load().sbamd_stream_wait_event(self._h, ev._e))

    def set_kernel(self, kernel, param=0):
        _check(self, _lib.load().sbamd_set_kernel(self._h, int(kernel), int(param)))

    def last_kernel(self):
        return _lib.load().sbamd_last_kernel(self._h)

    def last_splitk(self):
        """(depth, k_chunk) of the last call's split-K cut; depth 1 = not split."""
        depth, chunk = C.c_int64(), C.c_int64()
        _check(self, _lib.load().sbamd_last_splitk(self._h, C.byref(depth), C.byref(chunk)))
        return depth.value, chunk.value

    def last_tile(self):
        """(tile rows, tile columns, fused) of the last fp32 / fp64 call's matrix-instruction kernel; (0, 0, False) = none."""
        tm, tn, fused = C.c_int(), C.c_int(), C.c_int()
        _check(self, _lib.load().sbamd_last_tile(self._h, C.byref(tm), C.byref(tn), C.byref(fused)))
        return tm.value, tn.value, bool(fused.value)

    def last_trsm_launches(self):
        """(diagonal super-block launches, panel-update launches) of the last _trsm; (0, 0) = block by block."""
        diag, panel = C.c_int64(), C.c_int64()
        _check(self, _lib.load().sbamd_last_trsm_launches(self._h, C.byref(diag), C.byref(panel)))
        return diag.value, panel.value


class helper:
    """blas::helper allocation and copy routines (include/portblas_helper.h)."""

    @staticmethod
    def allocate(sb_handle, dtype, count):
        name = _dtype_name(dtype)
        raw = C.c_void_p()
        nbytes = max(int(count), 1) * _DTYPES[name][0]
        _check(sb_handle, _lib.load().sbamd_malloc(sb_handle._h, C.byref(raw), nbytes))
        return _Allocation(sb_handle, raw.value, name, count)

    @staticmethod
    def deallocate(buf):
        owner = buf._owner if buf._owner is not None else buf
        if isinstance(owner, _Allocation):
            owner.free()

    @staticmethod
    def copy_to_device(sb_handle, host, buf, count=None):
        host = np.ascontiguousarray(host)
        count = host.size if count is None else int(count)
        nbytes = count * buf.itemsize
        assert host.itemsize == buf.itemsize and host.size >= count
        # keep the host array alive until the copy has run: the returned event carries it, and the
        # handle holds it until its stream is next drained (SB_Handle.wait())
        sb_handle._pin.append(host)
        if len(sb_handle._pin) > 64:
            _check(sb_handle, _lib.load().sbamd_wait(sb_handle._h))
            sb_handle._pin = [host]
        _check(sb_handle, _lib.load().sbamd_copy_to_device(sb_handle._h, buf.ptr, host.ctypes.data, nbytes))
        ev = sb_handle.record_event()
        ev._keep = host
        return ev

    @staticmethod
    def copy_to_host(sb_handle, buf, host, count=None):
        assert host.flags["C_CONTIGUOUS"] or host.flags["F_CONTIGUOUS"]
        count = host.size if count is None else int(count)
        nbytes = count * buf.itemsize
        assert host.itemsize == buf.itemsize and host.size >= count
        _check(sb_handle, _lib.load().sbamd_copy_to_host(sb_handle._h, host.ctypes.data, buf.ptr, nbytes))
        return sb_handle.record_event()

    @staticmethod
    def fill_uniform(sb_handle, buf, count, seed, lo=-2.0, hi=5.0):
        code = _DTYPES[buf.dtype][1]
        _check(sb_handle, _lib.load().sbamd_fill_uniform(sb_handle._h, buf.ptr, int(count), code,
                                                          int(seed), float(lo), float(hi)))
        return sb_handle.record_event()


def uniform_reference(count, seed, lo=-2.0, hi=5.0, start=0):
    """Host replay of sbamd_fill_uniform for elements [start, start + count), as float64."""
    i = np.arange(start, start + count, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + i
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    u = (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return lo + (hi - lo) * u


# (input storage, output storage) -> C-ABI prefix.  The pairs are the
# reference's instantiations (src/interface/blas3/gemm.cpp.in:33-137) plus
# bfloat16 -> float.
_PREFIX = {
    ("float32", "float32"): ("sbamd_s", C.c_float),
    ("float64", "float64"): ("sbamd_d", C.c_double),
    ("float16", "float16"): ("sbamd_h", C.c_float),
    ("float16", "float32"): ("sbamd_hs", C.c_float),
    ("bfloat16", "float32"): ("sbamd_bs", C.c_float),
}


def _pick(a, b, c):
    if a.dtype != b.dtype:
        raise TypeError("A and B must share one element type")
    key = (a.dtype, c.dtype)
    if key not in _PREFIX:
        raise TypeError(f"no GEMM for {a.dtype} inputs with {c.dtype} output")
    return _PREFIX[key]


def _ch(x):
    return x.encode()[:1] if isinstance(x, str) else bytes([x])[:1]


def _gemm(sb_handle, _TransA, _TransB, _M, _N, _K, _alpha, a_, _lda, b_, _ldb, _beta, _C, _ldc,
          _dependencies=()):
    """blas::_gemm (include/interface/blas3_interface.h:86-95)."""
    prefix, scalar = _pick(a_, b_, _C)
    sb_handle.depends_on(_dependencies)
    fn = getattr(_lib.load(), prefix + "gemm")
    st = fn(sb_handle._h, _ch(_TransA), _ch(_TransB), int(_M), int(_N), int(_K), scalar(_alpha), a_.ptr,
            int(_lda), b_.ptr, int(_ldb), scalar(_beta), _C.ptr, int(_ldc))
    _check(sb_handle, st)
    return [sb_handle.record_event()]


def _gemm_batched(sb_handle, _TransA, _TransB, _M, _N, _K, _alpha, a_, _lda, b_, _ldb, _beta, _C,
                  _ldc, batch_size, batch_type=gemm_batch_type_t.strided, _dependencies=()):
    """blas::_gemm_batched (include/interface/blas3_interface.h:97-109)."""
    prefix, scalar = _pick(a_, b_, _C)
    sb_handle.depends_on(_dependencies)
    fn = getattr(_lib.load(), prefix + "gemm_batched")
    st = fn(sb_handle._h, _ch(_TransA), _ch(_TransB), int(_M), int(_N), int(_K), scalar(_alpha), a_.ptr, int(_lda),
            b_.ptr, int(_ldb), scalar(_beta), _C.ptr, int(_ldc), int(batch_size), int(batch_type))
    _check(sb_handle, st)
    return [sb_handle.record_event()]


def _gemm_strided_batched(sb_handle, _TransA, _TransB, _M, _N, _K, _alpha, a_, _lda, _stridea, b_,
                          _ldb, _strideb, _beta, _C, _ldc, _stridec, batch_size, _dependencies=()):
    """blas::_gemm_strided_batched (include/interface/blas3_interface.h:111-123)."""
    prefix, scalar = _pick(a_, b_, _C)
    sb_handle.depends_on(_dependencies)
    fn = getattr(_lib.load(), prefix + "gemm_strided_batched")
    st = fn(sb_handle._h, _ch(_TransA), _ch(_TransB), int(_M), int(_N), int(_K), scalar(_alpha), a_.ptr,
            int(_lda), int(_stridea), b_.ptr, int(_ldb), int(_strideb), scalar(_beta), _C.ptr,
            int(_ldc), int(_stridec), int(batch_size))
    _check(sb_handle, st)
    return [sb_handle.record_event()]


_REAL = {"float32": ("s", C.c_float), "float64": ("d", C.c_double)}


def _symm(sb_handle, _side, _uplo, _M, _N, _alpha, a_, _lda, b_, _ldb, _beta, _C, _ldc, _dependencies=()):
    """blas::_symm (include/interface/blas3_interface.h:136-145, src/interface/symm_interface.hpp:34-73)."""
    if not (a_.dtype == b_.dtype == _C.dtype) or _C.dtype not in _REAL:
        raise TypeError("_symm exists for float and double")
    prefix, scalar = _REAL[_C.dtype]
    sb_handle.depends_on(_dependencies)
    st = getattr(_lib.load(), f"sbamd_{prefix}symm")(sb_handle._h, _ch(_side), _ch(_uplo), int(_M), int(_N),
                                                     scalar(_alpha), a_.ptr, int(_lda), b_.ptr, int(_ldb),
                                                     scalar(_beta), _C.ptr, int(_ldc))
    _check(sb_handle, st)
    return [sb_handle.record_event()]


def _trsm(sb_handle, side, uplo, trans, diag, M, N, alpha, A, lda, B, ldb, _dependencies=()):
    """blas::_trsm (include/interface/blas3_interface.h:125-134, src/interface/trsm_interface.hpp:106-396)."""
    if A.dtype != B.dtype or B.dtype not in _REAL:
        raise TypeError("_trsm exists for float and double")
    prefix, scalar = _REAL[B.dtype]
    sb_handle.depends_on(_dependencies)
    st = getattr(_lib.load(), f"sbamd_{prefix}trsm")(sb_handle._h, _ch(side), _ch(uplo), _ch(trans), _ch(diag), int(M),
                                                     int(N), scalar(alpha), A.ptr, int(lda), B.ptr, int(ldb))
    _check(sb_handle, st)
    return [sb_handle.record_event()]
