"""Device context and device-resident arrays (thin wrappers over the C ABI).

Ownership: the Python wrapper owns device buffers through
``DeviceArray`` objects whose ``__del__`` returns the memory; kernels never
allocate; outputs go to caller-provided buffers.
"""
import ctypes as C
import threading

import numpy as np

from . import _lib
from ._lib import F32, F64

_DT = {np.dtype(np.float32): F32, np.dtype(np.float64): F64}


def dtype_code(dtype):
    try:
        return _DT[np.dtype(dtype)]
    except KeyError:
        raise ValueError(f"dtype must be float32 or float64, got {dtype}") from None


class Event:
    def __init__(self, ctx):
        self.ctx = ctx
        h = C.c_void_p()
        ctx._call("tramp_event_create", C.byref(h))
        self.handle = h

    def record(self):
        self.ctx._call("tramp_event_record", self.handle)
        return self

    def elapsed_ms(self, stop):
        ms = C.c_float()
        self.ctx._call("tramp_event_elapsed_ms", self.handle, stop.handle, C.byref(ms))
        return ms.value

    def __del__(self):
        try:
            self.ctx.release("tramp_event_destroy", self.handle)
        except Exception:
            pass
        self.handle = None


_tls = threading.local()


class Context:
    """One per GPU: a HIP stream plus the reduction workspace (tramp_ctx)."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        h = C.c_void_p()
        rc = self.lib.tramp_ctx_create(int(device), C.byref(h))
        if rc != _lib.OK:
            raise _lib.TrampHipError(rc, f"tramp_ctx_create(device={device})")
        self.handle = h
        self.device = int(device)
        # tramp_ctx_create made `device` HIP's current device on this thread without telling the cache below:
        # a context this thread drove before must call tramp_ctx_make_current again before its next launch
        _tls.current = None

    # -- plumbing
    def _call(self, name, *args):
        """Call a C entry point.  DeviceArray arguments are passed as their device pointers and stay referenced
        for the duration of the call (a temporary such as ``ctx.array(x)`` is therefore safe to pass directly;
        ``ctx.array(x).ptr`` is not - the array dies, and is freed, before the launch is issued)."""
        raw = [a.ptr if isinstance(a, DeviceArray) else a for a in args]
        if getattr(_tls, "current", None) is not self:
            # HIP's current device is per host thread: switch when this thread last drove another context
            self.lib.tramp_ctx_make_current(self.handle)
            _tls.current = self
        if name == "tramp_graph_begin":
            self._capturing = True
        try:
            rc = getattr(self.lib, name)(self.handle, *raw)
        finally:
            if name == "tramp_graph_end":
                self._capturing = False
                self._flush_released()
        if rc != _lib.OK:
            if name == "tramp_graph_begin":
                self._capturing = False
            _lib.check(rc, name, self.handle)

    # Destructors of wrapper objects (channel handles, events, communicators) run whenever Python's garbage collector
    # decides - also in the middle of a stream capture, where their hipFree / synchronisation would invalidate the
    # capture.  They hand the destroy call to release(): executed at once, or right after tramp_graph_end.
    _capturing = False

    def release(self, fn_name, handle):
        if not handle or not self.handle:
            return
        if self._capturing:
            self.__dict__.setdefault("_released", []).append((fn_name, handle))
        else:
            getattr(self.lib, fn_name)(self.handle, handle)

    def _flush_released(self):
        pending, self._released = self.__dict__.get("_released", []), []
        for fn_name, handle in pending:
            try:
                getattr(self.lib, fn_name)(self.handle, handle)
            except Exception:                   # noqa: BLE001
                pass

    def sync(self):
        self._call("tramp_sync")

    def device_name(self):
        buf = C.create_string_buffer(256)
        self._call("tramp_device_name", buf, 256)
        return buf.value.decode()

    def mem_info(self):
        f, t = C.c_size_t(), C.c_size_t()
        self._call("tramp_mem_info", C.byref(f), C.byref(t))
        return f.value, t.value

    def stream(self):
        s = C.c_void_p()
        self._call("tramp_ctx_stream", C.byref(s))
        return s.value

    def event(self):
        return Event(self)

    def prior_hint_stats(self):
        "(producers that left fused prior moments, prior updates that consumed them) - csrc/prior_hint.h"
        a, u = C.c_ulonglong(), C.c_ulonglong()
        self._call("tramp_prior_hint_stats", C.byref(a), C.byref(u))
        return a.value, u.value

    def resident_covers(self, dtype, n, batch):
        "do updates of `batch` instances of n components take the resident single-pass form? (csrc/ew_resident.h)"
        yes = C.c_int()
        self._call("tramp_resident_covers", _DT[np.dtype(dtype)], int(n), int(batch), C.byref(yes))
        return bool(yes.value)

    def resident_runs(self):
        "factor updates that took the resident single-pass pipeline since the context was made - csrc/ew_resident.h"
        r = C.c_ulonglong()
        self._call("tramp_resident_stats", C.byref(r))
        return r.value

    def factorize_peak_bytes(self):
        "device work space the last tramp_linear_factorize of this context held at most (bytes)"
        b = C.c_size_t()
        self._call("tramp_factorize_peak_bytes", C.byref(b))
        return b.value

    def coop_runs(self):
        "factor updates that took the one-launch form for a few large instances (csrc/ew_kernels.hip: ew_coop)"
        r = C.c_ulonglong()
        self._call("tramp_coop_stats", C.byref(r))
        return r.value

    def resident_gave_up(self):
        """True once a resident kernel of this context gave up a hand-off (its deadline passed: the affected instances came
        back NaN) - every later update takes the tile kernels (include/tramp_hip.h: tramp_resident_errors)"""
        r = C.c_ulonglong()
        self._call("tramp_resident_errors", C.byref(r))
        return bool(r.value)

    # -- int8-digit copies of resident matrices (csrc/gemm_sliced.h; opt-in engine of the dense passes)
    def attach_digits(self, mat, digits=0, which=3):
        """Give the resident matrix `mat` [R, C] a copy in signed radix-256 digits: every tramp_matvec_nt / _nn pass over
        it then runs on the int8 matrix cores (digits: 0 = 7 for f64, 4 for f32; which:
        1 row-dot form, 2 column form, 3 both).  Detach before the matrix changes or is released."""
        R, Cc = mat.shape
        self._call("tramp_matrix_attach_digits", _DT[np.dtype(mat.dtype)], int(R), int(Cc), mat.ptr, int(digits), int(which))

    def detach_digits(self, mat=None):
        self._call("tramp_matrix_detach_digits", None if mat is None else mat.ptr)

    def digits_info(self, mat):
        "(digits, forms, device bytes) of the digit copy attached to `mat`; digits == 0: none"
        dg, w, b = C.c_int(), C.c_int(), C.c_size_t()
        self._call("tramp_matrix_digits_info", mat.ptr, C.byref(dg), C.byref(w), C.byref(b))
        return dg.value, w.value, b.value

    def sliced_runs(self):
        "dense passes that ran on a digit copy since the context was made"
        r = C.c_ulonglong()
        self._call("tramp_sliced_stats", C.byref(r))
        return r.value

    def close(self):
        if self.handle:
            self.lib.tramp_ctx_destroy(self.handle)
            self.handle = None
        if getattr(_tls, "current", None) is self:
            _tls.current = None

    # -- arrays
    def empty(self, shape, dtype=np.float64):
        return DeviceArray(self, shape, dtype)

    def zeros(self, shape, dtype=np.float64):
        a = DeviceArray(self, shape, dtype)
        self._call("tramp_memset", a.ptr, 0, a.nbytes)
        return a

    def full(self, shape, value, dtype=np.float64):
        a = DeviceArray(self, shape, dtype)
        self._call("tramp_fill", dtype_code(dtype), a.ptr, a.size, float(value))
        return a

    def array(self, host, dtype=None):
        host = np.ascontiguousarray(host, dtype=dtype if dtype is not None else None)
        if host.dtype not in _DT and host.dtype != np.uint8:
            host = host.astype(np.float64)
        a = DeviceArray(self, host.shape, host.dtype)
        a.set(host)
        return a

    def scalars(self, values, batch=None):
        "double[batch] device array of message precisions / variances."
        v = np.atleast_1d(np.asarray(values, dtype=np.float64))
        if batch is not None and v.shape[0] != batch:
            if v.shape[0] != 1:
                raise ValueError(f"expected {batch} scalars, got {v.shape[0]}")
            v = np.repeat(v, batch)
        return self.array(v, np.float64)

    def randn(self, shape, dtype=np.float64, seed=0, scale=1.0):
        a = DeviceArray(self, shape, dtype)
        self._call("tramp_randn", dtype_code(dtype), a.ptr, a.size, int(seed), float(scale))
        return a


class DeviceArray:
    """A contiguous row-major array in HBM.  Vectors are [batch, n] (or [n])."""

    def __init__(self, ctx, shape, dtype=np.float64, ptr=None, owner=True):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.size = int(np.prod(self.shape)) if self.shape else 1
        self.nbytes = self.size * self.dtype.itemsize
        self._owner = owner and ptr is None
        if ptr is None:
            p = C.c_void_p()
            ctx._call("tramp_buf_alloc", self.nbytes, C.byref(p))
            ptr = p.value
        self.ptr = ptr

    @property
    def ndim(self):
        return len(self.shape)

    def set(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.size != self.size:
            raise ValueError(f"size mismatch: device {self.shape} vs host {host.shape}")
        self.ctx._call("tramp_h2d", self.ptr, host.ctypes.data, self.nbytes)
        return self

    def get(self):
        out = np.empty(self.shape, dtype=self.dtype)
        self.ctx._call("tramp_d2h", out.ctypes.data, self.ptr, self.nbytes)
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.get()
        return a.astype(dtype) if dtype is not None else a

    def copy_from(self, other):
        if other.nbytes != self.nbytes:
            raise ValueError("size mismatch")
        self.ctx._call("tramp_d2d", self.ptr, other.ptr, self.nbytes)
        return self

    def copy(self):
        return DeviceArray(self.ctx, self.shape, self.dtype).copy_from(self)

    def fill(self, value):
        if self.dtype == np.uint8:
            self.ctx._call("tramp_memset", self.ptr, int(value), self.nbytes)
        else:
            self.ctx._call("tramp_fill", dtype_code(self.dtype), self.ptr, self.size, float(value))
        return self

    def astype(self, dtype):
        out = DeviceArray(self.ctx, self.shape, dtype)
        self.ctx._call("tramp_convert", dtype_code(dtype), out.ptr, dtype_code(self.dtype), self.ptr,
                       self.size)
        return out

    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        v = DeviceArray(self.ctx, shape, self.dtype, ptr=self.ptr, owner=False)
        if v.size != self.size:
            raise ValueError("cannot reshape")
        v._base = self      # keep the owner alive
        return v

    def row(self, b):
        "View of instance b of a [batch, n] array."
        n = self.shape[-1]
        v = DeviceArray(self.ctx, (n,), self.dtype, ptr=self.ptr + b * n * self.dtype.itemsize,
                        owner=False)
        v._base = self
        return v

    def __del__(self):
        try:
            if self._owner and self.ptr and self.ctx.handle:
                self.ctx.lib.tramp_buf_free(self.ctx.handle, self.ptr)
        except Exception:
            pass
        self.ptr = None

    def __repr__(self):
        return f"DeviceArray(shape={self.shape}, dtype={self.dtype}, device={self.ctx.device})"


_default = {}


def default_context(device=0):
    """Process-wide context for `device` (created on first use)."""
    if device not in _default:
        _default[device] = Context(device)
    return _default[device]
