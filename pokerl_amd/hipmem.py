"""Minimal device-memory helper for callers that do not already own HBM buffers (tests, bench.py): hipMalloc / hipFree /
hipMemcpy through ctypes.  A torch / cupy user passes `tensor.data_ptr()` to the `_d` entry points instead."""
import ctypes as C

import numpy as np

_hip = None


def _lib():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
    return _hip


class DeviceBuffer:
    """hipMalloc'ed bytes on `device` (the caller's current device is left as it was)."""

    def __init__(self, nbytes, device=0):
        self.nbytes = int(nbytes)
        self.device = int(device)
        self.ptr = C.c_void_p()
        prev = C.c_int(-1)
        _lib().hipGetDevice(C.byref(prev))
        if _lib().hipSetDevice(self.device) != 0:
            raise RuntimeError("hipSetDevice(%d) failed" % self.device)
        rc = _lib().hipMalloc(C.byref(self.ptr), C.c_size_t(self.nbytes))
        if prev.value >= 0:
            _lib().hipSetDevice(prev.value)
        if rc != 0:
            raise MemoryError("hipMalloc(%d) failed: %d" % (self.nbytes, rc))

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        assert _lib().hipMemcpy(self.ptr, arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes), 1) == 0
        return self

    def download(self, dtype, count, offset_bytes=0):
        out = np.zeros(count, dtype)
        src = C.c_void_p(self.ptr.value + offset_bytes)
        assert _lib().hipMemcpy(out.ctypes.data_as(C.c_void_p), src, C.c_size_t(out.nbytes), 2) == 0
        return out

    def free(self):
        if self.ptr and self.ptr.value:
            _lib().hipFree(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedArray:
    """Owner of a pinned host allocation (pk_host_alloc = hipHostMalloc); `.array` is a numpy view of it.  Copies between a
    pinned array and the device run at PCIe line rate and hipMemcpyAsync does not block on them; the memory is released when
    this object is collected or free()d (the views must not outlive it: keep the object, not just `.array`)."""

    def __init__(self, shape, dtype):
        from . import _lib as L
        self._L = L
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        self.nbytes = max(n, 1)
        self._p = C.c_void_p()
        L.check(L.lib().pk_host_alloc(C.byref(self._p), C.c_size_t(self.nbytes)))
        buf = (C.c_char * self.nbytes).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self.array[...] = np.zeros((), dtype)

    def free(self):
        if self._p and self._p.value:
            self.array = None
            self._L.lib().pk_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float64):
    """A PinnedArray (use `.array`); see there."""
    return PinnedArray(shape, dtype)


class DeviceEvent:
    """A hipEvent_t for timing a region of a stream: pass `.handle` to VecGame.record_event (which records it on the
    handle's stream after completing deferred work); elapsed_ms(start, stop) waits for `stop`."""

    def __init__(self):
        self.handle = C.c_void_p()
        if _lib().hipEventCreate(C.byref(self.handle)) != 0:
            raise RuntimeError("hipEventCreate failed")

    @staticmethod
    def elapsed_ms(start, stop):
        if _lib().hipEventSynchronize(stop.handle) != 0:
            raise RuntimeError("hipEventSynchronize failed")
        ms = C.c_float(0.0)
        if _lib().hipEventElapsedTime(C.byref(ms), start.handle, stop.handle) != 0:
            raise RuntimeError("hipEventElapsedTime failed")
        return float(ms.value)

    def destroy(self):
        if self.handle and self.handle.value:
            _lib().hipEventDestroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
