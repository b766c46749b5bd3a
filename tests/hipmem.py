"""Plain device memory through ctypes on libamdhip64 (no torch): the GPU tests hand caller-owned device buffers to the
device-resident entry points exactly as a non-Python host would."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        h = C.CDLL("libamdhip64.so")
        h.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        h.hipFree.argtypes = [C.c_void_p]
        h.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        h.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        h.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        h.hipStreamSynchronize.argtypes = [C.c_void_p]
        h.hipStreamDestroy.argtypes = [C.c_void_p]
        _hip = h
    return _hip


class DevBuf:
    """A hipMalloc'ed buffer; .ptr is the device address."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        rc = hip().hipMalloc(C.byref(p), max(int(nbytes), 1))
        if rc != 0:
            raise MemoryError("hipMalloc(%d) failed: %d" % (nbytes, rc))
        self.ptr, self.nbytes = p.value, int(nbytes)

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        assert hip().hipMemcpy(b.ptr, a.ctypes.data, a.nbytes, 1) == 0
        return b

    def to_numpy(self, dtype, count):
        out = np.zeros(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        assert hip().hipMemcpy(out.ctypes.data, self.ptr, out.nbytes, 2) == 0  # blocking D2H on the null stream
        return out

    def free(self):
        if getattr(self, "ptr", None):
            hip().hipFree(self.ptr)
            self.ptr = None

    __del__ = free


def memset(buf, byte):
    """every byte of a DevBuf set to `byte` (blocking)"""
    assert hip().hipMemset(buf.ptr, int(byte), buf.nbytes) == 0
    assert hip().hipDeviceSynchronize() == 0


def device_sync():
    assert hip().hipDeviceSynchronize() == 0


def new_stream():
    s = C.c_void_p()
    assert hip().hipStreamCreate(C.byref(s)) == 0
    return s.value
