"""Device planes placed for the HBM of MI355X (hgi_planes_alloc, include/hgi.h).

A launch that streams one buffer in and another out runs 4-5 % faster when the two lie in different physical regions
of the device's memory.  `Planes(ctx, bytes, count)` owns `count` device buffers whose neighbours in the list lie in
different regions (best effort, established by timing; `separated` says whether it was), so an
image -> grid -> image chain alternates through them.  No reference counterpart: the reference's buffers are `Vec<u8>`.
"""
import ctypes

from . import _ffi


class _CudaArray:
    """Minimal __cuda_array_interface__ holder so that torch can view a raw device pointer without copying."""

    def __init__(self, ptr, shape, owner):
        self.__cuda_array_interface__ = {"shape": tuple(int(s) for s in shape), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2, "strides": None}
        self._owner = owner        # keeps the planes alive as long as a view exists


class Planes:
    def __init__(self, ctx, nbytes, count):
        self._ctx, self.nbytes, self.count = ctx, int(nbytes), int(count)
        arr = (ctypes.c_void_p * self.count)()
        sep = ctypes.c_int(0)
        _ffi.check(_ffi.lib().hgi_planes_alloc(ctx.handle, self.nbytes, self.count, arr, ctypes.byref(sep)))
        self._arr = arr
        self.pointers = [int(p or 0) for p in arr]
        self.separated = bool(sep.value)
        # what the library found and did (hgi_planes_report; the ctx overwrites it at its next hgi_planes_alloc)
        self.report = (_ffi.lib().hgi_planes_report(ctx.handle) or b"").decode()

    def torch(self, index, shape):
        """uint8 CUDA tensor viewing plane `index` (no copy; valid while this object lives)."""
        import torch
        n = 1
        for s in shape:
            n *= int(s)
        if n > self.nbytes:
            raise ValueError("shape %r needs %d bytes, the plane has %d" % (tuple(shape), n, self.nbytes))
        return torch.as_tensor(_CudaArray(self.pointers[index], shape, self), device="cuda:%d" % self._ctx.device)

    def probe_ms(self, src_index, dst_index):
        """Milliseconds of one decode launch streaming plane src -> plane dst (overwrites dst)."""
        ms = ctypes.c_float(0)
        _ffi.check(_ffi.lib().hgi_probe_pair_u8_dev(self._ctx.handle, self.pointers[src_index], self.pointers[dst_index],
                                                    self.nbytes, ctypes.byref(ms)))
        return ms.value

    def close(self):
        if getattr(self, "_arr", None) is not None and self._ctx.handle:
            _ffi.check(_ffi.lib().hgi_planes_free(self._ctx.handle, self.count, self._arr))
        self._arr = None
        self.pointers = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
