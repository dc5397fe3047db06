"""One banded deep-pyramid host encode and decode (16384^2, level 8) for a rocprofv3 --kernel-trace --memory-copy-trace run."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustyhgi_amd as H
from rustyhgi_amd import _ffi
L = _ffi.lib(); ctx = H.Context(0)
lut = np.zeros(256, np.uint8); err = np.zeros(1, np.uint8)
_ffi.check(L.hgi_linear_lut(2, lut.ctypes.data, err.ctypes.data))
W = Hh = 16384
src = np.random.default_rng(2).integers(0, 256, (Hh, W), dtype=np.uint8); dst = np.empty_like(src); back = np.empty_like(src)
for _ in range(3):
    _ffi.check(L.hgi_encode_u8(ctx.handle, src.ctypes.data, W, Hh, 8, 1, lut.ctypes.data, dst.ctypes.data))
for _ in range(3):
    _ffi.check(L.hgi_decode_u8(ctx.handle, dst.ctypes.data, W, Hh, 8, 1, back.ctypes.data))
