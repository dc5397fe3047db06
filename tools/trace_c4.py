"""C4 (16384^2, level 8, High) through the fused path, for a rocprofv3 kernel trace: which launches make up the step."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustyhgi_amd as H
from rustyhgi_amd import _ffi
L = _ffi.lib()
ctx = H.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
W = Hh = 16384; n = W * Hh
lut = np.zeros(256, np.uint8); err = np.zeros(1, np.uint8)
_ffi.check(L.hgi_linear_lut(3, lut.ctypes.data, err.ctypes.data))
img = torch.empty(n, dtype=torch.uint8, device="cuda"); grid = torch.empty_like(img); out = torch.empty_like(img)
_ffi.check(L.hgi_synth_u8_dev(ctx.handle, _ffi.SYNTH_RAMP, 0x48474933 + 4, 0, W, Hh, img.data_ptr(), 1, n))
for _ in range(8):
    _ffi.check(L.hgi_encode_u8_dev(ctx.handle, img.data_ptr(), W, Hh, 8, 1, lut.ctypes.data, grid.data_ptr(), 1, n))
    torch.cuda.synchronize()
    _ffi.check(L.hgi_decode_u8_dev(ctx.handle, grid.data_ptr(), W, Hh, 8, 1, out.data_ptr(), 1, n))
    torch.cuda.synchronize()
print("max abs err", int((img.to(torch.int16) - out.to(torch.int16)).abs().max().item()))
