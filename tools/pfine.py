"""Level-wise path on the C3 shard (64 x 4096^2, L4, Medium): one launch per level, so a kernel trace shows the finest
pass (P_fine, 75 % of the pixels, 1.75 B/px algorithmic) on its own.  Run under rocprofv3 --kernel-trace --stats."""
import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import rustyhgi_amd as H
from rustyhgi_amd import _ffi
L = _ffi.lib()
ctx = H.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream); ctx.set_path(_ffi.PATH_LEVELWISE)
W = Hh = 4096; NF = 64; n = NF * W * Hh
lut = np.zeros(256, np.uint8); err = np.zeros(1, np.uint8)
_ffi.check(L.hgi_linear_lut(2, lut.ctypes.data, err.ctypes.data))
img = torch.empty(n, dtype=torch.uint8, device="cuda"); grid = torch.empty_like(img); out = torch.empty_like(img)
_ffi.check(L.hgi_synth_u8_dev(ctx.handle, _ffi.SYNTH_RAMP, 0x48474933 + 3, 0, W, Hh, img.data_ptr(), NF, W * Hh))
# P_fine with the product kernels: levels = 1 IS the finest pass (lattice = even/even pixels, 25 %, copied through)
fused = H.Context(0); fused.set_stream(torch.cuda.current_stream().cuda_stream)
def timed(fn, reps=20):
    for _ in range(3): fn()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
te = timed(lambda: _ffi.check(L.hgi_encode_u8_dev(fused.handle, img.data_ptr(), W, Hh, 1, 1, lut.ctypes.data, grid.data_ptr(), NF, W * Hh)))
td = timed(lambda: _ffi.check(L.hgi_decode_u8_dev(fused.handle, grid.data_ptr(), W, Hh, 1, 1, out.data_ptr(), NF, W * Hh)))
alg = 1.75 * n
print("P_fine alone (fused kernels, levels=1, same-direction launches back to back): encode %.4f ms = %.0f GB/s, decode %.4f ms = %.0f GB/s "
      "(algorithmic 1.75 B/px; the launch also copies the 25 %% lattice through, 2 B/px moved)" % (te, alg / te * 1e-6, td, alg / td * 1e-6))
for _ in range(6):
    _ffi.check(L.hgi_encode_u8_dev(ctx.handle, img.data_ptr(), W, Hh, 4, 1, lut.ctypes.data, grid.data_ptr(), NF, W * Hh))
    _ffi.check(L.hgi_decode_u8_dev(ctx.handle, grid.data_ptr(), W, Hh, 4, 1, out.data_ptr(), NF, W * Hh))
torch.cuda.synchronize()
print("max abs err", int((img.to(torch.int16) - out.to(torch.int16)).abs().max().item()))
