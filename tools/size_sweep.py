"""Encode/decode latency over frame sizes and batch counts (Medium, Crossed), for choosing the tile geometry.
Library selectable with HGI_LIB_PATH.  usage: size_sweep.py [levels] [BxWxH ...]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustyhgi_amd as H
from rustyhgi_amd import _ffi
L = _ffi.lib()
ctx = H.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
lut = np.zeros(256, np.uint8); err = np.zeros(1, np.uint8)
_ffi.check(L.hgi_linear_lut(2, lut.ctypes.data, err.ctypes.data))
levels = int(sys.argv[1]) if len(sys.argv) > 1 else 4
def timed(fn, reps=30):
    for _ in range(5): fn()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
custom = [tuple(int(v) for v in a.split("x")) for a in sys.argv[2:]]
for (B, W, Hh) in custom or [(1, 256, 256), (1, 512, 512), (1, 1024, 1024), (1, 1920, 1080), (1, 2048, 2048), (1, 2560, 1440), (1, 3072, 3072),
                   (1, 3840, 2160), (1, 4096, 4096), (2, 4096, 4096), (4, 4096, 4096), (8, 4096, 4096), (16, 1920, 1080)]:
    n = B * W * Hh
    img = torch.empty(n, dtype=torch.uint8, device="cuda"); grid = torch.empty_like(img); out = torch.empty_like(img)
    _ffi.check(L.hgi_synth_u8_dev(ctx.handle, _ffi.SYNTH_RAMP, 0x48474933 + 3, 0, W, Hh, img.data_ptr(), B, W * Hh))
    te = timed(lambda: _ffi.check(L.hgi_encode_u8_dev(ctx.handle, img.data_ptr(), W, Hh, levels, 1, lut.ctypes.data, grid.data_ptr(), B, W * Hh)))
    td = timed(lambda: _ffi.check(L.hgi_decode_u8_dev(ctx.handle, grid.data_ptr(), W, Hh, levels, 1, out.data_ptr(), B, W * Hh)))
    t64 = B * ((W + 127) // 128) * ((Hh + 63) // 64)
    print("%2d x %4dx%4d L%d  tiles64 %6d : encode %7.1f us  decode %7.1f us" % (B, W, Hh, levels, t64, te, td))
