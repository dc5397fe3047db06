# per-GiB-offset time of the real encode / decode (64 frames of 4096^2 each) on the composed planes of the literal C3, next to what
# hgi_planes_alloc found (run on the knobs build with HGI_PLANES_TRACE=1 to see the class of every chunk) -> profiles/r04_per_offset.txt
import os, sys, re, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustyhgi_amd as H
from rustyhgi_amd import _ffi
L = _ffi.lib()
ctx = H.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
S, F = 4096, 512
planes = H.Planes(ctx, F * S * S, 3)
print("report:", planes.report, "| separated", planes.separated)
img, grid, out = (planes.torch(i, (F, S, S)) for i in range(3))
_ffi.check(L.hgi_synth_u8_dev(ctx.handle, _ffi.SYNTH_RAMP, 0x48474930 + 3, 0, S, S, img.data_ptr(), F, S * S))
lut = np.zeros(256, np.uint8); err = ctypes.c_uint8(0)
L.hgi_linear_lut(2, lut.ctypes.data, ctypes.byref(err))
def enc(a, b, n): _ffi.check(L.hgi_encode_u8_dev(ctx.handle, a, S, S, 4, 1, lut.ctypes.data, b, n, S * S))
def dec(a, b, n): _ffi.check(L.hgi_decode_u8_dev(ctx.handle, a, S, S, 4, 1, b, n, S * S))
for _ in range(30): enc(img.data_ptr(), grid.data_ptr(), F); dec(grid.data_ptr(), out.data_ptr(), F)
torch.cuda.synchronize()
def timed(fn, reps=12):
    ms = ctypes.c_float(0)
    for _ in range(4): fn()
    _ffi.check(L.hgi_timer_start(ctx.handle))
    for _ in range(reps): fn()
    _ffi.check(L.hgi_timer_stop(ctx.handle, ctypes.byref(ms)))
    return ms.value / reps
print("whole batch: encode %.4f ms decode %.4f ms" % (timed(lambda: enc(img.data_ptr(), grid.data_ptr(), F), 8), timed(lambda: dec(grid.data_ptr(), out.data_ptr(), F), 8)))
GiB = 1 << 30
for rnd in range(2):
    e = [timed(lambda m=m: enc(img.data_ptr() + m * GiB, grid.data_ptr() + m * GiB, 64)) for m in range(8)]
    d = [timed(lambda m=m: dec(grid.data_ptr() + m * GiB, out.data_ptr() + m * GiB, 64)) for m in range(8)]
    print("per offset, 64 frames: encode", " ".join("%.4f" % v for v in e))
    print("per offset, 64 frames: decode", " ".join("%.4f" % v for v in d))
# decode of grid chunk m into out chunk m' for all pairs (does the pairing matter, or the grid chunk alone?)
t = np.zeros((8, 8))
for m in range(8):
    for k in range(8):
        t[m, k] = timed(lambda: dec(grid.data_ptr() + m * GiB, out.data_ptr() + k * GiB, 64), 6)
print("decode grid chunk (row) -> out chunk (column), ms:")
for m in range(8): print("  ", " ".join("%.4f" % v for v in t[m]))
