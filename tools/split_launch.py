# the 512-frame decode (and encode) as ONE launch against the same frames in 2 / 4 / 8 / 16 launches, decode-only and in the bench
# pattern (encode, then decode of the grid just written); release library, composed planes -> profiles/r04_split_launch.txt
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustyhgi_amd as H
from rustyhgi_amd import _ffi
L = _ffi.lib()
ctx = H.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
S, F = 4096, int(os.environ.get("FRAMES", "512"))
planes = H.Planes(ctx, F * S * S, 3)
print("report:", planes.report, "| separated", planes.separated)
img, grid, out = (planes.torch(i, (F, S, S)) for i in range(3))
_ffi.check(L.hgi_synth_u8_dev(ctx.handle, _ffi.SYNTH_RAMP, 0x48474930 + 3, 0, S, S, img.data_ptr(), F, S * S))
lut = np.zeros(256, np.uint8); err = ctypes.c_uint8(0)
L.hgi_linear_lut(2, lut.ctypes.data, ctypes.byref(err))
n = S * S
def enc(parts):
    per = F // parts
    for p in range(parts):
        _ffi.check(L.hgi_encode_u8_dev(ctx.handle, img.data_ptr() + p * per * n, S, S, 4, 1, lut.ctypes.data, grid.data_ptr() + p * per * n, per, n))
def dec(parts):
    per = F // parts
    for p in range(parts):
        _ffi.check(L.hgi_decode_u8_dev(ctx.handle, grid.data_ptr() + p * per * n, S, S, 4, 1, out.data_ptr() + p * per * n, per, n))
for _ in range(30): enc(1); dec(1)
torch.cuda.synchronize()
ref = out.clone() if F <= 128 else None
def timed(fn, reps=10):
    ms = ctypes.c_float(0)
    for _ in range(3): fn()
    _ffi.check(L.hgi_timer_start(ctx.handle))
    for _ in range(reps): fn()
    _ffi.check(L.hgi_timer_stop(ctx.handle, ctypes.byref(ms)))
    return ms.value / reps
for rnd in range(2):
    print("round", rnd)
    for parts in (1, 2, 4, 8, 16, 1):
        d = timed(lambda: dec(parts))
        e = timed(lambda: enc(parts))
        pair = timed(lambda: (enc(1), dec(parts)))
        pair2 = timed(lambda: (enc(parts), dec(parts)))
        print("  %2d launches: decode only %.4f ms  encode only %.4f ms | encode(1 launch) + decode(split) %.4f ms | both split %.4f ms" % (parts, d, e, pair, pair2))
