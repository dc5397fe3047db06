"""Placement classes, part 3: which address bit is it?  One pool; dst = src + d for d = 8 .. 64 GiB, and a moved src.
usage: modes3.py [pool GiB]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustyhgi_amd as H
from rustyhgi_amd import _ffi
L = _ffi.lib()
GiB = 1 << 30; MiB = 1 << 20
POOL = int(sys.argv[1]) if len(sys.argv) > 1 else 72
NF = 64; W = Hh = 4096; n = NF * W * Hh
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = H.Context(0); ctx.set_stream(stream.cuda_stream)
lut = np.zeros(256, np.uint8); err = np.zeros(1, np.uint8)
_ffi.check(L.hgi_linear_lut(2, lut.ctypes.data, err.ctypes.data))
pool = torch.empty(POOL * GiB, dtype=torch.uint8, device="cuda")
base = pool.data_ptr()
print("pool %d GiB at %#x" % (POOL, base))
def enc(a, b): _ffi.check(L.hgi_encode_u8_dev(ctx.handle, a, W, Hh, 4, 1, lut.ctypes.data, b, NF, W * Hh))
def dec(a, b): _ffi.check(L.hgi_decode_u8_dev(ctx.handle, a, W, Hh, 4, 1, b, NF, W * Hh))
def cp(a, b): _ffi.check(L.hgi_copy_u8_dev(ctx.handle, a, b, n))
def timed(fn, reps=8):
    fn(); fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for s in (0, 5, 8):
    _ffi.check(L.hgi_synth_u8_dev(ctx.handle, _ffi.SYNTH_RAMP, 0x48474933 + 3, 0, W, Hh, base + s * GiB, NF, W * Hh))
    for d in list(range(2, 36)) + [40, 47, 48, 49, 56, 63, 64, 65]:
        if (s + d + 1) > POOL: continue
        a, b = base + s * GiB, base + (s + d) * GiB
        te = timed(lambda: enc(a, b)); td = timed(lambda: dec(a, b)); tc = timed(lambda: cp(a, b))
        print("src +%2d GiB  dst = src + %2d GiB : encode %.4f  decode %.4f  copy %.4f  %s" % (s, d, te, td, tc, "FAST" if td < 0.372 else ""))
