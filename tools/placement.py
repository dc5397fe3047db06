"""Does the decode/encode time depend on where the buffers sit?  Times the C3 shard (64 x 4096^2, L4, Medium)
for several relative placements of input and output inside one big allocation, in one process."""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import rustyhgi_amd as H
from rustyhgi_amd import _ffi
L = _ffi.lib()
ctx = H.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
W = Hh = 4096; NF = 64; n = NF * W * Hh
import numpy as np
lut = np.zeros(256, np.uint8); err = np.zeros(1, np.uint8)
_ffi.check(L.hgi_linear_lut(2, lut.ctypes.data, err.ctypes.data))
pool = torch.empty(3 * n + (256 << 20), dtype=torch.uint8, device="cuda")
base = pool.data_ptr()
print("pool at %#x" % base)
def at(off): return base + off
_ffi.check(L.hgi_synth_u8_dev(ctx.handle, _ffi.SYNTH_RAMP, 0x48474933 + 3, 0, W, Hh, at(0), NF, W * Hh))
def time_it(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for goff in (n, n + 4096, n + (1 << 20), n + (2 << 20) + 12288, n + (37 << 20)):
    for ooff_rel in (n, n + 8192, n + (3 << 20), n + (64 << 20) + 4096):
        ooff = goff + ooff_rel
        if ooff + n > pool.numel(): continue
        enc = lambda: _ffi.check(L.hgi_encode_u8_dev(ctx.handle, at(0), W, Hh, 4, 1, lut.ctypes.data, at(goff), NF, W * Hh))
        dec = lambda: _ffi.check(L.hgi_decode_u8_dev(ctx.handle, at(goff), W, Hh, 4, 1, at(ooff), NF, W * Hh))
        te = time_it(enc); td = time_it(dec)
        print("grid at +%11d  out at grid+%11d : encode %.4f ms  decode %.4f ms" % (goff, ooff_rel, te, td))
