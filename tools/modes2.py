"""Placement classes, part 2: inside ONE large allocation (physically as contiguous as the driver makes it), decode
grid -> out for many distances between the two, and encode img -> grid likewise.  If the fast / slow class is a function
of physical address bits, it shows as a pattern in the distance.  usage: modes2.py [pool GiB]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustyhgi_amd as H
from rustyhgi_amd import _ffi
L = _ffi.lib()
GiB = 1 << 30; MiB = 1 << 20
POOL = int(sys.argv[1]) if len(sys.argv) > 1 else 20
NF = 64; W = Hh = 4096; n = NF * W * Hh
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = H.Context(0); ctx.set_stream(stream.cuda_stream)
lut = np.zeros(256, np.uint8); err = np.zeros(1, np.uint8)
_ffi.check(L.hgi_linear_lut(2, lut.ctypes.data, err.ctypes.data))
pool = torch.empty(POOL * GiB, dtype=torch.uint8, device="cuda")
base = pool.data_ptr()
print("pool %d GiB at %#x" % (POOL, base))
_ffi.check(L.hgi_synth_u8_dev(ctx.handle, _ffi.SYNTH_RAMP, 0x48474933 + 3, 0, W, Hh, base, NF, W * Hh))
def enc(a, b): _ffi.check(L.hgi_encode_u8_dev(ctx.handle, a, W, Hh, 4, 1, lut.ctypes.data, b, NF, W * Hh))
def dec(a, b): _ffi.check(L.hgi_decode_u8_dev(ctx.handle, a, W, Hh, 4, 1, b, NF, W * Hh))
def timed(fn, reps=10):
    fn(); fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
dists = [GiB + m * 2 * MiB for m in range(0, 12)] + [GiB + m * MiB for m in (1, 3, 64, 65, 128, 256, 512)] + \
        [k * GiB for k in (2, 3, 4, 5, 6, 7, 8, 12, 16)] + [k * GiB + 2 * MiB for k in (2, 3, 4, 8)]
for d in dists:
    if d + n > POOL * GiB: continue
    te = timed(lambda: enc(base, base + d))
    td = timed(lambda: dec(base, base + d))
    print("dst = src + %6d MiB : encode %.4f  decode %.4f" % (d // MiB, te, td))
# the same with the source moved instead
for s in (2 * MiB, 4 * MiB, 1 * GiB, 1 * GiB + 2 * MiB, 3 * GiB):
    d = 8 * GiB
    te = timed(lambda: enc(base + s, base + d)); td = timed(lambda: dec(base + s, base + d))
    print("src at +%6d MiB, dst at +%d MiB : encode %.4f  decode %.4f" % (s // MiB, d // MiB, te, td))
