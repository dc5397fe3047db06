"""Why is a 16384^2 decode slower right after an encode?  Patterns of launches, each timed by events."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustyhgi_amd as H
from rustyhgi_amd import _ffi
L = _ffi.lib()
ctx = H.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
W = Hh = 16384; n = W * Hh; LEVELS = int(os.environ.get("C4_LEVELS", "8"))
lut = np.zeros(256, np.uint8); err = np.zeros(1, np.uint8)
_ffi.check(L.hgi_linear_lut(3, lut.ctypes.data, err.ctypes.data))
planes = H.Planes(ctx, n, 5)
img, grid, out, grid2, out2 = (planes.torch(i, (Hh, W)) for i in range(5))
_ffi.check(L.hgi_synth_u8_dev(ctx.handle, _ffi.SYNTH_RAMP, 0x48474933 + 4, 0, W, Hh, img.data_ptr(), 1, n))
def enc(dst=grid): _ffi.check(L.hgi_encode_u8_dev(ctx.handle, img.data_ptr(), W, Hh, LEVELS, 1, lut.ctypes.data, dst.data_ptr(), 1, n))
def dec(src=grid, dst=out): _ffi.check(L.hgi_decode_u8_dev(ctx.handle, src.data_ptr(), W, Hh, LEVELS, 1, dst.data_ptr(), 1, n))
def cp(): _ffi.check(L.hgi_copy_u8_dev(ctx.handle, img.data_ptr(), out2.data_ptr(), n))
enc(grid2)
ops = {"E": enc, "D": dec, "d": lambda: dec(grid2, out), "C": cp}
for pattern in ("E", "D", "ED", "EDD", "EED", "Ed", "Edd", "CD", "CDD", "DC", "EC"):
    for _ in range(30):
        for c in pattern: ops[c]()
    reps = 30
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(len(pattern) + 1)] for _ in range(reps)]
    for ev in evs:
        ev[0].record()
        for i, c in enumerate(pattern):
            ops[c](); ev[i + 1].record()
    torch.cuda.synchronize()
    t = [np.median([ev[i].elapsed_time(ev[i + 1]) for ev in evs]) * 1e3 for i in range(len(pattern))]
    print("%-4s " % pattern + "  ".join("%s %.1f" % (c, x) for c, x in zip(pattern, t)))
