"""C4 (16384^2, level 8, High, Crossed) for rocprofv3: warm-up, then REPS bench-pattern steps (encode -> decode) on
placed planes.  tools/c4_profile.sh runs it under --kernel-trace --stats and under --pmc FETCH_SIZE / WRITE_SIZE;
tools/summarize_c4.py turns the CSVs into profiles/r03_c4_summary.md.  Also prints its own hipEvent timing."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustyhgi_amd as H
from rustyhgi_amd import _ffi
L = _ffi.lib()
ctx = H.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
W = Hh = 16384; n = W * Hh; LEVELS = 8
REPS = int(os.environ.get("C4_REPS", "60"))
lut = np.zeros(256, np.uint8); err = np.zeros(1, np.uint8)
_ffi.check(L.hgi_linear_lut(3, lut.ctypes.data, err.ctypes.data))
planes = H.Planes(ctx, n, 3)
img, grid, out = (planes.torch(i, (Hh, W)) for i in range(3))
_ffi.check(L.hgi_synth_u8_dev(ctx.handle, _ffi.SYNTH_RAMP, 0x48474933 + 4, 0, W, Hh, img.data_ptr(), 1, n))
def enc(): _ffi.check(L.hgi_encode_u8_dev(ctx.handle, img.data_ptr(), W, Hh, LEVELS, 1, lut.ctypes.data, grid.data_ptr(), 1, n))
def dec(): _ffi.check(L.hgi_decode_u8_dev(ctx.handle, grid.data_ptr(), W, Hh, LEVELS, 1, out.data_ptr(), 1, n))
def cp(): _ffi.check(L.hgi_copy_u8_dev(ctx.handle, img.data_ptr(), out.data_ptr(), n))
for _ in range(int(os.environ.get("C4_WARM", "300"))): enc(); dec()      # the encoder's launch time follows the clock for tens of milliseconds
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(REPS)]
for e in ev:
    e[0].record(); enc(); e[1].record(); dec(); e[2].record()
torch.cuda.synchronize()
te = np.median([e[0].elapsed_time(e[1]) for e in ev]) * 1e3; td = np.median([e[1].elapsed_time(e[2]) for e in ev]) * 1e3
for _ in range(10): cp()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(REPS): cp()
b.record(); torch.cuda.synchronize()
tc = a.elapsed_time(b) / REPS * 1e3
dec()
torch.cuda.synchronize()
print("C4 hipEvents: separated=%s encode %.1f us (%.3f of 8 TB/s)  decode %.1f us (%.3f)  copy %.1f us (%.3f)  max abs err %d" % (
    planes.separated, te, 2 * n / te / 8e6, td, 2 * n / td / 8e6, tc, 2 * n / tc / 8e6,
    int((img[:2048].to(torch.int16) - out[:2048].to(torch.int16)).abs().max().item())))
