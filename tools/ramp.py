"""Does the step time drift after the GPU starts working?  400 bench steps (encode then decode, 64 x 4096^2, L4 Medium) from
an idle device, every launch timed; prints the series in groups of 10 steps."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustyhgi_amd as H
from rustyhgi_amd import _ffi
L = _ffi.lib()
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = H.Context(0); ctx.set_stream(stream.cuda_stream)
W = Hh = 4096; NF = 64; n = NF * W * Hh
lut = np.zeros(256, np.uint8); err = np.zeros(1, np.uint8)
_ffi.check(L.hgi_linear_lut(2, lut.ctypes.data, err.ctypes.data))
planes = H.Planes(ctx, n, 3)
img, grid, out = (planes.torch(i, (n,)) for i in range(3))
_ffi.check(L.hgi_synth_u8_dev(ctx.handle, _ffi.SYNTH_RAMP, 0x48474930 + 3, 0, W, Hh, img.data_ptr(), NF, W * Hh))
torch.cuda.synchronize()
print("planes separated:", planes.separated)
for idle in (2.0, 0.0):
    time.sleep(idle)
    N = 400
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(N)]
    for k in range(N):
        e = ev[k]
        e[0].record()
        _ffi.check(L.hgi_encode_u8_dev(ctx.handle, img.data_ptr(), W, Hh, 4, 1, lut.ctypes.data, grid.data_ptr(), NF, W * Hh))
        e[1].record()
        _ffi.check(L.hgi_decode_u8_dev(ctx.handle, grid.data_ptr(), W, Hh, 4, 1, out.data_ptr(), NF, W * Hh))
        e[2].record()
    torch.cuda.synchronize()
    te = [e[0].elapsed_time(e[1]) for e in ev]; td = [e[1].elapsed_time(e[2]) for e in ev]
    print("after %.0f s idle: step groups of 10 (mean encode / decode ms)" % idle)
    for g in range(0, N, 10):
        print("  steps %3d-%3d  (t = %5.1f ms)  enc %.4f  dec %.4f" % (g, g + 9, sum(te[:g]) + sum(td[:g]), np.mean(te[g:g + 10]), np.mean(td[g:g + 10])))
del img, grid, out
planes.close()
