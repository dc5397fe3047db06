"""Device histogram on the C3 shard's grids: time, read rate, entropy estimate per quantization level."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustyhgi_amd as H
from rustyhgi_amd import _ffi, entropy
from rustyhgi_amd.quantizator import Linear, QuantizationLevel
L = _ffi.lib()
ctx = H.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
W = Hh = 4096; NF = 64
img = torch.empty((NF, Hh, W), dtype=torch.uint8, device="cuda")
_ffi.check(L.hgi_synth_u8_dev(ctx.handle, _ffi.SYNTH_RAMP, 0x48474933 + 3, 0, W, Hh, img.data_ptr(), NF, W * Hh))
def timed(fn, reps=10):
    for _ in range(2): fn()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for lvl in (QuantizationLevel.Lossless, QuantizationLevel.Low, QuantizationLevel.Medium, QuantizationLevel.High):
    enc = H.Encoder(H.interpolator.Crossed(), Linear.from_level(lvl), 4, context=ctx)
    grid = enc.encode_batch(img)
    t = timed(lambda: entropy.histogram(grid, context=ctx))
    hist = entropy.histogram(grid, context=ctx)
    bpp = entropy.entropy_bits_per_pixel(hist)
    print("%-8s histogram of 64 x 4096^2 grids: %.3f ms = %.0f GB/s read; order-0 entropy %.2f bits/px (ratio %.1fx)" % (
        lvl.name, t, NF * W * Hh / t * 1e-6, bpp.mean(), 8.0 / bpp.mean()))
noise = torch.randint(0, 256, (NF, Hh, W), dtype=torch.uint8, device="cuda")
t = timed(lambda: entropy.histogram(noise, context=ctx))
print("uniform noise: %.3f ms = %.0f GB/s read" % (t, NF * W * Hh / t * 1e-6))
