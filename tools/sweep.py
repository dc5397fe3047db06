"""Encode/decode time of the batch shape (frames x 4096^2) over levels, quantizer and interpolator: looks for slow corners.
usage: sweep.py [frames]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustyhgi_amd as H
from rustyhgi_amd import _ffi
L = _ffi.lib()
ctx = H.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
W = Hh = 4096; NF = int(sys.argv[1]) if len(sys.argv) > 1 else 64; n = NF * W * Hh
img = torch.empty(n, dtype=torch.uint8, device="cuda"); grid = torch.empty_like(img); out = torch.empty_like(img)
_ffi.check(L.hgi_synth_u8_dev(ctx.handle, _ffi.SYNTH_RAMP, 0x48474933 + 3, 0, W, Hh, img.data_ptr(), NF, W * Hh))
def timed(fn, reps=10):
    for _ in range(2): fn()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
print("| levels | quantizer | interp | encode ms | decode ms | enc GB/s | dec GB/s | max err |")
print("|---|---|---|---|---|---|---|---|")
for q in (0, 2):
    lut = np.zeros(256, np.uint8); err = np.zeros(1, np.uint8)
    _ffi.check(L.hgi_linear_lut(q, lut.ctypes.data, err.ctypes.data))
    for interp in (1, 0):
        for lv in (1, 2, 3, 4, 5, 6, 7, 8, 12):
            te = timed(lambda: _ffi.check(L.hgi_encode_u8_dev(ctx.handle, img.data_ptr(), W, Hh, lv, interp, lut.ctypes.data, grid.data_ptr(), NF, W * Hh)))
            td = timed(lambda: _ffi.check(L.hgi_decode_u8_dev(ctx.handle, grid.data_ptr(), W, Hh, lv, interp, out.data_ptr(), NF, W * Hh)))
            mx = int((img.view(-1)[::7].to(torch.int16) - out.view(-1)[::7].to(torch.int16)).abs().max().item())
            print("| %d | %s | %s | %.4f | %.4f | %.0f | %.0f | %d |" % (lv, ("lossless", "low", "medium", "high")[q], ("lefttop", "crossed")[interp],
                  te, td, 2 * n / te * 1e-6, 2 * n / td * 1e-6, mx))
