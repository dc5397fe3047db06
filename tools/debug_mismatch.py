import sys, numpy as np
sys.path.insert(0, '.')
from oracle import hgi_oracle as O
import rustyhgi_amd as H
from rustyhgi_amd import _ffi
ctx = H.Context(0)
def run(W, Hh, q, L=4, reps=3):
    img = O.synth(O.SYNTH_RAMP, 0x48474933, 0, W, Hh)
    lut = O.linear_lut(q)[0]
    want = O.encode(img, L, lut)
    prev = None
    for r in range(reps):
        got = np.empty_like(img)
        _ffi.check(_ffi.lib().hgi_encode_u8(ctx.handle, img.ctypes.data, W, Hh, L, 1, lut.ctypes.data, got.ctypes.data))
        bad = np.argwhere(got != want)
        same = prev is not None and bad.shape == prev.shape and (bad == prev).all()
        print("%dx%d q%d L%d rep%d mismatches %d %s" % (W, Hh, q, L, r, len(bad), "(same set)" if same else ""))
        if len(bad) and not same:
            ys, xs = bad[:, 0], bad[:, 1]
            print("  tiles", sorted(set(zip((xs // 128).tolist(), (ys // 64).tolist())))[:8], "x%128", sorted(set((xs % 128).tolist()))[:20], "y%64", sorted(set((ys % 64).tolist()))[:20])
        prev = bad
    dec = np.empty_like(img)
    _ffi.check(_ffi.lib().hgi_decode_u8(ctx.handle, want.ctypes.data, W, Hh, L, 1, dec.ctypes.data))
    print("  decode mismatches", int((dec != O.decode(want, L)).sum()))
for a in ((4096, 4096, 2), (4096, 4096, 1), (4096, 4096, 3), (8192, 2048, 2), (2048, 8192, 2)):
    run(*a)
