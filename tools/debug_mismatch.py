"""Debug helper: encode/decode a few shapes through the C ABI (library selectable with HGI_LIB_PATH) and
describe where the result differs from the oracle (which tiles, which rows/columns inside the tile)."""
import sys, numpy as np
sys.path.insert(0, '.')
from oracle import hgi_oracle as O
import rustyhgi_amd as H
from rustyhgi_amd import _ffi
ctx = H.Context(0)
TH = int(sys.argv[1]) if len(sys.argv) > 1 else 64
def describe(name, got, want):
    bad = np.argwhere(got != want)
    print("  %s mismatches %d" % (name, len(bad)))
    if len(bad):
        ys, xs = bad[:, 0], bad[:, 1]
        tiles = sorted(set(zip((xs // 128).tolist(), (ys // TH).tolist())))
        print("    tiles(%d): %s ..." % (len(tiles), tiles[:10]), "x%128:", sorted(set((xs % 128).tolist()))[:24], "y%%%d:" % TH, sorted(set((ys % TH).tolist()))[:32])
        for y, x in bad[:4]:
            print("    (x=%d,y=%d) got %d want %d" % (x, y, got[y, x], want[y, x]))
def run(W, Hh, q, L):
    img = O.synth(O.SYNTH_NOISE, 0x48474933, 0, W, Hh)
    lut = O.linear_lut(q)[0]
    want = O.encode(img, L, lut)
    got = np.empty_like(img)
    _ffi.check(_ffi.lib().hgi_encode_u8(ctx.handle, img.ctypes.data, W, Hh, L, 1, lut.ctypes.data, got.ctypes.data))
    print("%dx%d q%d L%d" % (W, Hh, q, L))
    describe("encode", got, want)
    dec = np.empty_like(img)
    _ffi.check(_ffi.lib().hgi_decode_u8(ctx.handle, want.ctypes.data, W, Hh, L, 1, dec.ctypes.data))
    describe("decode", dec, O.decode(want, L))
for a in ((4096, 4096, 2, 4), (4096, 4096, 0, 6), (2048, 2048, 2, 4), (4096, 512, 2, 4), (512, 4096, 2, 4), (1024, 1024, 0, 6), (4096, 4096, 0, 5)):
    run(*a)
