"""Host-memory batches: frame-by-frame hgi_encode_u8 against the pipelined hgi_encode_u8_batch (PCIe-inclusive)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustyhgi_amd as H
from rustyhgi_amd import _ffi
L = _ffi.lib()
ctx = H.Context(0)
lut = np.zeros(256, np.uint8); err = np.zeros(1, np.uint8)
_ffi.check(L.hgi_linear_lut(2, lut.ctypes.data, err.ctypes.data))
def best(fn, reps=5):
    fn(); t = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
    return min(t)
for (B, W, Hh) in [(32, 1920, 1080), (16, 4096, 4096), (64, 4096, 4096), (256, 512, 512)]:
    n = W * Hh
    src = np.random.default_rng(1).integers(0, 256, (B, Hh, W), dtype=np.uint8)
    dst = np.empty_like(src); back = np.empty_like(src)
    def one_by_one():
        for f in range(B):
            _ffi.check(L.hgi_encode_u8(ctx.handle, src[f].ctypes.data, W, Hh, 4, 1, lut.ctypes.data, dst[f].ctypes.data))
    def batched():
        _ffi.check(L.hgi_encode_u8_batch(ctx.handle, src.ctypes.data, W, Hh, 4, 1, lut.ctypes.data, dst.ctypes.data, B, n))
    def batched_dec():
        _ffi.check(L.hgi_decode_u8_batch(ctx.handle, dst.ctypes.data, W, Hh, 4, 1, back.ctypes.data, B, n))
    t1, t2, t3 = best(one_by_one), best(batched), best(batched_dec)
    gb = B * n / 1e9
    print("%3d x %4dx%4d host frames: encode one by one %8.2f ms (%5.1f GB/s in)  batch call %8.2f ms (%5.1f GB/s in, %.2fx)  decode batch %8.2f ms" % (
        B, W, Hh, t1 * 1e3, gb / t1, t2 * 1e3, gb / t2, t1 / t2, t3 * 1e3))

# one large frame at a time: the banded host path (on the KNOBS build -- HGI_LIB_PATH=rustyhgi_amd/libhgi_hip_knobs.so -- HGI_NO_BANDS=1 turns it off)
for (W, Hh, LV) in [(1920, 1080, 4), (4096, 4096, 4), (8192, 8192, 4), (16384, 16384, 4), (16384, 16384, 8)]:
    src = np.random.default_rng(2).integers(0, 256, (Hh, W), dtype=np.uint8); dst = np.empty_like(src); back = np.empty_like(src)
    te = best(lambda: _ffi.check(L.hgi_encode_u8(ctx.handle, src.ctypes.data, W, Hh, LV, 1, lut.ctypes.data, dst.ctypes.data)))
    td = best(lambda: _ffi.check(L.hgi_decode_u8(ctx.handle, dst.ctypes.data, W, Hh, LV, 1, back.ctypes.data)))
    print("single host frame %5dx%5d L%d: hgi_encode_u8 %8.3f ms (%5.1f GB/s in)  hgi_decode_u8 %8.3f ms   bands %s" % (
        W, Hh, LV, te * 1e3, W * Hh / te / 1e9, td * 1e3, "off" if os.environ.get("HGI_NO_BANDS") else "on"))
