"""The archive's entropy stage, device against host: size and time of the DEFLATE stream of one grid written by
hgi_deflate_grid_dev (Huffman-coded literals, on the GPU) and by zlib at level 9 (what the reference's
Compression::best() does, one CPU thread), on real and synthetic residual grids."""
import os, sys, time, zlib, struct, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustyhgi_amd as H
from rustyhgi_amd import entropy
from oracle import hgi_oracle as O
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lena = np.fromfile(os.path.join(ROOT, "tests/golden/lena_256.u8"), np.uint8).reshape(256, 256)
fullhd = np.array(Image.open(os.path.join(ROOT, "tests/golden/fullhd_luma.png")))
cases = [("LENA 256x256", lena), ("fullhd luma 1920x1080", fullhd), ("ramp(3) 4096x4096", O.synth(O.SYNTH_RAMP, 0x48474933, 0, 4096, 4096)),
         ("xy 1920x1080 (criterion image)", O.synth(O.SYNTH_XY, 0, 0, 1920, 1080))]
print("| grid | level | pixels | zlib-9 bytes | device bytes | device / zlib-9 | zlib-9 ms (1 CPU thread) | device ms | speed-up |")
print("|---|---|---|---|---|---|---|---|---|")
for name, img in ([] if os.environ.get("BATCH_ONLY") else cases):
    for q, qn in ((0, "Lossless"), (1, "Low"), (2, "Medium"), (3, "High")):
        grid = O.encode(img, 4, O.linear_lut(q)[0])
        h, w = grid.shape
        body = struct.pack("<Q", w * h) + grid.tobytes() + struct.pack("<Q", w)
        t0 = time.perf_counter()
        co = zlib.compressobj(9, zlib.DEFLATED, -15)
        z = co.compress(body) + co.flush()
        tz = time.perf_counter() - t0
        d = torch.from_numpy(grid).cuda()
        for _ in range(3):
            s = entropy.deflate_grid(d)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            s = entropy.deflate_grid(d)
        td = (time.perf_counter() - t0) / reps
        assert zlib.decompressobj(-15).decompress(s) == body
        print("| %s | %s | %d | %d | %d | %.3f | %.2f | %.3f | %.0fx |" % (name, qn, w * h, len(z), len(s), len(s) / len(z), tz * 1e3, td * 1e3, tz / td))

# a batch: the C3 shard's 64 grids of 4096 x 4096 at Medium, one call against 64 single-frame calls
import ctypes
from rustyhgi_amd import _ffi
F, S = 64, 4096
ctx = H.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
imgs = torch.empty((F, S, S), dtype=torch.uint8, device="cuda"); grids = torch.empty_like(imgs)
lut = np.ascontiguousarray(O.linear_lut(2)[0])
_ffi.check(_ffi.lib().hgi_synth_u8_dev(ctx.handle, _ffi.SYNTH_RAMP, 0x48474930 + 3, 0, S, S, imgs.data_ptr(), F, S * S))
_ffi.check(_ffi.lib().hgi_encode_u8_dev(ctx.handle, imgs.data_ptr(), S, S, 4, 1, lut.ctypes.data, grids.data_ptr(), F, S * S))
torch.cuda.synchronize()
# through the C ABI with caller-owned, already touched host buffers (a fresh 1.2 GB numpy array would be timed page faults)
cap = S * S // 2
out = np.zeros((F, cap), np.uint8)
sizes = (ctypes.c_size_t * F)()
L = _ffi.lib()
def batch_call():
    _ffi.check(L.hgi_deflate_grids_dev(ctx.handle, grids.data_ptr(), S, S, F, S * S, out.ctypes.data, cap, sizes))
def single_calls():
    one = ctypes.c_size_t(0)
    for f in range(F):
        _ffi.check(L.hgi_deflate_grid_dev(ctx.handle, grids[f].data_ptr(), S, S, out[f].ctypes.data, cap, ctypes.byref(one)))
batch_call(); single_calls()
def best_of(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    if os.environ.get("BATCH_ONLY"): print("in order, ms:", " ".join("%.1f" % (t * 1e3) for t in ts))
    return sorted(ts)
tbs = best_of(batch_call, 7)
tb = tbs[len(tbs) // 2]
streams = [out[f, :sizes[f]].tobytes() for f in range(F)]
ts = best_of(single_calls, 3)[1]
assert streams[5] == entropy.deflate_grid(grids[5], context=ctx)
total = sum(len(s) for s in streams)
# the same call with the caller's buffer in pinned memory: the downloads no longer pass through the runtime's staging copies
pinned = torch.empty((F, cap), dtype=torch.uint8, pin_memory=True)
def batch_pinned():
    _ffi.check(L.hgi_deflate_grids_dev(ctx.handle, grids.data_ptr(), S, S, F, S * S, pinned.data_ptr(), cap, sizes))
batch_pinned()
tps = best_of(batch_pinned, 7)
assert bytes(pinned[5, :sizes[5]].numpy()) == streams[5]
print()
print("batch of %d grids %dx%d Medium (the C3 shard): %d -> %d bytes (%.2fx); hgi_deflate_grids_dev %.1f ms = %.2f ms per frame, %.1f GB/s of grid;"
      " %d single-frame calls %.1f ms (medians; batch call min %.1f max %.1f ms over 7)" % (F, S, S, F * S * S, total, F * S * S / total, tb * 1e3, tb * 1e3 / F, F * S * S / tb / 1e9, F, ts * 1e3, tbs[0] * 1e3, tbs[-1] * 1e3))
print("the batch call into a pinned host buffer: %.1f ms (median of 7) = %.1f GB/s of grid" % (tps[3] * 1e3, F * S * S / tps[3] / 1e9))
