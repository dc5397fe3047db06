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
for name, img in cases:
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
