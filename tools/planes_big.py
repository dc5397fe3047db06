"""hgi_planes_alloc at the sizes of the strong-scaling bench (512 / 256 / 128 / 64 frames of 4096^2 per GPU = planes of 8 / 4 /
2 / 1 GiB): seconds per call, whether the separation was established, the codec's step on the planes against the same step
on plain torch allocations in the same process, and that torch can view, launch on and download from composed planes.
    python tools/planes_big.py [frames ...]          -> profiles/r04_planes_big.txt"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustyhgi_amd as H
from rustyhgi_amd import _ffi
from rustyhgi_amd.interpolator import Crossed
from rustyhgi_amd.quantizator import Linear, QuantizationLevel

L = _ffi.lib()
S = 4096
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = H.Context(0)
ctx.set_stream(stream.cuda_stream)
enc = H.Encoder(Crossed(), Linear.from_level(QuantizationLevel.Medium), 4, context=ctx)
dec = H.Decoder(Crossed(), context=ctx)


def step_ms(img, grid, out, reps=10, warm=30):
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(reps)]
    for i in range(warm + reps):
        e = ev[max(i - warm, 0)]
        e[0].record()
        enc.encode_batch(img, out=grid)
        e[1].record()
        dec.decode_batch(grid, 4, out=out)
        e[2].record()
    torch.cuda.synchronize()
    return (float(np.mean([e[0].elapsed_time(e[1]) for e in ev])), float(np.mean([e[1].elapsed_time(e[2]) for e in ev])))


for F in [int(a) for a in sys.argv[1:]] or [512, 256, 128, 64]:
    n = F * S * S
    free0 = torch.cuda.mem_get_info()[0]
    t0 = time.perf_counter()
    planes = H.Planes(ctx, n, 3)
    dt = time.perf_counter() - t0
    img, grid, out = (planes.torch(i, (F, S, S)) for i in range(3))
    _ffi.check(L.hgi_synth_u8_dev(ctx.handle, _ffi.SYNTH_RAMP, 0x48474933, 0, S, S, img.data_ptr(), F, S * S))
    e, d = step_ms(img, grid, out)
    # the views behave like any tensor: a torch kernel on them, a download, equality with a plain copy
    plain = [torch.empty((F, S, S), dtype=torch.uint8, device="cuda") for _ in range(3)]
    plain[0].copy_(img)
    pe, pd = step_ms(*plain)
    same = bool(torch.equal(plain[1], grid)) and bool(torch.equal(plain[2], out))
    host = out[F - 1].cpu().numpy()
    err = int(np.abs(host.astype(np.int16) - img[F - 1].cpu().numpy().astype(np.int16)).max())
    span = min(n, 2 << 30)
    probes = []
    for _ in range(6):
        planes.probe_ms(0, 1)
    for a, b in ((0, 1), (1, 2), (0, 2)):
        probes.append(planes.probe_ms(a, b) / (span / (1 << 30)))
    print("%3d frames (planes of %.0f GiB): hgi_planes_alloc %.3f s  separated %s | per 64 frames: placed encode %.4f decode %.4f ms, "
          "plain torch %.4f / %.4f ms | same bytes %s, max err %d | probe ms per GiB 0->1 %.4f 1->2 %.4f 0->2 %.4f | free before %.1f GiB"
          % (F, n / 2**30, dt, planes.separated, e * 64 / F, d * 64 / F, pe * 64 / F, pd * 64 / F, same, err, probes[0], probes[1], probes[2],
             free0 / 2**30), flush=True)
    if n > (1 << 30):      # composed planes: every GiB offset of both neighbouring pairs (what "separated" promises)
        import ctypes
        GiB = 1 << 30
        for a, b in ((0, 1), (1, 2)):
            row = []
            for m in range(n // GiB):
                ms = ctypes.c_float(0)
                _ffi.check(L.hgi_probe_pair_u8_dev(ctx.handle, planes.pointers[a] + m * GiB, planes.pointers[b] + m * GiB, GiB, ctypes.byref(ms)))
                row.append(ms.value)
            print("      per GiB offset, plane %d -> %d: %s" % (a, b, " ".join("%.4f" % v for v in row)), flush=True)
    del img, grid, out, plain
    planes.close()
    torch.cuda.empty_cache()
    print("      after close: free %.1f GiB" % (torch.cuda.mem_get_info()[0] / 2**30), flush=True)
ctx.close()
