"""P_fine on the record (the north star's "level-0 interpolation pass", reference src/utils.rs:16-18 with
e = 1): the product kernels at levels = 1 on the C3 shard (64 x 4096^2, Medium, Crossed), on placed planes, encode and
decode alternating like the bench step.  At levels = 1 the finest pass IS the whole job: 3/4 of the pixels are new
(1.75 B/px algorithmic), the even/even quarter is the lattice, which the launch copies through (2 B/px cross HBM).
Then the level-wise path at levels = 4 -- one launch per level -- so that a kernel trace also shows the finest pass as a
launch of its own (k_encode_level / k_decode_level with substep 1).
Run plain, or under `rocprofv3 --kernel-trace --stats` (tools/profile.sh pass `pfine`)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustyhgi_amd as H
from rustyhgi_amd import _ffi
L = _ffi.lib()
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
fused = H.Context(0); fused.set_stream(stream.cuda_stream)
W = Hh = 4096; NF = 64; n = NF * W * Hh
lut = np.zeros(256, np.uint8); err = np.zeros(1, np.uint8)
_ffi.check(L.hgi_linear_lut(2, lut.ctypes.data, err.ctypes.data))
planes = H.Planes(fused, n, 3)
img, grid, out = (planes.torch(i, (n,)) for i in range(3))
_ffi.check(L.hgi_synth_u8_dev(fused.handle, _ffi.SYNTH_RAMP, 0x48474930 + 3, 0, W, Hh, img.data_ptr(), NF, W * Hh))
def alt(ctx, levels, reps):
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(reps)]
    for k in range(reps + 2):
        e = ev[max(k - 2, 0)]
        e[0].record()
        _ffi.check(L.hgi_encode_u8_dev(ctx.handle, img.data_ptr(), W, Hh, levels, 1, lut.ctypes.data, grid.data_ptr(), NF, W * Hh))
        e[1].record()
        _ffi.check(L.hgi_decode_u8_dev(ctx.handle, grid.data_ptr(), W, Hh, levels, 1, out.data_ptr(), NF, W * Hh))
        e[2].record()
    torch.cuda.synchronize()
    return float(np.mean([e[0].elapsed_time(e[1]) for e in ev])), float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
te, td = alt(fused, 1, 20)
alg, moved = 1.75 * n, 2.0 * n
mx = int((img[::7].to(torch.int16) - out[::7].to(torch.int16)).abs().max().item())
print("planes separated: %s" % planes.separated)
print("P_fine alone (k_enc_tiles / k_dec_tiles at levels=1, 64 x 4096^2): encode %.4f ms  decode %.4f ms  max abs err %d" % (te, td, mx))
for name, t in (("encode", te), ("decode", td)):
    print("  %s: algorithmic 1.75 B/px -> %.0f GB/s = %.3f of 8 TB/s;  moved 2 B/px -> %.0f GB/s = %.3f" %
          (name, alg / t * 1e-6, alg / t * 1e-6 / 8000, moved / t * 1e-6, moved / t * 1e-6 / 8000))
lw = H.Context(0); lw.set_stream(stream.cuda_stream); lw.set_path(_ffi.PATH_LEVELWISE)
te4, td4 = alt(lw, 4, 4)
print("level-wise path, levels=4 (one launch per level; its last launch is the finest pass): encode %.4f ms  decode %.4f ms" % (te4, td4))
del img, grid, out
planes.close()
