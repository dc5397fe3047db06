"""The entropy stage over the C3 shard's 64 grids (4096^2, Medium) into a pinned host buffer: one slot per stream
(hgi_deflate_grids_dev: one download per frame) against packed (hgi_deflate_grids_packed_dev: one download per group).
HGI_ENTROPY_GROUP_MIB selects the group size (read once per process: run once per value)."""
import os, sys, time, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustyhgi_amd as H
from rustyhgi_amd import _ffi
F, S = 64, 4096
ctx = H.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
imgs = torch.empty((F, S, S), dtype=torch.uint8, device="cuda"); grids = torch.empty_like(imgs)
L = _ffi.lib()
lut = np.zeros(256, np.uint8); err = np.zeros(1, np.uint8)
_ffi.check(L.hgi_linear_lut(2, lut.ctypes.data, err.ctypes.data))
_ffi.check(L.hgi_synth_u8_dev(ctx.handle, _ffi.SYNTH_RAMP, 0x48474930 + 3, 0, S, S, imgs.data_ptr(), F, S * S))
_ffi.check(L.hgi_encode_u8_dev(ctx.handle, imgs.data_ptr(), S, S, 4, 1, lut.ctypes.data, grids.data_ptr(), F, S * S))
torch.cuda.synchronize()
cap = S * S // 2 + 4096
out = torch.zeros((F, cap), dtype=torch.uint8, pin_memory=True)
sizes, offs = (ctypes.c_size_t * F)(), (ctypes.c_size_t * F)()
def strided(): _ffi.check(L.hgi_deflate_grids_dev(ctx.handle, grids.data_ptr(), S, S, F, S * S, out.data_ptr(), cap, sizes))
def packed(): _ffi.check(L.hgi_deflate_grids_packed_dev(ctx.handle, grids.data_ptr(), S, S, F, S * S, out.data_ptr(), F * cap, offs, sizes))
def med(fn, reps=7):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[reps // 2] * 1e3
ts, tp = med(strided), med(packed)
total = sum(sizes)
print("group %4s MiB: one copy per frame %.2f ms; packed %.2f ms = %.1f GB/s of stream (%d bytes), %.0f GB/s of grid" % (
    os.environ.get("HGI_ENTROPY_GROUP_MIB", "256"), ts, tp, total / tp / 1e6, total, F * S * S / tp / 1e6))
