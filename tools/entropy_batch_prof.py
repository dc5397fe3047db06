"""The entropy stage's batch call under rocprofv3 (kernel trace): 64 C3 grids, three calls."""
import os, sys, time, ctypes, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rustyhgi_amd as H
from rustyhgi_amd import _ffi
F, S = 64, 4096
ctx = H.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
imgs = torch.empty((F, S, S), dtype=torch.uint8, device="cuda"); grids = torch.empty_like(imgs)
L = _ffi.lib()
lut = np.ascontiguousarray(H.quantizator.Linear(H.quantizator.QuantizationLevel.Medium).table())
_ffi.check(L.hgi_synth_u8_dev(ctx.handle, _ffi.SYNTH_RAMP, 0x48474930 + 3, 0, S, S, imgs.data_ptr(), F, S * S))
_ffi.check(L.hgi_encode_u8_dev(ctx.handle, imgs.data_ptr(), S, S, 4, 1, lut.ctypes.data, grids.data_ptr(), F, S * S))
torch.cuda.synchronize()
cap = S * S // 2
out = torch.zeros((F, cap), dtype=torch.uint8, pin_memory=bool(os.environ.get("PINNED")))
sizes = (ctypes.c_size_t * F)()
for i in range(4):
    t0 = time.perf_counter()
    _ffi.check(L.hgi_deflate_grids_dev(ctx.handle, grids.data_ptr(), S, S, F, S * S, out.data_ptr(), cap, sizes))
    print("call %d: %.2f ms, %d bytes" % (i, (time.perf_counter() - t0) * 1e3, sum(sizes)))
