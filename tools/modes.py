"""Where do the two run-time "modes" of the tile kernels come from?  (decode 0.374 vs 0.39 ms, constant
within a process.)  One process, several independently allocated buffer triples: if the time follows the buffers, it is
placement (physical pages / TLB fragments / channel alignment); if every triple runs alike, it is the process (queue,
clocks).  usage: modes.py [triples] [frames]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustyhgi_amd as H
from rustyhgi_amd import _ffi
L = _ffi.lib()
NT = int(sys.argv[1]) if len(sys.argv) > 1 else 5
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W = Hh = 4096; n = NF * W * Hh
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = H.Context(0); ctx.set_stream(stream.cuda_stream)
lut = np.zeros(256, np.uint8); err = np.zeros(1, np.uint8)
_ffi.check(L.hgi_linear_lut(2, lut.ctypes.data, err.ctypes.data))
bufs = [[torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(3)] for _ in range(NT)]
for t in bufs:
    _ffi.check(L.hgi_synth_u8_dev(ctx.handle, _ffi.SYNTH_RAMP, 0x48474933 + 3, 0, W, Hh, t[0].data_ptr(), NF, W * Hh))
def enc(a, b): _ffi.check(L.hgi_encode_u8_dev(ctx.handle, a.data_ptr(), W, Hh, 4, 1, lut.ctypes.data, b.data_ptr(), NF, W * Hh))
def dec(a, b): _ffi.check(L.hgi_decode_u8_dev(ctx.handle, a.data_ptr(), W, Hh, 4, 1, b.data_ptr(), NF, W * Hh))
def alt(i, g, o, reps=12):
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(reps)]
    for k in range(reps + 2):
        e = ev[max(k - 2, 0)]
        e[0].record(); enc(i, g); e[1].record(); dec(g, o); e[2].record()
    torch.cuda.synchronize()
    return (float(np.mean([e[0].elapsed_time(e[1]) for e in ev])), float(np.mean([e[1].elapsed_time(e[2]) for e in ev])))
for rnd in range(2):
    for t in range(NT):
        i, g, o = bufs[t]
        te, td = alt(i, g, o)
        print("round %d triple %d  img %#x grid %#x out %#x : encode %.4f  decode %.4f" % (rnd, t, i.data_ptr(), g.data_ptr(), o.data_ptr(), te, td))
# cross combinations: the grid of triple 0 decoded into every out buffer; every grid into out 0
for t in range(NT):
    enc(bufs[0][0], bufs[t][1])
for t in range(NT):
    te, td = alt(bufs[0][0], bufs[0][1], bufs[t][2])
    print("grid 0 -> out %d : encode %.4f decode %.4f" % (t, te, td))
for t in range(NT):
    te, td = alt(bufs[0][0], bufs[t][1], bufs[0][2])
    print("img 0 -> grid %d -> out 0 : encode %.4f decode %.4f" % (t, te, td))
