"""A/B timing of several builds of the library in ONE process on the SAME placed planes (process-to-process and
placement noise is as large as the effects being looked for).  Interleaves the builds round by round and prints the
per-build median of the bench step (encode then decode; default 64 x 4096^2, L4, Medium; AB_F / AB_W / AB_H / AB_LEVELS /
AB_QUANT in the environment select another shape, e.g. C4: AB_F=1 AB_W=16384 AB_H=16384 AB_LEVELS=8 AB_QUANT=3).
usage: ab.py [-r rounds] [-s steps] variant ...      ("-" = rustyhgi_amd/libhgi_hip.so, "_x" = libhgi_hip_x.so)"""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rustyhgi_amd import _ffi
args = sys.argv[1:]
rounds, steps = 5, 12
while args and args[0] in ("-r", "-s"):
    if args[0] == "-r": rounds = int(args[1])
    else: steps = int(args[1])
    args = args[2:]
variants = args or ["-"]
NF = int(os.environ.get("AB_F", "64")); W = int(os.environ.get("AB_W", "4096")); Hh = int(os.environ.get("AB_H", "4096")); n = NF * W * Hh
LEVELS = int(os.environ.get("AB_LEVELS", "4")); QUANT = int(os.environ.get("AB_QUANT", "2"))      # C4: AB_F=1 AB_W=16384 AB_H=16384 AB_LEVELS=8 AB_QUANT=3
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
_ffi._share_torch_hip_runtime()
libs = {}
for v in variants:
    path = os.path.join(ROOT, "rustyhgi_amd", "libhgi_hip%s.so" % ("" if v == "-" else v))
    L = ctypes.CDLL(path)
    for name, res, argt in _ffi.SYMBOLS:
        try:
            fn = getattr(L, name)
        except AttributeError:
            continue
        fn.restype, fn.argtypes = res, argt
    h = ctypes.c_void_p()
    assert L.hgi_ctx_create(0, ctypes.byref(h)) == 0
    assert L.hgi_ctx_set_stream(h, ctypes.c_void_p(stream.cuda_stream)) == 0
    libs[v] = (L, h)
L0, h0 = libs[variants[0]]
planes = (ctypes.c_void_p * 3)(); sep = ctypes.c_int(0)
assert L0.hgi_planes_alloc(h0, n, 3, planes, ctypes.byref(sep)) == 0, L0.hgi_last_error()
img, grid, out = [int(p) for p in planes]
print("planes separated:", bool(sep.value), " %d x %dx%d levels %d quant %d" % (NF, W, Hh, LEVELS, QUANT))
lut = np.zeros(256, np.uint8); err = np.zeros(1, np.uint8)
assert L0.hgi_linear_lut(QUANT, lut.ctypes.data, err.ctypes.data) == 0
assert L0.hgi_synth_u8_dev(h0, _ffi.SYNTH_RAMP, 0x48474930 + 3, 0, W, Hh, img, NF, W * Hh) == 0
class _Arr:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2, "strides": None}
res = {v: ([], []) for v in variants}
ref_sum = None      # every build must produce the first build's bytes (cheap fingerprint: strided samples of grid and output)
for rnd in range(rounds):
    for v in variants:
        L, h = libs[v]
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
        for k in range(steps + 2):
            e = ev[max(k - 2, 0)]
            e[0].record()
            assert L.hgi_encode_u8_dev(h, img, W, Hh, LEVELS, 1, lut.ctypes.data, grid, NF, W * Hh) == 0, L.hgi_last_error()
            e[1].record()
            assert L.hgi_decode_u8_dev(h, grid, W, Hh, LEVELS, 1, out, NF, W * Hh) == 0
            e[2].record()
        torch.cuda.synchronize()
        if rnd == 0:
            gt = torch.as_tensor(_Arr(grid, n), device="cuda"); ot = torch.as_tensor(_Arr(out, n), device="cuda")
            fp = (int(gt[::257].to(torch.int64).sum()), int(ot[::263].to(torch.int64).sum()), int(gt[-4096:].to(torch.int64).sum()))
            if ref_sum is None: ref_sum = fp
            print("variant[%-6s] fingerprint %s %s" % (v, fp, "(same bytes)" if fp == ref_sum else "*** DIFFERENT BYTES ***"))
        res[v][0].append(float(np.mean([e[0].elapsed_time(e[1]) for e in ev])))
        res[v][1].append(float(np.mean([e[1].elapsed_time(e[2]) for e in ev])))
base = None
for v in variants:
    e, d = np.median(res[v][0]), np.median(res[v][1])
    if base is None: base = (e, d)
    print("variant[%-6s] encode %.4f (%+5.1f %%)  decode %.4f (%+5.1f %%)   min %.4f / %.4f  max %.4f / %.4f" %
          (v, e, 100 * (e / base[0] - 1), d, 100 * (d / base[1] - 1), min(res[v][0]), min(res[v][1]), max(res[v][0]), max(res[v][1])))
