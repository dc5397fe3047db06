"""Placement classes, part 4: how many classes are there, and how does the allocator hand them out?
N separate 1 GiB allocations; class of each relative to buffer 0 by decode time (0 -> j); consistency on random pairs."""
import os, sys, random, subprocess, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustyhgi_amd as H
from rustyhgi_amd import _ffi
L = _ffi.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for cmd in (["rocm-smi", "--showmemorypartition", "--showcomputepartition"], ["rocm-smi", "--showmeminfo", "vram"]):
    try:
        print(subprocess.run(cmd, capture_output=True, text=True, timeout=60).stdout)
    except Exception as e:
        print(cmd, "failed:", e)
for p in ("/sys/class/drm/card0/device/current_memory_partition", "/sys/class/drm/card1/device/current_memory_partition",
          "/sys/class/drm/card0/device/current_compute_partition"):
    try: print(p, open(p).read().strip())
    except Exception as e: print(p, e)
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 64; W = Hh = 4096; n = NF * W * Hh
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = H.Context(0); ctx.set_stream(stream.cuda_stream)
lut = np.zeros(256, np.uint8); err = np.zeros(1, np.uint8)
_ffi.check(L.hgi_linear_lut(2, lut.ctypes.data, err.ctypes.data))
bufs = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(N)]
_ffi.check(L.hgi_synth_u8_dev(ctx.handle, _ffi.SYNTH_RAMP, 0x48474933 + 3, 0, W, Hh, bufs[0].data_ptr(), NF, W * Hh))
def dec(a, b): _ffi.check(L.hgi_decode_u8_dev(ctx.handle, a.data_ptr(), W, Hh, 4, 1, b.data_ptr(), NF, W * Hh))
def timed(fn, reps=6):
    fn(); fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
t = [0.0] + [timed(lambda: dec(bufs[0], bufs[j])) for j in range(1, N)]
thr = (min(t[1:]) + max(t[1:])) / 2
cls = [None] + [int(x < thr) for x in t[1:]]     # 1 = fast with buffer 0 = "other class than buffer 0"
print("threshold %.4f  fast %d slow %d" % (thr, sum(c == 1 for c in cls[1:]), sum(c == 0 for c in cls[1:])))
for j in range(1, N):
    print("buf %2d at %#x : decode(0 -> %2d) %.4f  %s" % (j, bufs[j].data_ptr(), j, t[j], "FAST" if cls[j] else "slow"))
print("pattern:", "".join("F" if c else "s" for c in cls[1:]))
random.seed(1)
ok = bad = 0
for _ in range(40):
    j, k = random.sample(range(1, N), 2)
    tt = timed(lambda: dec(bufs[j], bufs[k]), 4)
    pred_fast = cls[j] != cls[k]
    good = (tt < thr) == pred_fast
    ok += good; bad += not good
    print("pair %2d -> %2d : %.4f  predicted %s  %s" % (j, k, tt, "FAST" if pred_fast else "slow", "" if good else "MISMATCH"))
print("two-class model: %d consistent, %d not" % (ok, bad))
# one big allocation: are its 1 GiB planes of one class?
del bufs[N // 2:]
torch.cuda.empty_cache()
big = torch.empty(16 * (1 << 30), dtype=torch.uint8, device="cuda")
class P:
    def __init__(s, p): s.p = p
    def data_ptr(s): return s.p
for i in range(16):
    tt = timed(lambda: dec(bufs[0], P(big.data_ptr() + i * (1 << 30))), 4)
    print("big 16 GiB allocation, plane %2d : decode(0 -> plane) %.4f %s" % (i, tt, "FAST" if tt < thr else "slow"))
