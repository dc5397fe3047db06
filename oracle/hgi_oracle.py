"""ctypes front-end of oracle/hgi_oracle.c (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED by the reference's own tests -- see the header of hgi_oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# HGI_ORACLE_SO: another build of the same source (tests/test_sanitizers.py loads the ASan/UBSan one in a child process)
_SO = os.environ.get("HGI_ORACLE_SO") or os.path.join(_HERE, "_build", "libhgi_oracle.so")

LEFTTOP, CROSSED = 0, 1
LOSSLESS, LOW, MEDIUM, HIGH = 0, 1, 2, 3
SYNTH_XY, SYNTH_NOISE, SYNTH_RAMP = 0, 1, 2


def build(force=False):
    """Compile the C oracle with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "hgi_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        u8p, u32, u64, i32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int
        L.hgi_oracle_linear_lut.argtypes = [i32, u8p, u8p]
        L.hgi_oracle_linear_lut.restype = i32
        L.hgi_oracle_noop_lut.argtypes = [u8p]
        L.hgi_oracle_encode.argtypes = [u8p, u32, u32, u32, i32, u8p, u8p, u8p, ctypes.POINTER(u64)]
        L.hgi_oracle_encode.restype = i32
        L.hgi_oracle_decode.argtypes = [u8p, u32, u32, u32, i32, u8p]
        L.hgi_oracle_decode.restype = i32
        L.hgi_oracle_sq_error.argtypes = [u8p, u8p, ctypes.c_size_t, ctypes.POINTER(u64),
                                          ctypes.POINTER(u32)]
        L.hgi_oracle_sq_error.restype = u64
        L.hgi_oracle_synth.argtypes = [i32, u64, u64, u32, u32, u8p]
        L.hgi_oracle_synth.restype = i32
        L.hgi_oracle_bench_batch.argtypes = [u8p, u8p, u8p, u32, u32, u32, i32, u8p,
                                             ctypes.c_size_t, i32,
                                             ctypes.POINTER(ctypes.c_double),
                                             ctypes.POINTER(ctypes.c_double)]
        L.hgi_oracle_bench_batch.restype = ctypes.c_double
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def linear_lut(level):
    lut = np.zeros(256, np.uint8)
    err = np.zeros(1, np.uint8)
    if lib().hgi_oracle_linear_lut(int(level), _p(lut), _p(err)):
        raise ValueError("bad quantization level %r" % (level,))
    return lut, int(err[0])


def noop_lut():
    lut = np.zeros(256, np.uint8)
    lib().hgi_oracle_noop_lut(_p(lut))
    return lut


def encode(img, levels, lut, interp=CROSSED, want_rec=False):
    """img: (H, W) uint8.  Returns grid [, rec, fallbacks]."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    lut = np.ascontiguousarray(lut, dtype=np.uint8)
    assert lut.size == 256
    grid = np.empty_like(img)
    rec = np.empty_like(img) if want_rec else None
    fb = ctypes.c_uint64(0)
    rc = lib().hgi_oracle_encode(_p(img), w, h, int(levels), int(interp), _p(lut), _p(grid),
                                 _p(rec) if want_rec else None, ctypes.byref(fb))
    if rc:
        raise ValueError("hgi_oracle_encode rc=%d" % rc)
    return (grid, rec, fb.value) if want_rec else grid


def decode(grid, levels, interp=CROSSED):
    grid = np.ascontiguousarray(grid, dtype=np.uint8)
    h, w = grid.shape
    out = np.empty_like(grid)
    rc = lib().hgi_oracle_decode(_p(grid), w, h, int(levels), int(interp), _p(out))
    if rc:
        raise ValueError("hgi_oracle_decode rc=%d" % rc)
    return out


def sq_error(before, after):
    """Returns (sum_sq, integer_mse, max_abs) as `hgi test` computes them (src/main.rs:84-106)."""
    before = np.ascontiguousarray(before, dtype=np.uint8)
    after = np.ascontiguousarray(after, dtype=np.uint8)
    mse, mx = ctypes.c_uint64(0), ctypes.c_uint32(0)
    sd = lib().hgi_oracle_sq_error(_p(before), _p(after), before.size, ctypes.byref(mse),
                                   ctypes.byref(mx))
    return int(sd), int(mse.value), int(mx.value)


def synth(kind, seed, frame, w, h):
    out = np.empty((h, w), np.uint8)
    if lib().hgi_oracle_synth(int(kind), int(seed), int(frame), w, h, _p(out)):
        raise ValueError("bad synth kind")
    return out


def bench_batch(imgs, levels, lut, threads, interp=CROSSED):
    """imgs: (F, H, W) uint8.  Returns dict(wall_s, enc_cpu_s, dec_cpu_s, grids, outs)."""
    imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
    f, h, w = imgs.shape
    grids, outs = np.empty_like(imgs), np.empty_like(imgs)
    lut = np.ascontiguousarray(lut, dtype=np.uint8)
    e, d = ctypes.c_double(0), ctypes.c_double(0)
    wall = lib().hgi_oracle_bench_batch(_p(imgs), _p(grids), _p(outs), w, h, int(levels),
                                        int(interp), _p(lut), f, int(threads),
                                        ctypes.byref(e), ctypes.byref(d))
    return dict(wall_s=wall, enc_cpu_s=e.value, dec_cpu_s=d.value, grids=grids, outs=outs)
