"""The entropy stage's host side (hgi_huffman_plan, include/hgi.h): code lengths, canonical codes and the RFC 1951 block
header for a histogram of literals, checked by packing a complete stream with them in Python and handing it to zlib's
inflate -- any inflate must read what the device writes with the same plan (the reference's flate2 DeflateDecoder reads
raw DEFLATE the same way, src/archive.rs:52-53)."""
import ctypes
import zlib

import numpy as np
import pytest

from rustyhgi_amd import _ffi


def plan(hist):
    hist = np.ascontiguousarray(hist, np.uint64)
    assert hist.shape == (257,)
    lens = np.zeros(257, np.uint8)
    codes = np.zeros(257, np.uint16)
    header = np.zeros(512, np.uint8)
    bits = ctypes.c_size_t(0)
    _ffi.check(_ffi.lib().hgi_huffman_plan(hist.ctypes.data, lens.ctypes.data, codes.ctypes.data, header.ctypes.data, header.size,
                                           ctypes.byref(bits)))
    return lens, codes, header, bits.value


def pack(data, lens, codes, header, header_bits):
    """header + the literals of `data` + end of block, least significant bit first (what the device kernels produce)"""
    out = bytearray(header[:(header_bits + 7) // 8].tobytes())
    at = header_bits
    syms = list(data) + [256]
    for s in syms:
        assert lens[s] > 0, "symbol %d has no code" % s
        v, n = int(codes[s]), int(lens[s])
        for i in range(n):
            if at >> 3 >= len(out):
                out.append(0)
            out[at >> 3] |= ((v >> i) & 1) << (at & 7)
            at += 1
    return bytes(out)


def roundtrip(data):
    data = bytes(data)
    hist = np.bincount(np.frombuffer(data, np.uint8), minlength=256).astype(np.uint64)
    hist = np.append(hist, np.uint64(1))
    lens, codes, header, bits = plan(hist)
    assert lens.max() <= 15 and lens[256] > 0
    # a complete prefix code over the used symbols (Kraft sum exactly 1), zero length exactly for unused ones
    used = lens > 0
    assert (used == (hist > 0)).all()
    assert sum(2.0 ** -int(l) for l in lens[used]) == 1.0
    stream = pack(data, lens, codes, header, bits)
    assert zlib.decompressobj(-15).decompress(stream) == data
    return len(stream)


def test_plan_on_residual_like_data():
    rng = np.random.default_rng(7)
    # noise around zero mod 256, like a residual grid
    data = (rng.normal(0, 6, 50000).round().astype(np.int64) % 256).astype(np.uint8).tobytes()
    n = roundtrip(data)
    co = zlib.compressobj(9, zlib.DEFLATED, -15, 9, zlib.Z_HUFFMAN_ONLY)
    ref = len(co.compress(data) + co.flush())
    assert n <= ref * 1.01, (n, ref)             # as tight as zlib's own Huffman-only stream


@pytest.mark.parametrize("case", ["two_symbols", "all_equal", "one_dominant", "fibonacci", "every_byte_once", "empty"])
def test_plan_corner_cases(case):
    if case == "two_symbols":
        data = bytes([0, 0, 0, 7] * 10)
    elif case == "all_equal":
        data = bytes(range(256)) * 16
    elif case == "one_dominant":
        data = bytes([0] * 100000 + list(range(1, 256)))
    elif case == "fibonacci":          # frequencies that make an unbounded Huffman tree 30+ levels deep: the 15-bit limit
        fib = [1, 1]
        while len(fib) < 34:
            fib.append(fib[-1] + fib[-2])
        data = b"".join(bytes([i]) * min(f, 200000) for i, f in enumerate(fib))
    elif case == "every_byte_once":
        data = bytes(range(256))
    else:
        data = b""
    if case == "empty":                # only the end-of-block symbol: a single code of one bit
        lens, codes, header, bits = plan(np.append(np.zeros(256, np.uint64), np.uint64(1)))
        assert lens[256] == 1 and lens[:256].max() == 0
        assert zlib.decompressobj(-15).decompress(pack(b"", lens, codes, header, bits)) == b""
        return
    roundtrip(data)


def test_plan_rejects_bad_arguments():
    L = _ffi.lib()
    bits = ctypes.c_size_t(0)
    z = np.zeros(257, np.uint64)
    lens, codes, header = np.zeros(257, np.uint8), np.zeros(257, np.uint16), np.zeros(512, np.uint8)
    assert L.hgi_huffman_plan(z.ctypes.data, lens.ctypes.data, codes.ctypes.data, header.ctypes.data, 512, ctypes.byref(bits)) == _ffi.EINVAL
    z[256] = 1
    assert L.hgi_huffman_plan(z.ctypes.data, lens.ctypes.data, codes.ctypes.data, header.ctypes.data, 2, ctypes.byref(bits)) == _ffi.EINVAL
    assert L.hgi_huffman_plan(None, lens.ctypes.data, codes.ctypes.data, header.ctypes.data, 512, ctypes.byref(bits)) == _ffi.EINVAL
