"""The entropy stage's host side (hgi_huffman_plan, include/hgi.h): code lengths, canonical codes and the RFC 1951 block
header for a histogram of literal / length symbols, checked by tokenising data the way the device kernels do (a Python
restatement of the rule in hgi_entropy.hip), packing a complete stream with the plan and handing it to zlib's inflate --
any inflate must read what the device writes (the reference's flate2 DeflateDecoder reads raw DEFLATE the same way,
src/archive.rs:52-53)."""
import ctypes
import zlib

import numpy as np
import pytest

from rustyhgi_amd import _ffi

NSYM, CHUNK = 286, 1024
LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LBITS = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]


def length_symbol(length):
    s = max(i for i in range(29) if LBASE[i] <= length)
    return 257 + s, LBITS[s], length - LBASE[s]


THRESHOLDS = (3, 4, 6, 10)          # kMatchThresholdHost (rustyhgi_amd/csrc/hgi_kernels.h)


def tokens(data, min_match=3):
    """Literal bytes and ('m', length) run matches: a byte equal to its predecessor inside its 1 KiB chunk continues a
    run; the run's bytes after its head are cut into pieces of 258, pieces of >= min_match become matches of distance 1."""
    out = []
    for c0 in range(0, len(data), CHUNK):
        chunk = data[c0:c0 + CHUNK]
        i = 0
        while i < len(chunk):
            out.append(chunk[i])                 # head
            j = i + 1
            while j < len(chunk) and chunk[j] == chunk[j - 1]:
                j += 1
            m = j - i - 1                        # continuing bytes
            while m > 0:
                piece = min(m, 258)
                if piece >= min_match:
                    out.append(("m", piece))
                else:
                    out.extend([chunk[i]] * piece)
                m -= piece
            i = j
    return out


def plan(hist):
    hist = np.ascontiguousarray(hist, np.uint64)
    assert hist.shape == (NSYM,)
    lens = np.zeros(NSYM, np.uint8)
    codes = np.zeros(NSYM, np.uint16)
    header = np.zeros(640, np.uint8)
    bits = ctypes.c_size_t(0)
    _ffi.check(_ffi.lib().hgi_huffman_plan(hist.ctypes.data, lens.ctypes.data, codes.ctypes.data, header.ctypes.data, header.size,
                                           ctypes.byref(bits)))
    return lens, codes, header, bits.value


def pack(toks, lens, codes, header, header_bits):
    """header + tokens + end of block, least significant bit first (what the device kernels produce)"""
    out = bytearray(header[:(header_bits + 7) // 8].tobytes())
    at = [header_bits]

    def put(v, n):
        for i in range(n):
            if at[0] >> 3 >= len(out):
                out.append(0)
            out[at[0] >> 3] |= ((v >> i) & 1) << (at[0] & 7)
            at[0] += 1

    for t in list(toks) + [256]:
        if isinstance(t, tuple):
            sym, eb, ex = length_symbol(t[1])
            assert lens[sym] > 0
            put(int(codes[sym]), int(lens[sym]))
            put(ex, eb)
            put(0, 1)                            # distance code 0 = distance 1: the one-bit code "0"
        else:
            assert lens[t] > 0, "symbol %d has no code" % t
            put(int(codes[t]), int(lens[t]))
    return bytes(out)


def histogram(toks):
    hist = np.zeros(NSYM, np.uint64)
    for t in toks:
        hist[length_symbol(t[1])[0] if isinstance(t, tuple) else t] += 1
    hist[256] = 1
    return hist


def roundtrip(data):
    data = bytes(data)
    toks = tokens(data)
    hist = histogram(toks)
    lens, codes, header, bits = plan(hist)
    assert lens.max() <= 15 and lens[256] > 0
    # a complete prefix code over the used symbols (Kraft sum exactly 1), zero length exactly for unused ones
    used = lens > 0
    assert (used == (hist > 0)).all()
    if used.sum() > 1:
        assert sum(2.0 ** -int(l) for l in lens[used]) == 1.0
    stream = pack(toks, lens, codes, header, bits)
    assert zlib.decompressobj(-15).decompress(stream) == data
    return len(stream)


def token_bits(toks, lens):
    total = 0
    for t in toks:
        if isinstance(t, tuple):
            sym, eb, _ = length_symbol(t[1])
            total += int(lens[sym]) + eb + 1
        else:
            total += int(lens[t])
    return total


def stage_stream(grid_bytes, width):
    """The whole stream the entropy stage writes for a grid, restated: the grid's tokens under each candidate threshold,
    the code each histogram asks for, the smallest payload wins (first on a tie); around the tokens the eight literals of
    the u64 length and of the u64 width (never part of a run), then end of block."""
    import struct
    grid_bytes = bytes(grid_bytes)
    prefix, suffix = list(struct.pack("<Q", len(grid_bytes))), list(struct.pack("<Q", width))
    best = None
    for mm in THRESHOLDS:
        toks = prefix + tokens(grid_bytes, mm) + suffix
        hist = histogram(toks)
        lens, codes, header, bits = plan(hist)
        payload = token_bits(toks + [256], lens)
        if best is None or payload < best[0]:
            best = (payload, toks, lens, codes, header, bits)
        if not grid_bytes:
            break
    _, toks, lens, codes, header, bits = best
    return pack(toks, lens, codes, header, bits)


def zlib_size(data, strategy):
    co = zlib.compressobj(9, zlib.DEFLATED, -15, 9, strategy)
    return len(co.compress(data) + co.flush())


def test_plan_on_residual_like_data():
    rng = np.random.default_rng(7)
    # noise around zero mod 256, like a busy residual grid: no runs to speak of, the Huffman code is everything
    data = (rng.normal(0, 6, 50000).round().astype(np.int64) % 256).astype(np.uint8).tobytes()
    assert roundtrip(data) <= zlib_size(data, zlib.Z_HUFFMAN_ONLY) * 1.01
    # mostly zeros with bursts, like a smooth image at Medium: the run matches carry it -- on par with zlib's own RLE
    smooth = np.zeros(200000, np.uint8)
    at = rng.integers(0, smooth.size, 6000)
    smooth[at] = rng.integers(1, 256, at.size)
    smooth = smooth.tobytes()
    n = roundtrip(smooth)
    assert n <= zlib_size(smooth, zlib.Z_RLE) * 1.03, (n, zlib_size(smooth, zlib.Z_RLE))
    assert n < zlib_size(smooth, zlib.Z_HUFFMAN_ONLY) * 0.6


def test_length_symbols_cover_every_length():
    for length in range(3, 259):
        sym, eb, ex = length_symbol(length)
        assert 257 <= sym <= 285 and LBASE[sym - 257] + ex == length and 0 <= ex < (1 << eb) + (eb == 0)


@pytest.mark.parametrize("case", ["two_symbols", "all_equal", "one_dominant", "fibonacci", "every_byte_once", "runs_of_every_length",
                                  "run_across_chunks", "empty"])
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
        data = b"".join(bytes([i, 255 - i]) * min(f, 100000) for i, f in enumerate(fib))
    elif case == "every_byte_once":
        data = bytes(range(256))
    elif case == "runs_of_every_length":   # 1 .. 300 repeats, separated by a changing byte: every length symbol, 258 + leftovers
        data = b"".join(bytes([1 + r % 200]) + bytes([0]) * r for r in range(1, 301))
    elif case == "run_across_chunks":      # a run is cut where a 1 KiB chunk ends
        data = bytes([9]) * 5000 + bytes([3, 3]) + bytes([0]) * 1023
    else:
        data = b""
    if case == "empty":                # only the end-of-block symbol: a single code of one bit
        hist = np.zeros(NSYM, np.uint64)
        hist[256] = 1
        lens, codes, header, bits = plan(hist)
        assert lens[256] == 1 and lens[:256].max() == 0
        assert zlib.decompressobj(-15).decompress(pack([], lens, codes, header, bits)) == b""
        return
    roundtrip(data)


def test_plan_rejects_bad_arguments():
    L = _ffi.lib()
    bits = ctypes.c_size_t(0)
    z = np.zeros(NSYM, np.uint64)
    lens, codes, header = np.zeros(NSYM, np.uint8), np.zeros(NSYM, np.uint16), np.zeros(640, np.uint8)
    assert L.hgi_huffman_plan(z.ctypes.data, lens.ctypes.data, codes.ctypes.data, header.ctypes.data, 640, ctypes.byref(bits)) == _ffi.EINVAL
    z[256] = 1
    assert L.hgi_huffman_plan(z.ctypes.data, lens.ctypes.data, codes.ctypes.data, header.ctypes.data, 2, ctypes.byref(bits)) == _ffi.EINVAL
    assert L.hgi_huffman_plan(None, lens.ctypes.data, codes.ctypes.data, header.ctypes.data, 640, ctypes.byref(bits)) == _ffi.EINVAL


def test_restated_stage_stream_is_ordinary_deflate():
    """stage_stream (what the GPU tests compare the device's bytes with) inflates to the grid's bincode image, and its
    adaptive threshold never loses to the fixed one."""
    import struct
    rng = np.random.default_rng(11)
    for w, h in ((64, 40), (259, 9), (1, 1), (5, 0)):
        data = (rng.geometric(0.5, w * h) - 1).astype(np.uint8).tobytes() if w * h else b""
        s = stage_stream(data, w)
        assert zlib.decompressobj(-15).decompress(s) == struct.pack("<Q", w * h) + data + struct.pack("<Q", w)
