"""`.hgi` container: byte layout of the header, the reference's own `serde` test
(src/lib.rs:99-125) with the oracle standing in for the encoder, and the sizes SURVEY Appendix B.2 lists."""
import io
import struct
import zlib

import numpy as np
import pytest

from rustyhgi_amd import Archive, Grid, Metadata
from rustyhgi_amd.interpolator import InterpolationType
from rustyhgi_amd.quantizator import QuantizationLevel


def test_serde_round_trip_like_lib_rs(oracle):
    levels, (width, height) = 3, (8, 8)                              # src/lib.rs:101-103
    image = oracle.synth(oracle.SYNTH_XY, 0, 0, width, height)
    grid = Grid(oracle.encode(image, levels, oracle.linear_lut(0)[0]), width)
    metadata = Metadata(QuantizationLevel.Lossless, InterpolationType.Crossed, width, height, levels)
    archive = Archive(metadata, grid)
    buffer = io.BytesIO()
    archive.serialize_to_writer(buffer)
    back = Archive.deserialize_from_reader(io.BytesIO(buffer.getvalue()))
    assert back == archive                                            # :124


def test_wire_layout():
    grid = Grid(np.arange(12, dtype=np.uint8), 4)
    buf = io.BytesIO()
    Archive(Metadata(QuantizationLevel.Medium, InterpolationType.Crossed, 4, 3, 2), grid).serialize_to_writer(buf)
    raw = buf.getvalue()
    assert raw[:4] == bytes([0x55, 0xA5, 0xAD, 0xBA])               # MAGIC 0xBAADA555 little endian
    assert struct.unpack("<IIIIQ", raw[4:28]) == (2, 0, 4, 3, 2)      # bincode 1.x Metadata, 24 bytes
    body = zlib.decompress(raw[28:], -15)                            # raw DEFLATE
    assert body == struct.pack("<Q", 12) + bytes(range(12)) + struct.pack("<Q", 4)
    with pytest.raises(ValueError, match="incorrect magic number"):  # src/archive.rs:48-50
        Archive.deserialize_from_reader(io.BytesIO(b"\x00" * 40))


def test_lena_archive_sizes(oracle, lena):
    """`hgi test res/LENA.TIF` (defaults L=4 Medium): 64 kb -> ~15 kb, ratio ~4.08 (the
    deflate implementation differs from miniz, so sizes are checked to within 2 %)."""
    expect = {0: 50444, 1: 21632, 2: 16067, 3: 13934}
    for q, size in expect.items():
        grid = Grid(oracle.encode(lena, 4, oracle.linear_lut(q)[0]), 256)
        buf = io.BytesIO()
        Archive(Metadata(q, InterpolationType.Crossed, 256, 256, 4), grid).serialize_to_writer(buf)
        assert abs(len(buf.getvalue()) - size) <= 0.02 * size, (q, len(buf.getvalue()))
        back = Archive.deserialize_from_reader(io.BytesIO(buf.getvalue()))
        assert (oracle.decode(back.grid.as_image(), 4) == oracle.decode(grid.as_image(), 4)).all()


def test_auto_entropy_with_a_host_grid_ends_in_zlib():
    """device_entropy="auto" is a selection rule that can end in zlib; for a grid in host memory it does so at once (the
    C++ serialize_auto() accepts a host grid too).  device_entropy=True on a host grid still refuses."""
    grid = Grid(np.arange(64, dtype=np.uint8), 8)
    archive = Archive(Metadata(QuantizationLevel.Low, InterpolationType.Crossed, 8, 8, 2), grid)
    buf, plain = io.BytesIO(), io.BytesIO()
    assert archive.serialize_to_writer(buf, device_entropy="auto") == "zlib"
    assert archive.serialize_to_writer(plain) == "zlib"
    assert buf.getvalue() == plain.getvalue()
    assert Archive.deserialize_from_reader(io.BytesIO(buf.getvalue())) == archive
    with pytest.raises(TypeError):
        archive.serialize_to_writer(io.BytesIO(), device_entropy=True)


def _archive_bytes(width, height, body, q=2, i=0, scale=2):
    enc = zlib.compressobj(9, zlib.DEFLATED, -15)
    return struct.pack("<I", 0xBAADA555) + struct.pack("<IIIIQ", q, i, width, height, scale) + enc.compress(body) + enc.flush()


def test_reader_refuses_what_the_cpp_reader_refuses():
    """The Python reader accepts exactly the files include/hgi_archive.hpp's deserialize() accepts (tests/cpp/
    test_archive_hardening.cpp holds the same cases): the header is untrusted, so a grid whose width or size disagrees with the
    metadata, a header announcing more than the stream can hold, a stream that ends early or carries more than announced are
    all refused -- deliberately stricter than the reference's reader (src/archive.rs:43-55), which trusts both."""
    good = struct.pack("<Q", 12) + bytes(range(12)) + struct.pack("<Q", 4)
    ok = Archive.deserialize_from_reader(io.BytesIO(_archive_bytes(4, 3, good)))
    assert ok.grid.width == 4 and ok.grid.buffer.tolist() == list(range(12))
    cases = {
        "grid width does not match": _archive_bytes(4, 3, struct.pack("<Q", 12) + bytes(range(12)) + struct.pack("<Q", 6)),
        "grid size does not match": _archive_bytes(4, 3, struct.pack("<Q", 11) + bytes(range(12)) + b"\x00" + struct.pack("<Q", 4)[:7]),
        "exceeds what the stream can hold": _archive_bytes(1 << 20, 1 << 20, good),
        "corrupt grid stream": _archive_bytes(4, 3, good)[:-3],                                   # ends early
        "corrupt grid stream|": _archive_bytes(4, 3, good + b"\x00" * 40),                        # more than announced
        "corrupt grid stream||": _archive_bytes(4, 4, good),                                      # less than announced
        "truncated archive": _archive_bytes(4, 3, good)[:20],
    }
    for what, raw in cases.items():
        with pytest.raises(ValueError, match=what.rstrip("|")):
            Archive.deserialize_from_reader(io.BytesIO(raw))
    # a zip bomb behind an honest-looking header: 64 MiB of zeros announced as 4 x 3 -- never inflated beyond 28 bytes + 1
    bomb = _archive_bytes(4, 3, b"\x00" * (64 << 20))
    assert len(bomb) < 100000
    with pytest.raises(ValueError, match="corrupt grid stream"):
        Archive.deserialize_from_reader(io.BytesIO(bomb))


def test_lz77_probe_rule_matches_the_cpp_rule(tmp_path, oracle, lena):
    """`device_entropy="auto"` (Python) and serialize_auto (include/hgi_archive.hpp) must send the same grids to zlib: the
    probe -- zlib level 1 over up to 1 MiB from the middle, scaled, against three quarters of the device stream -- is
    restated in both; here both run on the same bytes and thresholds (no GPU: the device stream's size is an input)."""
    import os
    import subprocess
    from conftest import ROOT
    from rustyhgi_amd.archive import lz77_would_win
    src = tmp_path / "rule.cpp"
    src.write_text('#include <cstdio>\n#include <cstdlib>\n#include <fstream>\n#include <iterator>\n#include "hgi_archive.hpp"\n'
                   'int main(int argc, char **argv) {\n'
                   '    std::ifstream f(argv[1], std::ios::binary);\n'
                   '    hgi::Grid g;\n'
                   '    g.buffer.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());\n'
                   '    g.width = (size_t)atol(argv[2]);\n'
                   '    for (int i = 3; i < argc; ++i) printf("%d\\n", hgi::archive_detail::lz77_would_win(g, (size_t)atol(argv[i])) ? 1 : 0);\n'
                   '    return 0;\n}\n')
    exe = tmp_path / "rule"
    lib = os.path.join(ROOT, "rustyhgi_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-L", lib, "-lhgi_hip", "-lz",
                           "-Wl,-rpath," + lib, "-o", str(exe)])
    grids = {
        "periodic": oracle.synth(oracle.SYNTH_XY, 0, 0, 1920, 1080),                                   # the criterion image: LZ77 wins by far
        "lena": oracle.encode(lena, 4, oracle.linear_lut(2)[0]),
        "noise": oracle.synth(oracle.SYNTH_NOISE, 7, 0, 512, 512),
        "tiny": np.zeros((32, 32), np.uint8),                                                          # below the probe's minimum
    }
    for name, g in grids.items():
        raw = np.ascontiguousarray(g).tobytes()
        path = tmp_path / (name + ".u8")
        path.write_bytes(raw)
        # thresholds around the crossover of each grid, found from the probe itself, plus the extremes
        got1 = len(zlib.compress(raw[(len(raw) - min(len(raw), 1 << 20)) // 2:][:min(len(raw), 1 << 20)], 1))
        cross = int(got1 / min(len(raw), 1 << 20) * len(raw) / 0.75)
        sizes = [1, max(cross - 64, 1), cross, cross + 64, 10 * len(raw) + 100]
        out = subprocess.run([str(exe), str(path), str(g.shape[1])] + [str(v) for v in sizes], capture_output=True, text=True, check=True)
        cpp = [line == "1" for line in out.stdout.split()]
        py = [lz77_would_win(raw, v) for v in sizes]
        assert cpp == py, (name, sizes, cpp, py)
    assert lz77_would_win(np.ascontiguousarray(grids["periodic"]).tobytes(), 1664713)       # profiles/r03_bench_cpp.txt: device stream of that image
