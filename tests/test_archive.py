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
