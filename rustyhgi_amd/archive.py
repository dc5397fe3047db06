"""Mirror of the reference's `.hgi` container (src/archive.rs:13-56), byte-compatible on the wire:

    55 A5 AD BA                      MAGIC 0xBAADA555, u32 little endian            (:13, :32)
    bincode 1.x(Metadata)            u32 quantization_level (variant index), u32 interpolation,
                                     u32 width, u32 height, u64 scale_level          (:15-22, :33)
    raw DEFLATE, best compression, of bincode 1.x(Grid) = u64 len | bytes | u64 width  (:34-39; src/grid.rs:2-5)

bincode 1.x defaults: little endian, fixed-width integers, enum variant as u32, usize as u64, Vec
length as u64.  Any valid raw-DEFLATE stream decodes, so archives written here are
readable by the reference and vice versa; the compressed bytes themselves need not match miniz's.
This is host-side I/O around the codec, not part of the hot path.
"""
import struct
import zlib

import numpy as np

from .grid import Grid
from .interpolator import InterpolationType
from .quantizator import QuantizationLevel

MAGIC = 0xBAADA555


class Metadata:
    """src/archive.rs:15-22."""

    def __init__(self, quantization_level, interpolation, width, height, scale_level):
        self.quantization_level = QuantizationLevel(quantization_level)
        self.interpolation = InterpolationType(interpolation)
        self.width, self.height, self.scale_level = int(width), int(height), int(scale_level)

    def __eq__(self, o):
        return isinstance(o, Metadata) and vars(self) == vars(o)

    def __repr__(self):
        return "Metadata(%s, %s, %dx%d, scale_level=%d)" % (self.quantization_level.name, self.interpolation.name,
                                                            self.width, self.height, self.scale_level)


def lz77_would_win(raw, device_stream_bytes):
    """The selection rule of device_entropy="auto" (and of include/hgi_archive.hpp's serialize_auto): zlib level 1 over up
    to 1 MiB from the middle of the grid, scaled to the whole grid, against the device stream's exact size."""
    n = len(raw)
    probe = min(n, 1 << 20)
    if probe < 4096:
        return False
    at = (n - probe) // 2
    got = len(zlib.compress(raw[at:at + probe], 1))
    return got / probe * n <= 0.75 * device_stream_bytes


class Archive:
    """src/archive.rs:24-28, with G = Grid."""

    def __init__(self, metadata, grid):
        self.metadata, self.grid = metadata, grid

    def __eq__(self, o):
        return isinstance(o, Archive) and self.metadata == o.metadata and self.grid == o.grid

    def serialize_to_writer(self, w, device_entropy=False):
        """src/archive.rs:31-41.  device_entropy=True (grid buffer = a CUDA tensor): the DEFLATE stream is written by the
        device's entropy stage (rustyhgi_amd.entropy.deflate_grid: Huffman-coded literals and run matches) instead of zlib at
        level 9 -- the same container, readable by the same readers, two orders of magnitude sooner.
        device_entropy="auto": the device stream unless an LZ77 probe of the grid (zlib level 1 on up to 1 MiB from its
        middle -- it finds long-distance repeats as surely as level 9) predicts a stream more than a quarter smaller: the
        device codes literals and runs only, so an exactly periodic grid (the criterion harness's `(x*y) as u8` frame,
        benches/bench.rs:26-28: 19x smaller under LZ77) goes to zlib the way the reference writes it.  With the grid in
        HOST memory "auto" ends in zlib at once (there is no device stream to weigh; the C++ serialize_auto() uploads a host
        grid instead -- the archives are readable alike), while device_entropy=True insists and raises TypeError.
        Returns which writer produced the stream: "device" or "zlib" (both truthy; the reference's method returns
        Result<(), _>, so nothing meaningful can have been tested on the old None)."""
        m = self.metadata
        w.write(struct.pack("<I", MAGIC))
        w.write(struct.pack("<IIIIQ", int(m.quantization_level), int(m.interpolation), m.width, m.height, m.scale_level))
        buf = self.grid.buffer
        raw = None
        on_device = type(buf).__module__.startswith("torch") and bool(getattr(buf, "is_cuda", False))
        if device_entropy == "auto" and not on_device:
            device_entropy = False                       # the selection rule can end in zlib: for a host grid it does
        if device_entropy:
            from .entropy import deflate_grid
            if not on_device:
                raise TypeError("device_entropy needs the grid on the device (a CUDA tensor)")
            stream = deflate_grid(buf.reshape(-1, self.grid.width))
            keep = True
            if device_entropy == "auto":
                raw = buf.cpu().numpy().tobytes()
                keep = not lz77_would_win(raw, len(stream))
            if keep:
                w.write(stream)
                return "device"
        if raw is None:
            if type(buf).__module__.startswith("torch"):
                buf = buf.cpu().numpy()
            raw = np.ascontiguousarray(buf, dtype=np.uint8).tobytes()
        body = struct.pack("<Q", len(raw)) + raw + struct.pack("<Q", self.grid.width)
        enc = zlib.compressobj(9, zlib.DEFLATED, -15)          # raw DEFLATE, Compression::best()
        w.write(enc.compress(body) + enc.flush())
        return "zlib"

    @classmethod
    def deserialize_from_reader(cls, r):
        """src/archive.rs:43-55."""
        head = r.read(4)
        if len(head) != 4 or struct.unpack("<I", head)[0] != MAGIC:
            raise ValueError("incorrect magic number")               # :48-50
        meta = r.read(24)
        if len(meta) != 24:
            raise ValueError("truncated archive")
        q, i, width, height, scale = struct.unpack("<IIIIQ", meta)
        stream = r.read()
        # The header is untrusted (same acceptance as include/hgi_archive.hpp's deserialize(), deliberately stricter than the
        # reference's reader, which trusts both -- INTEGRATION.md): never inflate more than the metadata announces, and refuse
        # a size the stream cannot hold (DEFLATE expands at most 1032 : 1).  Inflated in bounded pieces.
        want = width * height + 16
        if want > len(stream) * 1032 + 64:
            raise ValueError("grid size in the metadata exceeds what the stream can hold")
        z = zlib.decompressobj(-15)
        parts, got, data_in = [], 0, stream
        while got <= want and not z.eof:
            piece = z.decompress(data_in, min(1 << 26, want + 1 - got))
            data_in = z.unconsumed_tail
            if not piece and not data_in:
                break                                    # no input left and the stream has not ended
            parts.append(piece)
            got += len(piece)
        if not z.eof or got != want:
            raise ValueError("corrupt grid stream")
        body = b"".join(parts)
        (n,) = struct.unpack_from("<Q", body, 0)
        if n != width * height:
            raise ValueError("grid size does not match the metadata")
        data = np.frombuffer(body, np.uint8, n, 8).copy()
        (gwidth,) = struct.unpack_from("<Q", body, 8 + n)
        if gwidth != width:
            raise ValueError("grid width does not match the metadata")
        return cls(Metadata(q, i, width, height, scale), Grid(data, gwidth))
