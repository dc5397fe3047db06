// .hgi archive: the reference's wire format (src/archive.rs:13-56 of pl0q1n/RustyHGI) for the C++
// hosts -- the `hgi` CLI (cli/hgi_cli.cpp) and the criterion-harness port (benches/bench.cpp).
//
//   55 A5 AD BA                      u32 LE magic 0xBAADA555                                   (src/archive.rs:13, :32)
//   u32 quantization_level (0..3)    bincode 1.x defaults: little-endian, fixed width,          (:15-22, :33)
//   u32 interpolation (0..2)         enum variant index as u32, usize as u64
//   u32 width, u32 height, u64 scale_level
//   raw DEFLATE of { u64 N, N grid bytes, u64 grid.width }                                       (:34-40)
//
// zlib supplies raw DEFLATE (window bits -15) at level 9 where the reference uses flate2's Compression::best(); any
// valid DEFLATE stream decodes on either side, the compressed bytes themselves may differ from miniz's.
// Link with -lz.
#pragma once
#include <zlib.h>

#include <algorithm>

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "hgi.hpp"

namespace hgi {

struct ArchiveError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

constexpr uint32_t kArchiveMagic = 0xBAADA555u;

struct Metadata {   // src/archive.rs:15-22
    quantizator::QuantizationLevel quantization_level;
    interpolator::InterpolationType interpolation;
    uint32_t width, height;
    uint64_t scale_level;
};

// `Archive<G>` (src/archive.rs:24-28)
struct Archive {
    Metadata metadata;
    Grid grid;
};

namespace archive_detail {
inline void put_le(std::vector<uint8_t> &o, uint64_t v, int bytes)
{
    for (int i = 0; i < bytes; ++i) o.push_back(uint8_t(v >> (8 * i)));
}
inline uint64_t get_le(const std::vector<uint8_t> &b, size_t at, int bytes)
{
    if (at + bytes > b.size()) throw ArchiveError("truncated archive");
    uint64_t v = 0;
    for (int i = 0; i < bytes; ++i) v |= uint64_t(b[at + i]) << (8 * i);
    return v;
}
}  // namespace archive_detail

// `bincode::serialized_size(&archive)` (benches/bench.rs:119): metadata 24 B + grid (8 + N + 8) B, uncompressed
inline size_t serialized_size(const Metadata &, const Grid &grid) { return 24 + 16 + grid.buffer.size(); }

// `Archive::serialize_to_writer` (src/archive.rs:31-41), appending to `out` (the reference's `W: Write`)
inline void serialize_into(std::vector<uint8_t> &out, const Metadata &m, const Grid &grid)
{
    using archive_detail::put_le;
    put_le(out, kArchiveMagic, 4);
    put_le(out, (uint32_t)m.quantization_level, 4);
    put_le(out, (uint32_t)m.interpolation, 4);
    put_le(out, m.width, 4);
    put_le(out, m.height, 4);
    put_le(out, m.scale_level, 8);
    std::vector<uint8_t> body;   // bincode(Grid): Vec<u8> = u64 length + bytes, then usize width as u64
    body.reserve(grid.buffer.size() + 16);
    put_le(body, grid.buffer.size(), 8);
    body.insert(body.end(), grid.buffer.begin(), grid.buffer.end());
    put_le(body, grid.width, 8);
    z_stream z{};
    if (deflateInit2(&z, 9, Z_DEFLATED, -15, 9, Z_DEFAULT_STRATEGY) != Z_OK) throw ArchiveError("deflateInit2 failed");
    // zlib counts in 32-bit `uInt`s: feed and drain in pieces of at most 1 GiB so that grids of 4 GiB and more stream through
    constexpr size_t kPiece = size_t(1) << 30;
    const size_t head = out.size();
    size_t in_at = 0, out_at = head;
    int rc = Z_OK;
    while (rc != Z_STREAM_END) {
        if (z.avail_in == 0 && in_at < body.size()) {
            const size_t take = std::min(kPiece, body.size() - in_at);
            z.next_in = body.data() + in_at;
            z.avail_in = (uInt)take;
            in_at += take;
        }
        if (out.size() - out_at < 65536) out.resize(out.size() + std::max<size_t>(65536, body.size() / 8));
        const size_t room = std::min(kPiece, out.size() - out_at);
        z.next_out = out.data() + out_at;
        z.avail_out = (uInt)room;
        rc = deflate(&z, in_at == body.size() ? Z_FINISH : Z_NO_FLUSH);
        out_at += room - z.avail_out;
        if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) break;
    }
    deflateEnd(&z);
    if (rc != Z_STREAM_END) throw ArchiveError("deflate failed");
    out.resize(out_at);
}

// The same container with the entropy stage on the device (hgi_deflate_grid_dev, include/hgi.h): the grid stays where
// the encoder left it (d_grid, m.width x m.height bytes of device memory), the DEFLATE stream is one dynamic-Huffman block
// of literals written by the GPU.  Readable by deserialize() below and by the reference's reader alike.
inline void serialize_device_into(std::vector<uint8_t> &out, const Metadata &m, hgi_ctx *ctx, const void *d_grid)
{
    using archive_detail::put_le;
    put_le(out, kArchiveMagic, 4);
    put_le(out, (uint32_t)m.quantization_level, 4);
    put_le(out, (uint32_t)m.interpolation, 4);
    put_le(out, m.width, 4);
    put_le(out, m.height, 4);
    put_le(out, m.scale_level, 8);
    const size_t n = size_t(m.width) * m.height, head = out.size(), cap = n + n / 8 + 1024;
    out.resize(head + cap);
    size_t bytes = 0;
    if (hgi_deflate_grid_dev(ctx, d_grid, m.width, m.height, out.data() + head, cap, &bytes) != HGI_OK)
        throw ArchiveError(std::string("device entropy stage: ") + hgi_last_error());
    out.resize(head + bytes);
}

// Selection rule between the two writers (`--entropy auto`).  The device stream codes literals and runs only; where a
// grid repeats itself at a distance (synthetic, periodic images: the criterion harness's `(x*y) as u8` frame is 19x
// smaller under LZ77) zlib must write the archive.  The rule costs one fast probe: the device stream is produced first
// (fractions of a millisecond; its size is exact), then up to 1 MiB from the middle of the grid goes through zlib at
// level 1 -- which finds long-distance repeats as surely as level 9, at memcpy-like speed on such data -- and if that
// ratio, applied to the whole grid, undercuts the device stream by more than a quarter, the grid is compressed the
// reference's way (level 9).  Returns true when the device stream was kept.  `grid` is the host copy of d_grid.
namespace archive_detail {
// true: an LZ77 probe of the grid predicts a stream more than a quarter smaller than the device's `device_stream_bytes`
inline bool lz77_would_win(const Grid &grid, size_t device_stream_bytes)
{
    const size_t n = grid.buffer.size(), probe = std::min<size_t>(n, size_t(1) << 20), at = (n - probe) / 2;
    if (probe < 4096) return false;
    uLongf got = compressBound((uLong)probe);
    std::vector<uint8_t> tmp(got);
    if (compress2(tmp.data(), &got, grid.buffer.data() + at, (uLong)probe, 1) != Z_OK) return false;
    return double(got) / double(probe) * double(n) <= 0.75 * double(device_stream_bytes);
}
}  // namespace archive_detail

inline bool serialize_auto_into(std::vector<uint8_t> &out, const Metadata &m, const Grid &grid, hgi_ctx *ctx, const void *d_grid)
{
    std::vector<uint8_t> dev;
    serialize_device_into(dev, m, ctx, d_grid);
    const bool keep = !archive_detail::lz77_would_win(grid, dev.size() - 28);
    if (keep)
        out.insert(out.end(), dev.begin(), dev.end());
    else
        serialize_into(out, m, grid);
    return keep;
}

// ... and with the grid in host memory (hgi_deflate_grid): what the CLI's `--entropy device` writes
inline std::vector<uint8_t> serialize_device(const Metadata &m, const Grid &grid, hgi_ctx *ctx)
{
    using archive_detail::put_le;
    std::vector<uint8_t> out;
    put_le(out, kArchiveMagic, 4);
    put_le(out, (uint32_t)m.quantization_level, 4);
    put_le(out, (uint32_t)m.interpolation, 4);
    put_le(out, m.width, 4);
    put_le(out, m.height, 4);
    put_le(out, m.scale_level, 8);
    const size_t n = grid.buffer.size(), head = out.size(), cap = n + n / 8 + 1024;
    const uint32_t height = grid.width ? (uint32_t)(n / grid.width) : 0;
    out.resize(head + cap);
    size_t bytes = 0;
    if (hgi_deflate_grid(ctx, grid.buffer.data(), (uint32_t)grid.width, height, out.data() + head, cap, &bytes) != HGI_OK)
        throw ArchiveError(std::string("device entropy stage: ") + hgi_last_error());
    out.resize(head + bytes);
    return out;
}

// `--entropy auto` with the grid in host memory (the CLI): the rule of serialize_auto_into(); *used_device says which
inline std::vector<uint8_t> serialize_auto(const Metadata &m, const Grid &grid, hgi_ctx *ctx, bool *used_device = nullptr)
{
    std::vector<uint8_t> dev = serialize_device(m, grid, ctx);
    const bool keep = !archive_detail::lz77_would_win(grid, dev.size() - 28);
    if (used_device) *used_device = keep;
    if (keep) return dev;
    std::vector<uint8_t> out;
    serialize_into(out, m, grid);
    return out;
}

inline std::vector<uint8_t> serialize(const Metadata &m, const Grid &grid)
{
    std::vector<uint8_t> out;
    serialize_into(out, m, grid);
    return out;
}

// `Archive::deserialize_from_reader` (src/archive.rs:43-55)
inline void deserialize(const std::vector<uint8_t> &b, Metadata &m, Grid &grid)
{
    using archive_detail::get_le;
    if (get_le(b, 0, 4) != kArchiveMagic) throw ArchiveError("incorrect magic number");   // :48-50
    m.quantization_level = (quantizator::QuantizationLevel)get_le(b, 4, 4);
    m.interpolation = (interpolator::InterpolationType)get_le(b, 8, 4);
    m.width = (uint32_t)get_le(b, 12, 4);
    m.height = (uint32_t)get_le(b, 16, 4);
    m.scale_level = get_le(b, 20, 8);
    if (b.size() < 28) throw ArchiveError("truncated archive");
    // The header is untrusted: never allocate width * height up front.  DEFLATE expands at most 1032 : 1, so the stream
    // that follows bounds what the body can be; the buffer grows as inflate produces output, in pieces of at most 1 GiB
    // per call (zlib counts in 32-bit uInts).
    const uint64_t want = uint64_t(m.width) * m.height + 16;
    const uint64_t most = uint64_t(b.size() - 28) * 1032 + 64;
    if (want > most) throw ArchiveError("grid size in the metadata exceeds what the stream can hold");
    constexpr size_t kPiece = size_t(1) << 30;
    std::vector<uint8_t> body;
    z_stream z{};
    if (inflateInit2(&z, -15) != Z_OK) throw ArchiveError("inflateInit2 failed");
    size_t in_at = 28, out_at = 0;
    int rc = Z_OK;
    while (rc != Z_STREAM_END) {
        if (z.avail_in == 0 && in_at < b.size()) {
            const size_t take = std::min(kPiece, b.size() - in_at);
            z.next_in = const_cast<uint8_t *>(b.data()) + in_at;
            z.avail_in = (uInt)take;
            in_at += take;
        }
        if (out_at == body.size()) {
            if (body.size() >= want) break;                      // more output than the metadata announces
            body.resize((size_t)std::min<uint64_t>(want, std::max<uint64_t>(uint64_t(body.size()) * 2, 1 << 16)));
        }
        const size_t room = std::min(kPiece, body.size() - out_at);
        z.next_out = body.data() + out_at;
        z.avail_out = (uInt)room;
        rc = inflate(&z, Z_NO_FLUSH);
        out_at += room - z.avail_out;
        if (rc != Z_OK && rc != Z_STREAM_END) break;             // Z_BUF_ERROR: no input left and the stream has not ended
    }
    inflateEnd(&z);
    if (rc != Z_STREAM_END || out_at != want) throw ArchiveError("corrupt grid stream");
    const uint64_t n = get_le(body, 0, 8);
    if (n != uint64_t(m.width) * m.height) throw ArchiveError("grid size does not match the metadata");
    grid.buffer.assign(body.begin() + 8, body.begin() + 8 + n);
    grid.width = get_le(body, 8 + n, 8);
    if (grid.width != m.width) throw ArchiveError("grid width does not match the metadata");
}

}  // namespace hgi
