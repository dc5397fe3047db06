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
    const size_t head = out.size(), bound = deflateBound(&z, (uLong)body.size());
    out.resize(head + bound);
    z.next_in = body.data();
    z.avail_in = (uInt)body.size();
    z.next_out = out.data() + head;
    z.avail_out = (uInt)bound;
    const int rc = deflate(&z, Z_FINISH);
    deflateEnd(&z);
    if (rc != Z_STREAM_END) throw ArchiveError("deflate failed");
    out.resize(head + z.total_out);
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
    std::vector<uint8_t> body(size_t(m.width) * m.height + 16);
    z_stream z{};
    if (inflateInit2(&z, -15) != Z_OK) throw ArchiveError("inflateInit2 failed");
    z.next_in = const_cast<uint8_t *>(b.data()) + 28;
    z.avail_in = (uInt)(b.size() - 28);
    z.next_out = body.data();
    z.avail_out = (uInt)body.size();
    const int rc = inflate(&z, Z_FINISH);
    inflateEnd(&z);
    if (rc != Z_STREAM_END || z.total_out != body.size()) throw ArchiveError("corrupt grid stream");
    const uint64_t n = get_le(body, 0, 8);
    if (n != size_t(m.width) * m.height) throw ArchiveError("grid size does not match the metadata");
    grid.buffer.assign(body.begin() + 8, body.begin() + 8 + n);
    grid.width = get_le(body, 8 + n, 8);
}

}  // namespace hgi
