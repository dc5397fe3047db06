// hgi.hpp -- header-only C++ mirror of the `hgi` crate's codec surface over the C ABI (hgi.h).
//
// The reference is compiled Rust with generic structs and static dispatch:
//     hgi::{Encoder, Decoder}                      src/lib.rs:21-23
//     hgi::interpolator::{Crossed, LeftTop, ...}   src/interpolator.rs
//     hgi::quantizator::{Linear, NoOp, ...}        src/quantizator.rs
// This header keeps those names, argument meanings and ownership rules so that a port of
// benches/bench.rs or of the tests in src/lib.rs reads line for line like the original:
//
//     auto quantizator = hgi::quantizator::Linear::from(hgi::quantizator::QuantizationLevel::Medium);
//     hgi::Encoder<hgi::interpolator::Crossed, hgi::quantizator::Linear> encoder({}, quantizator, levels);
//     hgi::Grid grid = encoder.encode(image);                 // src/encoder.rs:39
//     hgi::Decoder<hgi::interpolator::Crossed> decoder({});
//     hgi::GrayImage out = decoder.decode({width, height}, levels, grid);   // src/decoder.rs:18
//
// Per-pixel trait methods cannot cross to a GPU: any Quantizator is tabulated
// into its 256-entry table on the host, and the zero-sized interpolator types select a device
// predictor through `kernel_id`.  Everything computes in libhgi_hip.so; there is no host fallback --
// a missing device surfaces as hgi::Error.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "hgi.h"

namespace hgi {

struct Error : std::runtime_error {
    hgi_status status;
    Error(hgi_status st, const std::string &what) : std::runtime_error(what), status(st) {}
};

inline void check(hgi_status st)
{
    if (st != HGI_OK) throw Error(st, std::string("hgi: ") + hgi_last_error());
}

// `GrayImage = ImageBuffer<Luma<u8>, Vec<u8>>`: tightly packed rows, stride == width.
struct GrayImage {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> data;
    GrayImage() = default;
    GrayImage(uint32_t w, uint32_t h) : width(w), height(h), data(size_t(w) * h, 0) {}   // GrayImage::new zeroes
    std::pair<uint32_t, uint32_t> dimensions() const { return {width, height}; }
    uint8_t &operator()(uint32_t x, uint32_t y) { return data[size_t(y) * width + x]; }
    uint8_t operator()(uint32_t x, uint32_t y) const { return data[size_t(y) * width + x]; }
    bool operator==(const GrayImage &o) const { return width == o.width && height == o.height && data == o.data; }
};

// src/grid.rs:2-27
struct Grid {
    std::vector<uint8_t> buffer;
    size_t width = 0;
    Grid() = default;
    Grid(size_t w, size_t h) : buffer(w * h), width(w) {}
    void set(std::pair<uint32_t, uint32_t> at, uint8_t value) { buffer[size_t(at.second) * width + at.first] = value; }
    uint8_t get(uint32_t column, uint32_t line) const { return buffer[size_t(line) * width + column]; }
    bool operator==(const Grid &o) const { return width == o.width && buffer == o.buffer; }
};

namespace quantizator {

enum class QuantizationLevel { Lossless = 0, Low = 1, Medium = 2, High = 3 };   // src/quantizator.rs:3-8

// `trait Quantizator: From<QuantizationLevel>` (src/quantizator.rs:12-15) as a C++ concept-by-convention:
//   static Q from(QuantizationLevel);  uint8_t quantize(uint8_t) const;  uint8_t error() const;
struct NoOp {                                                   // src/quantizator.rs:17-34
    static NoOp from(QuantizationLevel) { return {}; }
    uint8_t quantize(uint8_t value) const { return value; }
    uint8_t error() const { return 0; }
};

struct Linear {                                                 // src/quantizator.rs:36-74
    std::array<uint8_t, 256> table{};
    uint8_t max_error = 0;
    static Linear from(QuantizationLevel level)
    {
        Linear q;
        check(hgi_linear_lut(static_cast<int>(level), q.table.data(), &q.max_error));
        return q;
    }
    uint8_t quantize(uint8_t value) const { return table[value]; }
    uint8_t error() const { return max_error; }
};

template <class Q>
std::array<uint8_t, 256> tabulate(const Q &q)                  // any Quantizator crosses as its table
{
    std::array<uint8_t, 256> t{};
    for (unsigned i = 0; i < 256; ++i) t[i] = q.quantize(static_cast<uint8_t>(i));
    return t;
}

}  // namespace quantizator

namespace interpolator {

enum class InterpolationType { Crossed = 0, Line = 1, Previous = 2 };   // src/interpolator.rs:4-9 (metadata tag)

struct LeftTop {                                                // src/interpolator.rs:15-28
    static constexpr hgi_interp kernel_id = HGI_INTERP_LEFTTOP;
};
struct Crossed {                                                // src/interpolator.rs:30-91
    static constexpr hgi_interp kernel_id = HGI_INTERP_CROSSED;
};

}  // namespace interpolator

// One hgi_ctx per Context; Encoder/Decoder borrow it (default: a process-wide ctx on device 0).
class Context {
  public:
    explicit Context(int device = 0) { check(hgi_ctx_create(device, &ctx_)); }
    ~Context() { hgi_ctx_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    hgi_ctx *get() const { return ctx_; }
    static Context &global()
    {
        static Context c(0);
        return c;
    }

  private:
    hgi_ctx *ctx_ = nullptr;
};

template <class I, class Q>
class Encoder {
  public:
    // `Encoder::new(interpolator, quantizator, scale_level)` -- src/encoder.rs:18
    Encoder(I interpolator, Q quantizator, size_t scale_level, Context &ctx = Context::global())
        : interpolator_(interpolator), quantizator_(std::move(quantizator)), scale_level_(scale_level), ctx_(ctx),
          table_(quantizator::tabulate(quantizator_))
    {
    }

    // `encode(&mut self, input: GrayImage) -> Grid` -- src/encoder.rs:39 (infallible there; here a
    // device failure throws hgi::Error).  The image is taken by value exactly as in the reference.
    Grid encode(GrayImage input)
    {
        Grid grid(input.width, input.height);
        check(hgi_encode_u8(ctx_.get(), input.data.data(), input.width, input.height,
                            static_cast<uint32_t>(scale_level_), I::kernel_id, table_.data(), grid.buffer.data()));
        return grid;
    }

  private:
    I interpolator_;
    Q quantizator_;
    size_t scale_level_;
    Context &ctx_;
    std::array<uint8_t, 256> table_;
};

template <class I>
class Decoder {
  public:
    // `Decoder::new(interpolator)` -- src/decoder.rs:14
    explicit Decoder(I interpolator, Context &ctx = Context::global()) : interpolator_(interpolator), ctx_(ctx) {}

    // `decode(&mut self, (width, height), levels, grid: &Grid) -> GrayImage` -- src/decoder.rs:18
    GrayImage decode(std::pair<uint32_t, uint32_t> dimensions, size_t levels, const Grid &grid)
    {
        GrayImage image(dimensions.first, dimensions.second);
        check(hgi_decode_u8(ctx_.get(), grid.buffer.data(), dimensions.first, dimensions.second,
                            static_cast<uint32_t>(levels), I::kernel_id, image.data.data()));
        return image;
    }

  private:
    I interpolator_;
    Context &ctx_;
};

}  // namespace hgi
