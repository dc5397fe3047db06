// hgi -- C++ counterpart of the reference CLI (src/main.rs:41-134, src/options.rs:13-65) on the
// MI355X library.  Same subcommands, flags and defaults:
//
//   hgi encode -i <input> -o <output> [-l <level>=4] [-q lossless|low|medium|high = medium]
//   hgi decode -i <input.hgi> -o <output>
//   hgi test <input> [-s <suffix>=""] [-l <level>=4] [-q <quantizator>=medium]
//   (encode / test also take --entropy zlib|device|auto, default zlib: `device` has the GPU write the archive's DEFLATE stream
//    as Huffman-coded literals -- same container, readable by the same readers; no counterpart in the reference)
//
// `hgi test` prints the reference's four report lines (src/main.rs:108-111, integer MSE division at
// :106) and writes "<stem><suffix>.pgm" and "<stem><suffix>.hgi" into the working directory.
// Differences, all forced by this image (no png/jpeg/tiff development headers):
// images are read from binary PGM (P5) or uncompressed 8-bit grayscale TIFF (which covers the
// reference's res/LENA.TIF) and written as PGM; and a failure exits with status 1 (the reference
// prints the error and exits 0, src/main.rs:130-134).
// The archive is the reference's wire format (src/archive.rs:13-56); zlib supplies raw
// DEFLATE at level 9 where the reference uses flate2's Compression::best().
//
//   build: g++ -O2 -std=c++17 -Iinclude cli/hgi_cli.cpp -Lrustyhgi_amd -lhgi_hip -lz -o hgi
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <sstream>
#include <string>

#include "hgi.hpp"
#include "hgi_archive.hpp"

using hgi::GrayImage;
using hgi::Grid;
using hgi::interpolator::Crossed;
using hgi::interpolator::InterpolationType;
using hgi::quantizator::Linear;
using hgi::quantizator::QuantizationLevel;

namespace {

using Failure = hgi::ArchiveError;   // one error type for I/O, options and the archive
using hgi::deserialize;
using hgi::Metadata;
using hgi::serialize;

std::vector<uint8_t> read_file(const std::string &path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw Failure("cannot open " + path);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

void write_file(const std::string &path, const std::vector<uint8_t> &bytes)
{
    std::ofstream f(path, std::ios::binary);
    if (!f || !f.write(reinterpret_cast<const char *>(bytes.data()), (std::streamsize)bytes.size()))
        throw Failure("cannot write " + path);
}

// ---- images -------------------------------------------------------------------------------------
GrayImage read_pgm(const std::vector<uint8_t> &b)
{
    size_t pos = 2;
    auto token = [&]() {
        for (;;) {
            while (pos < b.size() && std::isspace(b[pos])) ++pos;
            if (pos < b.size() && b[pos] == '#')
                while (pos < b.size() && b[pos] != '\n') ++pos;
            else
                break;
        }
        size_t s = pos;
        while (pos < b.size() && !std::isspace(b[pos])) ++pos;
        return std::string(b.begin() + s, b.begin() + pos);
    };
    long w = std::stol(token()), h = std::stol(token()), maxv = std::stol(token());
    ++pos;   // the single whitespace after maxval
    if (w <= 0 || h <= 0 || maxv != 255 || pos + size_t(w) * h > b.size()) throw Failure("unsupported PGM (need binary P5, maxval 255)");
    GrayImage img((uint32_t)w, (uint32_t)h);
    std::memcpy(img.data.data(), b.data() + pos, size_t(w) * h);
    return img;
}

GrayImage read_tiff(const std::vector<uint8_t> &b)   // baseline, uncompressed, 8-bit, one sample per pixel
{
    const bool le = b[0] == 'I';
    auto u16 = [&](size_t o) { return le ? uint32_t(b[o] | b[o + 1] << 8) : uint32_t(b[o] << 8 | b[o + 1]); };
    auto u32 = [&](size_t o) { return le ? u16(o) | u16(o + 2) << 16 : u16(o) << 16 | u16(o + 2); };
    if (b.size() < 8 || u16(2) != 42) throw Failure("not a TIFF file");
    size_t ifd = u32(4);
    uint32_t n = u16(ifd), width = 0, height = 0, bits = 1, comp = 1, photo = 1, spp = 1, rows = 0xFFFFFFFFu;
    std::vector<uint32_t> offs, counts;
    for (uint32_t i = 0; i < n; ++i) {
        size_t e = ifd + 2 + 12 * i;
        uint32_t tag = u16(e), type = u16(e + 2), cnt = u32(e + 4);
        size_t esz = type == 3 ? 2 : 4, vo = cnt * esz <= 4 ? e + 8 : u32(e + 8);
        auto val = [&](uint32_t k) { return type == 3 ? u16(vo + 2 * k) : u32(vo + 4 * k); };
        switch (tag) {
        case 256: width = val(0); break;
        case 257: height = val(0); break;
        case 258: bits = val(0); break;
        case 259: comp = val(0); break;
        case 262: photo = val(0); break;
        case 277: spp = val(0); break;
        case 278: rows = val(0); break;
        case 273: for (uint32_t k = 0; k < cnt; ++k) offs.push_back(val(k)); break;
        case 279: for (uint32_t k = 0; k < cnt; ++k) counts.push_back(val(k)); break;
        default: break;
        }
    }
    if (bits != 8 || comp != 1 || spp != 1 || photo > 1 || !width || !height || offs.empty())
        throw Failure("unsupported TIFF (need uncompressed 8-bit grayscale)");
    GrayImage img(width, height);
    size_t done = 0, total = size_t(width) * height;
    for (size_t s = 0; s < offs.size() && done < total; ++s) {
        size_t want = std::min<size_t>(total - done, s < counts.size() ? counts[s] : size_t(rows) * width);
        if (offs[s] + want > b.size()) throw Failure("truncated TIFF strip");
        std::memcpy(img.data.data() + done, b.data() + offs[s], want);
        done += want;
    }
    if (done != total) throw Failure("TIFF strips do not cover the image");
    if (photo == 0)
        for (auto &v : img.data) v = 255 - v;   // WhiteIsZero
    return img;
}

GrayImage open_image(const std::string &path)   // image::open(path)?.to_luma(), src/main.rs:42,74
{
    std::vector<uint8_t> b = read_file(path);
    if (b.size() > 2 && b[0] == 'P' && b[1] == '5') return read_pgm(b);
    if (b.size() > 4 && ((b[0] == 'I' && b[1] == 'I') || (b[0] == 'M' && b[1] == 'M'))) return read_tiff(b);
    throw Failure("unsupported image format (binary PGM or uncompressed 8-bit TIFF): " + path);
}

void save_pgm(const GrayImage &img, const std::string &path)
{
    std::string head = "P5\n" + std::to_string(img.width) + " " + std::to_string(img.height) + "\n255\n";
    std::vector<uint8_t> out(head.begin(), head.end());
    out.insert(out.end(), img.data.begin(), img.data.end());
    write_file(path, out);
}

// ---- options (src/options.rs) ------------------------------------------------------------------------
QuantizationLevel parse_level(std::string v)   // case-insensitive, not typo tolerant (src/options.rs:61)
{
    std::transform(v.begin(), v.end(), v.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    if (v == "lossless") return QuantizationLevel::Lossless;
    if (v == "low") return QuantizationLevel::Low;
    if (v == "medium") return QuantizationLevel::Medium;
    if (v == "high") return QuantizationLevel::High;
    throw Failure("'" + v + "' isn't a valid value for '--quantizator <quantization_level>' [values: Lossless, Low, Medium, High]");
}

struct Opts {
    std::string cmd, input, output, suffix;
    size_t level = 4;                                            // src/options.rs:54
    QuantizationLevel quant = QuantizationLevel::Medium;         // src/options.rs:62
    int entropy = 0;   // --entropy zlib (0, the reference's writer) | device (1: the GPU's entropy stage) | auto (2: device unless an LZ77 probe says zlib wins); no reference flag
};

std::vector<uint8_t> write_archive(const Opts &o, const Metadata &metadata, const Grid &grid)
{
    if (o.entropy == 1) return hgi::serialize_device(metadata, grid, hgi::Context::global().get());
    if (o.entropy == 2) return hgi::serialize_auto(metadata, grid, hgi::Context::global().get());
    return serialize(metadata, grid);
}

Opts parse(int argc, char **argv)
{
    if (argc < 2) throw Failure("usage: hgi <encode|decode|test> ...");
    Opts o;
    o.cmd = argv[1];
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() {
            if (i + 1 >= argc) throw Failure("missing value for " + a);
            return std::string(argv[++i]);
        };
        if (a == "-i" || a == "--input") o.input = next();
        else if (a == "-o" || a == "--output") o.output = next();
        else if (a == "-l" || a == "--level") o.level = std::stoul(next());
        else if (a == "-q" || a == "--quantizator") o.quant = parse_level(next());
        else if (a == "-s" || a == "--suffix") o.suffix = next();
        else if (a == "--entropy") {
            const std::string v = next();
            if (v != "device" && v != "zlib" && v != "auto") throw Failure("'" + v + "' isn't a valid value for '--entropy' [values: zlib, device, auto]");
            o.entropy = v == "device" ? 1 : v == "auto" ? 2 : 0;
        }
        else if (o.cmd == "test" && o.input.empty() && a[0] != '-') o.input = a;   // positional <input>
        else throw Failure("unexpected argument '" + a + "'");
    }
    return o;
}

std::string file_stem(const std::string &path)
{
    size_t slash = path.find_last_of('/');
    std::string name = slash == std::string::npos ? path : path.substr(slash + 1);
    size_t dot = name.find_last_of('.');
    return dot == std::string::npos || dot == 0 ? name : name.substr(0, dot);
}

// ---- subcommands -------------------------------------------------------------------------------------
void encode(const Opts &o)   // src/main.rs:41-61
{
    if (o.input.empty() || o.output.empty()) throw Failure("encode needs -i <input> -o <output>");
    GrayImage image = open_image(o.input);
    Linear quantizator = Linear::from(o.quant);
    hgi::Encoder<Crossed, Linear> encoder(Crossed{}, quantizator, o.level);
    const uint32_t width = image.width, height = image.height;
    Grid grid = encoder.encode(std::move(image));
    Metadata metadata{o.quant, InterpolationType::Crossed, width, height, o.level};
    write_file(o.output, write_archive(o, metadata, grid));
}

void decode(const Opts &o)   // src/main.rs:63-71 (always Crossed; metadata.interpolation is ignored there too)
{
    if (o.input.empty() || o.output.empty()) throw Failure("decode needs -i <input> -o <output>");
    Metadata m;
    Grid grid;
    deserialize(read_file(o.input), m, grid);
    hgi::Decoder<Crossed> decoder(Crossed{});
    save_pgm(decoder.decode({m.width, m.height}, m.scale_level, grid), o.output);
}

void test(const Opts &o)   // src/main.rs:73-120
{
    if (o.input.empty()) throw Failure("test needs <input>");
    GrayImage image_before = open_image(o.input);
    Linear quantizator = Linear::from(o.quant);
    hgi::Encoder<Crossed, Linear> encoder(Crossed{}, quantizator, o.level);
    Grid grid = encoder.encode(image_before);   // a copy, as `image_before.clone()` at :79
    hgi::Decoder<Crossed> decoder(Crossed{});
    GrayImage image_after = decoder.decode(image_before.dimensions(), o.level, grid);

    size_t sd = 0;   // :84-92
    for (size_t i = 0; i < image_before.data.size(); ++i) {
        size_t diff = (size_t)std::abs(int(image_before.data[i]) - int(image_after.data[i]));
        sd += diff * diff;
    }
    Metadata metadata{o.quant, InterpolationType::Crossed, image_before.width, image_before.height, o.level};
    std::vector<uint8_t> buffer = write_archive(o, metadata, grid);

    const uint32_t uncompressed = image_before.height * image_before.width;
    sd /= uncompressed;   // :106 integer division
    const size_t compressed = buffer.size();
    std::printf("Uncompressed: %u kb\n", uncompressed / 1024);              // :108
    std::printf("Compressed:   %zu kb\n", compressed / 1024);               // :109
    std::printf("Ratio:        %.2f\n", double(uncompressed) / double(compressed));   // :110
    std::printf("SD:           %.2f\n", std::sqrt(double(sd)));             // :111

    const std::string filename = file_stem(o.input) + o.suffix;   // :113
    save_pgm(image_after, filename + ".pgm");
    write_file(filename + ".hgi", buffer);
}

}  // namespace

int main(int argc, char **argv)
{
    try {
        Opts o = parse(argc, argv);
        if (o.cmd == "encode") encode(o);
        else if (o.cmd == "decode") decode(o);
        else if (o.cmd == "test") test(o);
        else throw Failure("unknown subcommand '" + o.cmd + "' (encode, decode, test)");
    } catch (const std::exception &e) {
        std::fprintf(stderr, "An error occured: %s\n", e.what());   // src/main.rs:132 (sic)
        return 1;
    }
    return 0;
}
