// C++ port of the reference's unit tests (src/lib.rs:25-97) on the hgi.hpp mirror.
// The reference's lossy tests shadow `image` with the decoded image (src/lib.rs:61) and therefore
// compare the decoded image with itself; here the comparison is the one the test meant:
// |original - decoded| <= quantizator.error().  Golden rows are SURVEY.md Appendix B.1.
#include <cstdio>
#include <cstdlib>

#include "hgi.hpp"

using hgi::GrayImage;
using hgi::interpolator::Crossed;
using hgi::quantizator::Linear;
using hgi::quantizator::QuantizationLevel;

static int failures = 0;
#define EXPECT(cond)                                                        \
    do {                                                                    \
        if (!(cond)) {                                                      \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);     \
            ++failures;                                                     \
        }                                                                   \
    } while (0)

static GrayImage get_test_image(uint32_t width, uint32_t height)   // src/lib.rs:35-43
{
    GrayImage image(width, height);
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < width; ++x) image(x, y) = static_cast<uint8_t>(x * y);
    return image;
}

static hgi::Grid test_error(QuantizationLevel quantization_level)   // src/lib.rs:45-77
{
    const size_t levels = 3;
    const uint32_t width = 12, height = 8;
    GrayImage image = get_test_image(width, height);

    Linear quantizator = Linear::from(quantization_level);
    const int max_error = quantizator.error();
    Crossed interpolator;
    hgi::Encoder<Crossed, Linear> encoder(interpolator, quantizator, levels);
    hgi::Grid grid = encoder.encode(image);

    hgi::Decoder<Crossed> decoder(Crossed{});
    GrayImage decoded = decoder.decode({width, height}, levels, grid);

    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < width; ++x) {
            int diff = std::abs(int(image(x, y)) - int(decoded(x, y)));
            EXPECT(diff <= max_error);
        }
    return grid;
}

int main()
{
    try {
        hgi::Grid lossless = test_error(QuantizationLevel::Lossless);   // lossless_compression
        test_error(QuantizationLevel::Low);                            // low_compression
        hgi::Grid medium = test_error(QuantizationLevel::Medium);      // medium_compression
        test_error(QuantizationLevel::High);                           // high_compression

        const uint8_t row0_lossless[12] = {0, 255, 252, 253, 0, 251, 244, 249, 0, 247, 248, 251};
        const uint8_t row7_lossless[12] = {253, 4, 5, 12, 13, 20, 21, 28, 29, 36, 55, 62};
        const uint8_t row0_medium[12] = {0, 0, 0, 0, 0, 254, 246, 251, 0, 251, 246, 254};
        for (uint32_t x = 0; x < 12; ++x) {
            EXPECT(lossless.get(x, 0) == row0_lossless[x]);
            EXPECT(lossless.get(x, 7) == row7_lossless[x]);
            EXPECT(medium.get(x, 0) == row0_medium[x]);
        }

        // the grid half of src/lib.rs:99-125 (`serde`): 8x8, levels 3, Lossless; round trip is exact
        GrayImage image = get_test_image(8, 8);
        hgi::Encoder<Crossed, Linear> encoder(Crossed{}, Linear::from(QuantizationLevel::Lossless), 3);
        hgi::Grid grid = encoder.encode(image);
        hgi::Decoder<Crossed> decoder(Crossed{});
        EXPECT(decoder.decode({8, 8}, 3, grid) == image);
        EXPECT(grid == encoder.encode(image));

        // an interpolator/levels the device cannot serve must surface as an error, not a host fallback
        bool threw = false;
        try {
            hgi::Decoder<Crossed>(Crossed{}).decode({8, 8}, 40, grid);
        } catch (const hgi::Error &e) {
            threw = e.status == HGI_EINVAL;
        }
        EXPECT(threw);
    } catch (const hgi::Error &e) {
        std::printf("FAIL: %s\n", e.what());
        return 2;
    }
    std::printf(failures ? "FAILED (%d)\n" : "ok: lossless/low/medium/high compression + grid round trip\n", failures);
    return failures ? 1 : 0;
}
