// The C++ .hgi reader (include/hgi_archive.hpp) on hostile input, built with g++ -fsanitize=address,undefined by
// tests/test_sanitizers.py.  No GPU, no libhgi_hip: only the zlib container code runs.
// Reference: src/archive.rs:43-55 (`deserialize_from_reader`; its errors are `Box<Error>`s, ours ArchiveError).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/hgi_archive.hpp"

using namespace hgi;

static int failures = 0;
#define EXPECT(cond, what)                          \
    do {                                            \
        if (!(cond)) {                              \
            fprintf(stderr, "FAILED: %s\n", what);  \
            ++failures;                             \
        }                                           \
    } while (0)

template <typename F>
static bool throws_archive_error(F f)
{
    try {
        f();
    } catch (const ArchiveError &) {
        return true;
    } catch (const std::exception &e) {
        fprintf(stderr, "  (threw %s instead of ArchiveError)\n", e.what());
        return false;
    }
    return false;
}

int main()
{
    Metadata m{quantizator::QuantizationLevel::Medium, interpolator::InterpolationType::Crossed, 37, 23, 4};
    Grid g;
    g.width = 37;
    g.buffer.resize(37 * 23);
    uint64_t x = 88172645463325252ull;
    for (auto &b : g.buffer) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        b = (x & 7) ? 0 : uint8_t(x >> 32);
    }
    const std::vector<uint8_t> good = serialize(m, g);
    Metadata m2{};
    Grid g2;
    deserialize(good, m2, g2);
    EXPECT(g2.buffer == g.buffer && g2.width == 37 && m2.width == 37 && m2.height == 23 && m2.scale_level == 4, "round trip");

    // 1. a 28-byte header that announces 2^32-1 x 2^32-1 pixels: refused before anything is allocated
    {
        std::vector<uint8_t> bad(good.begin(), good.begin() + 28);
        for (int i = 12; i < 20; ++i) bad[i] = 0xFF;
        EXPECT(throws_archive_error([&] { Metadata a; Grid b; deserialize(bad, a, b); }), "exabyte header with no stream");
        bad.insert(bad.end(), good.begin() + 28, good.end());
        EXPECT(throws_archive_error([&] { Metadata a; Grid b; deserialize(bad, a, b); }), "exabyte header with a small stream");
    }
    // 2. truncations at every length
    for (size_t n = 0; n < good.size(); ++n) {
        std::vector<uint8_t> cut(good.begin(), good.begin() + n);
        EXPECT(throws_archive_error([&] { Metadata a; Grid b; deserialize(cut, a, b); }), "truncated archive accepted");
    }
    // 3. the grid's own width disagrees with the metadata: same pixel count, other shape
    {
        Grid other = g;
        other.width = 23;
        const std::vector<uint8_t> odd = serialize(m, other);
        EXPECT(throws_archive_error([&] { Metadata a; Grid b; deserialize(odd, a, b); }), "grid.width != metadata.width accepted");
    }
    // 4. metadata smaller / larger than the stream's grid
    {
        std::vector<uint8_t> bad = good;
        bad[16] = 22;      // height 22
        EXPECT(throws_archive_error([&] { Metadata a; Grid b; deserialize(bad, a, b); }), "stream longer than the metadata");
        bad[16] = 24;
        EXPECT(throws_archive_error([&] { Metadata a; Grid b; deserialize(bad, a, b); }), "stream shorter than the metadata");
    }
    // 5. random byte flips: either a clean error or a grid of the announced size -- never a crash (ASan / UBSan watch)
    for (int it = 0; it < 3000; ++it) {
        std::vector<uint8_t> mut = good;
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const int flips = 1 + int(x % 3);
        for (int k = 0; k < flips; ++k) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            mut[x % mut.size()] ^= uint8_t(1u << ((x >> 40) & 7));
        }
        try {
            Metadata a;
            Grid b;
            deserialize(mut, a, b);
            EXPECT(b.buffer.size() == size_t(a.width) * a.height && b.width == a.width, "accepted archive with inconsistent sizes");
        } catch (const ArchiveError &) {
        }
    }
    if (failures) return 1;
    printf("archive hardening ok\n");
    return 0;
}
