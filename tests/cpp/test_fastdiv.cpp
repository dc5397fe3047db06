// rustyhgi_amd/csrc/hgi_fastdiv.h against the machine's division: every divisor up to 70 000 and a spread of larger ones
// (tile counts: tiles per frame, per band, rows per band), numerators at the edges and at random.
#include <cstdint>
#include <cstdio>

#include "../../rustyhgi_amd/csrc/hgi_fastdiv.h"

int main()
{
    uint64_t x = 0x9E3779B97F4A7C15ull, bad = 0, checked = 0;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 16); };
    auto check = [&](uint32_t d) {
        const hgi::FastDiv f = hgi::make_fastdiv(d);
        const uint32_t edge[] = {0u, 1u, d - 1, d, d + 1, 2 * d - 1, 2 * d, 0x7FFFFFFFu, 0x80000000u, 0xFFFFFFFEu, 0xFFFFFFFFu};
        for (uint32_t n : edge) bad += hgi::fdiv(n, f) != n / d, ++checked;
        for (int i = 0; i < 64; ++i) {
            const uint32_t n = rnd() >> (rnd() & 31);
            bad += hgi::fdiv(n, f) != n / d, ++checked;
        }
    };
    for (uint32_t d = 1; d <= 70000; ++d) check(d);
    for (int i = 0; i < 200000; ++i) check(rnd() | 1u);
    for (int l = 1; l < 32; ++l) { check(1u << l); check((1u << l) - 1); check((1u << l) + 1); }
    check(0xFFFFFFFFu);
    std::printf("fastdiv: %llu checks, %llu wrong\n", (unsigned long long)checked, (unsigned long long)bad);
    return bad ? 1 : 0;
}
