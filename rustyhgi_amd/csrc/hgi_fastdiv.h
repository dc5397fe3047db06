// Exact unsigned 32-bit division by a launch-wide constant in five scalar operations (Granlund & Montgomery, "Division by
// invariant integers using multiplication"): the host derives the multiplier once per launch, the kernels never divide.
// Plain C++ (tests/cpp/test_fastdiv.cpp checks it with g++); the tile kernels use it for their block -> tile index math,
// which used to hold eight 32-bit divisions per tile -- ~ 200 instructions of a 1 200-instruction encode tile.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define HGI_HD __host__ __device__
#else
#define HGI_HD
#endif

namespace hgi {

struct FastDiv {
    uint32_t m, s1, s2, d;
};

// n / d == (t + ((n - t) >> s1)) >> s2 with t = mulhi(m, n), for every 32-bit n; d >= 1 (d == 0 is never divided by:
// an unused slot gets the identity)
inline FastDiv make_fastdiv(uint32_t d)
{
    FastDiv f = {1u, 0u, 0u, d};
    if (d <= 1) return f;
    uint32_t l = 0;
    while (((uint64_t)1 << l) < d) ++l;
    f.m = (uint32_t)(((((uint64_t)1 << l) - d) << 32) / d + 1);
    f.s1 = 1;
    f.s2 = l - 1;
    return f;
}

HGI_HD inline uint32_t fdiv(uint32_t n, const FastDiv &f)
{
    const uint32_t t = (uint32_t)(((uint64_t)f.m * n) >> 32);
    return (t + ((n - t) >> f.s1)) >> f.s2;
}

}  // namespace hgi
