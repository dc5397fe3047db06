// Fuzz of the entropy stage's host planner (rustyhgi_amd/csrc/hgi_huffman_host.h) as plain C++ under
// g++ -fsanitize=address,undefined: tests/test_sanitizers.py builds and runs it, then inflates the blocks it emits.
//   fuzz_huffman <cases> <seed> <out-file>
// For every random 286-bin histogram: code lengths <= 15, Kraft sum <= 1 (== 1 with two or more symbols in use), the
// canonical codes are prefix-free, and the header + an end-of-block code form a complete (empty-payload) DEFLATE block,
// written to <out-file> as [u32 nbytes][bytes] records for zlib to parse.  Also drives plan_frame() the way
// hgi_capi.hip does (five histograms per frame) and checks its bookkeeping, and the BitWriter's overflow report.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../rustyhgi_amd/csrc/hgi_huffman_host.h"

using namespace hgi;

static uint64_t rng_state;
static uint64_t rnd()
{
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

#define REQUIRE(cond, ...)                                   \
    do {                                                     \
        if (!(cond)) {                                       \
            fprintf(stderr, "case %d: ", c);                 \
            fprintf(stderr, __VA_ARGS__);                    \
            fprintf(stderr, "  [%s]\n", #cond);              \
            return 1;                                        \
        }                                                    \
    } while (0)

static void random_hist(uint64_t *h, int shape)
{
    for (int i = 0; i < kDeflateSymbols; ++i) h[i] = 0;
    switch (shape) {
    case 0:      // everything in use, flat-ish
        for (int i = 0; i < kDeflateSymbols; ++i) h[i] = 1 + rnd() % 1000;
        break;
    case 1:      // geometric around zero: what a residual grid looks like
        for (int i = 0; i < 256; ++i) {
            const int d = i < 128 ? i : 256 - i;
            h[i] = d < 40 ? (uint64_t)(1ull << (40 - d)) + rnd() % 7 : rnd() % 3;
        }
        for (int i = 257; i < kDeflateSymbols; ++i) h[i] = rnd() % 5000;
        break;
    case 2:      // a handful of symbols
        for (int k = 0, n = 1 + (int)(rnd() % 6); k < n; ++k) h[rnd() % kDeflateSymbols] = 1 + rnd() % 100000;
        break;
    case 3:      // Fibonacci-like weights: the deepest trees, forces the length limiter
        {
            uint64_t a = 1, b = 1;
            for (int k = 0; k < 60; ++k) {
                h[rnd() % kDeflateSymbols] += a;
                const uint64_t t = a + b;
                a = b;
                b = t;
            }
        }
        break;
    case 4:      // huge counts (64 x 16384^2 pixels): no overflow in the weights
        for (int i = 0; i < kDeflateSymbols; ++i) h[i] = rnd() % 3 ? rnd() >> 20 : 0;
        break;
    default:     // sparse with zero runs of every length (exercises code-length symbols 17 / 18)
        for (int i = 0; i < kDeflateSymbols; ++i) h[i] = rnd() % 7 == 0 ? 1 + rnd() % 50 : 0;
        break;
    }
    h[256] = 1;      // end of block is always in use
}

int main(int argc, char **argv)
{
    const int cases = argc > 1 ? atoi(argv[1]) : 10000;
    rng_state = argc > 2 ? strtoull(argv[2], nullptr, 0) : 1;
    FILE *out = argc > 3 ? fopen(argv[3], "wb") : nullptr;
    int c = 0;
    for (; c < cases; ++c) {
        uint64_t hist[kDeflateSymbols];
        random_hist(hist, c % 6);
        uint8_t lens[kDeflateSymbols];
        uint16_t codes[kDeflateSymbols];
        uint8_t header[640 + 8];
        memset(header, 0, sizeof(header));
        const size_t bits = huffman_plan(hist, lens, codes, header, 640);
        REQUIRE(bits > 0 && bits <= 8 * 640, "header of %zu bits", bits);
        int used = 0;
        uint64_t kraft = 0;
        for (int s = 0; s < kDeflateSymbols; ++s) {
            REQUIRE(lens[s] <= 15, "symbol %d has length %d", s, lens[s]);
            REQUIRE((hist[s] != 0) == (lens[s] != 0), "symbol %d: count %llu, length %d", s, (unsigned long long)hist[s], lens[s]);
            if (lens[s]) {
                ++used;
                kraft += 1ull << (15 - lens[s]);
            }
        }
        REQUIRE(kraft <= (1ull << 15), "Kraft sum %llu / 32768", (unsigned long long)kraft);
        REQUIRE(used < 2 || kraft == (1ull << 15), "incomplete code: Kraft sum %llu / 32768 with %d symbols", (unsigned long long)kraft, used);
        // prefix-free: un-reverse the codes, sort by (length, code) order = canonical order, neighbours must differ on the
        // shorter one's length
        for (int a = 0; a < kDeflateSymbols; ++a) {
            if (!lens[a]) continue;
            for (int b = a + 1; b < kDeflateSymbols; ++b) {
                if (!lens[b]) continue;
                const int l = lens[a] < lens[b] ? lens[a] : lens[b];
                // codes are stored bit-reversed: the first l transmitted bits are the low l bits
                REQUIRE(((codes[a] ^ codes[b]) & ((1u << l) - 1u)) != 0, "codes of %d and %d share a prefix", a, b);
            }
            if (c % 50) break;      // the full O(n^2) check on every 50th case, one row otherwise
        }
        // a complete block: header, then the end-of-block code
        uint64_t at = bits;
        REQUIRE(huff::put_bits(header, sizeof(header), at, codes[256], lens[256]), "end of block does not fit");
        const uint32_t nbytes = (uint32_t)((at + 7) / 8);
        if (out) {
            fwrite(&nbytes, 4, 1, out);
            fwrite(header, 1, nbytes, out);
        }
        // too small a buffer is reported, never overrun
        uint8_t tiny[16];
        REQUIRE(huffman_plan(hist, lens, codes, tiny, sizeof(tiny)) == 0 || bits <= 8 * sizeof(tiny), "a %zu-bit header fit 16 bytes", bits);
        // the per-frame plan, as hgi_capi.hip drives it
        if (c % 10 == 0) {
            static uint64_t five[kMatchThresholds + 1][kDeflateSymbols];
            for (int v = 0; v <= kMatchThresholds; ++v) {
                random_hist(five[v], (c / 10 + v) % 6);
                five[v][256] = 0;
            }
            uint8_t prefix[8], suffix[8];
            for (int i = 0; i < 8; ++i) {
                prefix[i] = (uint8_t)rnd();
                suffix[i] = (uint8_t)rnd();
            }
            huff::FramePlan p;
            REQUIRE(huff::plan_frame(five, true, prefix, suffix, p), "plan_frame failed");
            REQUIRE(p.block.front_bytes == (p.block.base_bits + 7) / 8 && p.block.front_bytes <= sizeof(p.block.front), "front %u bytes for %llu bits",
                    p.block.front_bytes, (unsigned long long)p.block.base_bits);
            REQUIRE(p.block.tail_bits <= 9 * 15 && p.block.tail_bits > 0, "tail of %u bits", p.block.tail_bits);
            REQUIRE(p.exact_bits >= p.block.base_bits + p.block.tail_bits - 0, "stream of %llu bits", (unsigned long long)p.exact_bits);
            bool listed = false;
            for (int v = 0; v < kMatchThresholds; ++v) listed |= p.block.min_match == kMatchThresholdHost[v];
            REQUIRE(listed, "threshold %u", p.block.min_match);
        }
    }
    // the bit writer reports what does not fit instead of dropping it
    {
        huff::BitWriter bw;
        for (int i = 0; i < 400 && !bw.overflow; ++i) bw.put(0xFFFF, 16);
        REQUIRE(bw.overflow && bw.bits <= 8 * sizeof(bw.bytes), "BitWriter took %zu bits without complaint", bw.bits);
        uint8_t small[4] = {0, 0, 0, 0};
        uint64_t at = 30;
        REQUIRE(!huff::put_bits(small, sizeof(small), at, 0xFF, 8) && at == 30 && small[3] == 0, "put_bits wrote beyond its buffer");
    }
    if (out) fclose(out);
    printf("fuzz_huffman: %d cases ok\n", cases);
    return 0;
}
