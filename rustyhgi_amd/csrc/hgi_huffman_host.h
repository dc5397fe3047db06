// Host side of the entropy stage, plain C++17 (no HIP): the length-limited canonical Huffman code for the 286 DEFLATE
// literal / length symbols, the RFC 1951 dynamic-block header that announces it, and the per-frame plan block the
// device's pack pass reads.  hgi_entropy.hip and hgi_capi.hip include it; tests/cpp/fuzz_huffman.cpp builds it alone
// with g++ -fsanitize=address,undefined and fuzzes it (tests/test_sanitizers.py).
// Reference: the step behind src/archive.rs:34-40 (flate2's DeflateEncoder); any inflate reads what this plans.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace hgi {

constexpr int kDeflateSymbols = 286;      // literals 0..255, end of block 256, match lengths 257..285 (RFC 1951 3.2.5)
constexpr int kMatchThresholds = 4;       // run pieces become matches from this length on: candidates, the host picks
constexpr uint32_t kMatchThresholdHost[kMatchThresholds] = {3, 4, 6, 10};
// What the host hands each frame's passes 2 and 3 (one upload for a whole group of frames): the code, the chosen
// threshold, the bytes in front of the grid's tokens and the bits behind them.
constexpr size_t kPlanBytes = 2048;
struct DeflatePlan {
    uint32_t table[kDeflateSymbols];      // reversed code | length << 24
    uint32_t min_match;
    uint32_t front_bytes;                 // block header + the eight literals of the u64 length: ceil(base_bits / 8)
    uint64_t base_bits;                   // where the grid's tokens start
    uint32_t tail_bits;                   // the eight literals of the u64 width + end of block: <= 9 x 15 bits
    uint32_t tail[5];
    uint32_t out_off[2];                  // where this frame's stream starts in the launch's output buffer (u64, low word first; a multiple of 4)
    uint32_t reserved[4];
    uint8_t front[kPlanBytes - 1208];     // zero padded
};
static_assert(sizeof(DeflatePlan) == kPlanBytes, "plan block layout");

namespace huff {
typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

// Optimal prefix-code lengths for `n` symbols (two-queue Huffman on the sorted frequencies), then limited to `maxlen`
// bits by moving leaves up the tree until the Kraft sum fits (the shortest over-long codes pay), lengths handed out in
// order of frequency.  Symbols of frequency 0 get length 0; a single used symbol gets length 1.
inline void code_lengths(const u64 *freq, int n, int maxlen, u8 *len)
{
    constexpr int kMax = kDeflateSymbols;      // n <= kMax: everything on the stack (this runs per frame, per candidate)
    int used[kMax], m = 0;
    for (int i = 0; i < n; ++i) {
        len[i] = 0;
        if (freq[i]) used[m++] = i;
    }
    if (m == 0) return;
    if (m == 1) {
        len[used[0]] = 1;
        return;
    }
    std::sort(used, used + m, [&](int a, int b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
    // nodes 0..m-1: leaves in ascending frequency; m..2m-2: internal nodes in order of creation (also ascending)
    u64 w[2 * kMax];
    int parent[2 * kMax], depth[2 * kMax];
    for (int i = 0; i < m; ++i) w[i] = freq[used[i]];
    int leaf = 0, inner = m, next = m;
    auto take = [&]() {
        if (leaf < m && (inner >= next || w[leaf] <= w[inner])) return leaf++;
        return inner++;
    };
    for (; next < 2 * m - 1; ++next) {
        const int a = take(), b = take();
        w[next] = w[a] + w[b];
        parent[a] = parent[b] = next;
    }
    depth[2 * m - 2] = 0;
    for (int i = 2 * m - 3; i >= 0; --i) depth[i] = depth[parent[i]] + 1;
    // how many codes of each length; fold what is too long into maxlen and repair the Kraft sum
    int count[17] = {0};
    for (int i = 0; i < m; ++i) ++count[std::min(depth[i], maxlen)];
    u64 kraft = 0;                                       // in units of 2^-maxlen
    for (int l = 1; l <= maxlen; ++l) kraft += (u64)count[l] << (maxlen - l);
    while (kraft > ((u64)1 << maxlen)) {
        // take one code of the longest length away with one of the next shorter length that exists: the shorter one
        // becomes two codes one bit longer, and one code of length maxlen disappears into that pair
        --count[maxlen];
        for (int l = maxlen - 1; l > 0; --l)
            if (count[l]) {
                --count[l];
                count[l + 1] += 2;
                break;
            }
        --kraft;
    }
    // longest codes to the rarest symbols
    int at = 0;
    for (int l = maxlen; l >= 1; --l)
        for (int k = 0; k < count[l]; ++k) len[used[at++]] = (u8)l;
}

// canonical codes (RFC 1951 3.2.2), returned bit-reversed: DEFLATE packs codes starting from their most significant
// bit into a stream that fills bytes from the least significant bit, so a reversed code can simply be OR-ed in
inline void canonical_codes(const u8 *len, int n, int maxlen, uint16_t *code)
{
    u32 count[17] = {0}, next[18] = {0};
    for (int i = 0; i < n; ++i) ++count[len[i]];
    count[0] = 0;
    u32 c = 0;
    for (int l = 1; l <= maxlen; ++l) {
        c = (c + count[l - 1]) << 1;
        next[l] = c;
    }
    for (int i = 0; i < n; ++i) {
        code[i] = 0;
        if (!len[i]) continue;
        u32 v = next[len[i]]++, r = 0;
        for (int b = 0; b < len[i]; ++b) r |= ((v >> b) & 1u) << (len[i] - 1 - b);
        code[i] = (uint16_t)r;
    }
}

struct BitWriter {
    u8 bytes[640];
    size_t bits = 0;
    bool overflow = false;          // a put that did not fit: the header is unusable and huffman_plan says so
    BitWriter() { std::memset(bytes, 0, sizeof(bytes)); }
    void put(u32 value, int n)      // n <= 16 bits, least significant first
    {
        if (bits + (size_t)n > 8 * sizeof(bytes) - 32) {             // (a header is < 300 bytes)
            overflow = true;
            return;
        }
        u32 v = (value & ((1u << n) - 1u)) << (bits & 7);
        for (size_t at = bits >> 3; v; ++at, v >>= 8) bytes[at] |= (u8)v;
        bits += (size_t)n;
    }
};

// RFC 1951 3.2.5: match length 3..258 -> length symbol 257..285, number of extra bits, value of the extra bits
inline void deflate_length_symbol(uint32_t length, uint32_t *symbol, uint32_t *extra_bits, uint32_t *extra)
{
    static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const u8 bits[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    int s = 28;
    while (s > 0 && base[s] > length) --s;
    *symbol = 257u + (u32)s;
    *extra_bits = bits[s];
    *extra = length - base[s];
}

// The bits the tokens themselves take under the best code for `hist` (codes + length extra bits + one distance bit per
// match; without the block header, which varies by a few dozen bits between candidates): what the choice of the match
// threshold is made on.
inline uint64_t huffman_payload_bits(const uint64_t hist[kDeflateSymbols])
{
    u8 lens[kDeflateSymbols];
    code_lengths(hist, kDeflateSymbols, 15, lens);
    static const u8 extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    u64 total = 0;
    for (int s = 0; s < kDeflateSymbols; ++s) total += hist[s] * lens[s];
    for (int s = 257; s < kDeflateSymbols; ++s) total += hist[s] * (extra[s - 257] + 1u);
    return total;
}

// Code for the 286 literal / end-of-block / length symbols from their frequencies (hist[256] = end of block, normally
// 1) and the header of the one block that carries them: BFINAL = 1, BTYPE = dynamic, 286 literal/length codes, two
// distance codes of one bit each (code 0 = distance 1, the only distance a run match uses; the second only completes the
// code, which is what every inflate accepts), the code lengths themselves Huffman-coded with zero runs folded
// (RFC 1951 3.2.7).  Returns the header's length in bits, 0 if it does not fit.
inline size_t huffman_plan(const uint64_t hist[kDeflateSymbols], uint8_t lens[kDeflateSymbols], uint16_t codes[kDeflateSymbols], uint8_t *header,
                    size_t header_cap)
{
    code_lengths(hist, kDeflateSymbols, 15, lens);
    canonical_codes(lens, kDeflateSymbols, 15, codes);
    // the 288 code lengths to transmit, zero runs as symbols 17 (3..10) / 18 (11..138)
    constexpr int kSeq = kDeflateSymbols + 2;
    u8 seq[kSeq];
    std::memcpy(seq, lens, kDeflateSymbols);
    seq[kSeq - 2] = seq[kSeq - 1] = 1;
    struct Item {
        u8 sym, extra_bits;
        uint16_t extra;
    };
    Item items[kSeq];
    int nitems = 0;
    for (int i = 0; i < kSeq;) {
        int run = 1;
        while (i + run < kSeq && seq[i + run] == seq[i]) ++run;
        if (seq[i] == 0 && run >= 3) {
            const int r = std::min(run, 138);
            if (r <= 10)
                items[nitems++] = {17, 3, (uint16_t)(r - 3)};
            else
                items[nitems++] = {18, 7, (uint16_t)(r - 11)};
            i += r;
        } else {
            items[nitems++] = {seq[i], 0, 0};
            ++i;
        }
    }
    u64 clfreq[19] = {0};
    for (int k = 0; k < nitems; ++k) ++clfreq[items[k].sym];
    u8 cllen[19];
    uint16_t clcode[19];
    code_lengths(clfreq, 19, 7, cllen);
    canonical_codes(cllen, 19, 7, clcode);
    static const u8 order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && cllen[order[hclen - 1]] == 0) --hclen;
    BitWriter bw;
    bw.put(1, 1);                               // BFINAL
    bw.put(2, 2);                               // BTYPE = 10: dynamic Huffman
    bw.put(kDeflateSymbols - 257, 5);           // HLIT
    bw.put(2 - 1, 5);                           // HDIST
    bw.put((u32)(hclen - 4), 4);                // HCLEN
    for (int i = 0; i < hclen; ++i) bw.put(cllen[order[i]], 3);
    for (int k = 0; k < nitems; ++k) {
        const Item &it = items[k];
        bw.put(clcode[it.sym], cllen[it.sym]);
        if (it.extra_bits) bw.put(it.extra, it.extra_bits);
    }
    const size_t nbytes = (bw.bits + 7) / 8;
    if (bw.overflow || nbytes > header_cap) return 0;
    std::memcpy(header, bw.bytes, nbytes);
    return bw.bits;
}

// Append `nb` bits of `value`, least significant first, at bit position `at` of `v` (`cap_bytes` long).  Returns false --
// and writes nothing -- when they would not fit.
inline bool put_bits(uint8_t *v, size_t cap_bytes, uint64_t &at, uint32_t value, int nb)
{
    if (nb < 0 || (at + (uint64_t)nb + 7) / 8 > cap_bytes) return false;
    for (int i = 0; i < nb; ++i, ++at) v[at >> 3] |= (uint8_t)(((value >> i) & 1u) << (at & 7));
    return true;
}

struct FramePlan {
    DeflatePlan block;               // what the device gets
    uint64_t exact_bits = 0;         // length of the whole stream, known from the histograms
};

// hists: [kMatchThresholds + 1][kDeflateSymbols] as downloaded (modified in place): slot v < kMatchThresholds counts the
// tokens that depend on candidate threshold v, the last slot those that do not.  prefix / suffix: the eight bytes in front
// of and behind the grid in its bincode image (u64 length, u64 width).  Picks the threshold whose code gives the smallest
// stream, builds that code, the block header and the literals around the grid.  False when the header does not fit.
inline bool plan_frame(uint64_t (*hists)[kDeflateSymbols], bool have_grid, const uint8_t prefix[8], const uint8_t suffix[8], FramePlan &p)
{
    if (have_grid)
        for (int v = 0; v < kMatchThresholds; ++v)
            for (int sym = 0; sym < kDeflateSymbols; ++sym) hists[v][sym] += hists[kMatchThresholds][sym];
    // for each candidate threshold: the code its histogram asks for and the exact size of the tokens under it; keep the
    // smallest (the 16 bytes around the grid are literals under every threshold)
    uint64_t best = ~0ull;
    int pick = 0;
    for (int v = 0; v < kMatchThresholds; ++v) {
        uint64_t *hv = hists[v];
        for (int i = 0; i < 8; ++i) {
            ++hv[prefix[i]];
            ++hv[suffix[i]];
        }
        hv[256] = 1;      // end of block
        const uint64_t total = huffman_payload_bits(hv);
        if (total < best) {
            best = total;
            pick = v;
        }
        if (!have_grid) break;
    }
    uint8_t lens[kDeflateSymbols];
    uint16_t codes[kDeflateSymbols];
    DeflatePlan &d = p.block;
    std::memset(&d, 0, sizeof(d));
    const size_t bits = huffman_plan(hists[pick], lens, codes, d.front, sizeof(d.front) - 32);
    if (!bits) return false;
    d.min_match = kMatchThresholdHost[pick];
    p.exact_bits = best + bits;
    d.base_bits = bits;
    for (int i = 0; i < 8; ++i)
        if (!put_bits(d.front, sizeof(d.front), d.base_bits, codes[prefix[i]], lens[prefix[i]])) return false;
    d.front_bytes = (uint32_t)((d.base_bits + 7) / 8);
    uint64_t tb = 0;
    uint8_t tail[sizeof(d.tail)] = {};
    for (int i = 0; i < 8; ++i)
        if (!put_bits(tail, sizeof(tail), tb, codes[suffix[i]], lens[suffix[i]])) return false;
    if (!put_bits(tail, sizeof(tail), tb, codes[256], lens[256])) return false;
    d.tail_bits = (uint32_t)tb;
    std::memcpy(d.tail, tail, sizeof(tail));
    for (int v = 0; v < kDeflateSymbols; ++v) d.table[v] = (uint32_t)codes[v] | ((uint32_t)lens[v] << 24);
    return true;
}

}  // namespace huff

using huff::deflate_length_symbol;
using huff::huffman_payload_bits;
using huff::huffman_plan;

}  // namespace hgi
