// Entropy stage on the device: raw DEFLATE (RFC 1951) of a residual grid as ONE dynamic-Huffman block of literals and
// run matches.
//
// The reference serialises a grid by handing its bincode image to flate2's DEFLATE at the best level
// (src/archive.rs:34-40), on the CPU, one thread: 8 ms for a 1920 x 1080 grid against 7 us for the encode that made it
// (profiles/r02_bench_cpp.txt), seconds for a 4096 x 4096 one.  Residual grids are noise around zero: what LZ77 finds in
// them is runs (of zeros, in smooth regions) and nothing else -- zlib's run-length-only strategy (Z_RLE) is within 1 % of
// its level 9 on them, and where there are no runs the plain Huffman code alone is SMALLER than level 9 (LENA / Medium:
// 14.5 against 16.0 kB).  Both parallelise:
//   tokens     a byte equal to its predecessor continues a run; a run's bytes after its first are covered by matches of
//              distance 1 and length 3..258 (leftovers of 1-2 bytes stay literals).  Runs are cut at 1 KiB chunk
//              boundaries, so every token is decided inside one workgroup (cost: one extra literal per KiB of run).
//   histogram  of the 286 literal / length symbols the tokens use, for four match thresholds   (device, pass 1)
//   code       for each threshold: length-limited canonical Huffman code, exact stream size;    (host, a few hundred symbols)
//              the smallest wins (short runs of a byte whose literal costs 1 bit are cheaper as literals) + block header
//   bits       per chunk (pass 2), exclusive scan over the chunks, and every token OR-ed into place (pass 3)
// The stream is ordinary DEFLATE: flate2 / zlib / miniz inflate it; `Archive::deserialize_from_reader`
// (src/archive.rs:43-55) reads archives written this way unchanged.
#include <algorithm>
#include <cstring>
#include <vector>

#include "hgi_kernels.h"

namespace hgi {

namespace {

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

// ---------------------------------------------------------------------------------------------------------------
// host: code construction
// ---------------------------------------------------------------------------------------------------------------
// Optimal prefix-code lengths for `n` symbols (two-queue Huffman on the sorted frequencies), then limited to `maxlen`
// bits by moving leaves up the tree until the Kraft sum fits (the shortest over-long codes pay), lengths handed out in
// order of frequency.  Symbols of frequency 0 get length 0; a single used symbol gets length 1.
void code_lengths(const u64 *freq, int n, int maxlen, u8 *len)
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
void canonical_codes(const u8 *len, int n, int maxlen, uint16_t *code)
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
    BitWriter() { std::memset(bytes, 0, sizeof(bytes)); }
    void put(u32 value, int n)      // n <= 16 bits, least significant first
    {
        if (bits + (size_t)n > 8 * sizeof(bytes) - 32) return;      // (a header is < 300 bytes)
        u32 v = (value & ((1u << n) - 1u)) << (bits & 7);
        for (size_t at = bits >> 3; v; ++at, v >>= 8) bytes[at] |= (u8)v;
        bits += (size_t)n;
    }
};

}  // namespace

// RFC 1951 3.2.5: match length 3..258 -> length symbol 257..285, number of extra bits, value of the extra bits
void deflate_length_symbol(uint32_t length, uint32_t *symbol, uint32_t *extra_bits, uint32_t *extra)
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
uint64_t huffman_payload_bits(const uint64_t hist[kDeflateSymbols])
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
size_t huffman_plan(const uint64_t hist[kDeflateSymbols], uint8_t lens[kDeflateSymbols], uint16_t codes[kDeflateSymbols], uint8_t *header,
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
    if (nbytes > header_cap) return 0;
    std::memcpy(header, bw.bytes, nbytes);
    return bw.bits;
}

// ---------------------------------------------------------------------------------------------------------------
// device: tokens, histogram, sum, scan, pack
// ---------------------------------------------------------------------------------------------------------------
namespace {

constexpr int kPackThreads = 256;
constexpr int kBytesPerThread = 4;
constexpr int kChunk = kPackThreads * kBytesPerThread;   // bytes per workgroup: runs never cross a chunk boundary
constexpr u32 kNone = 0xFFFFFFFFu;
__device__ constexpr u32 kMatchThreshold[kMatchThresholds] = {3, 4, 6, 10};      // = kMatchThresholdHost (hgi_kernels.h)

// What the (up to) four bytes of this thread emit.  A byte "continues" when it equals its predecessor inside the chunk.
// For a continuing byte at offset o (0-based) inside the run's tail of M continuing bytes, piece = o / 258 and the piece's
// length is min(258, M - 258 * piece): the first byte of a piece of >= min_match (>= 3) emits the match, the others of that
// piece nothing; bytes of shorter pieces stay literals.  With k = bytes since the run's head (>= 1) and rem = continuing bytes
// left including this one: o = k - 1, M - 258 * piece = rem + (o % 258).
struct Tokens {
    u32 lit[4];      // the byte, or kNone beyond the end of the data
    u32 piece[4];    // 0: the byte heads a run (or stands alone); else the length (1..258) of the run piece it lies in
    u32 first[4];    // ... and whether it is that piece's first byte
};

// what byte i emits when pieces of at least `min_match` become matches: literal 0..255, kNone (nothing: covered by a
// match), or 0x80000000 | match length
__device__ __forceinline__ u32 token_symbol(const Tokens &tk, int i, u32 min_match)
{
    if (tk.lit[i] == kNone) return kNone;
    if (tk.piece[i] >= min_match) return tk.first[i] ? (0x80000000u | tk.piece[i]) : kNone;
    return tk.lit[i];
}

__device__ __forceinline__ Tokens tokens_of_thread(const u8 *__restrict__ src, u64 n, u64 chunk0, u32 *lds /* 2 * kPackThreads + 8 */)
{
    const u32 t = threadIdx.x;
    const u64 at = chunk0 + (u64)t * kBytesPerThread;
    const int cnt = at >= n ? 0 : (at + 4 <= n ? 4 : (int)(n - at));
    u32 w = 0;
    if (cnt == 4) {
        __builtin_memcpy(&w, src + at, 4);
    } else {
        for (int i = 0; i < cnt; ++i) w |= (u32)src[at + i] << (8 * i);
    }
    const u32 prev = (t > 0 && cnt > 0) ? src[at - 1] : 0x100u;      // the chunk's first byte never continues
    u32 b[4], cont[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = (w >> (8 * i)) & 255u;
    cont[0] = cnt > 0 && b[0] == prev;
#pragma unroll
    for (int i = 1; i < 4; ++i) cont[i] = i < cnt && b[i] == b[i - 1];
    // positions inside the chunk: p = 4 t + i.  head(p) = last position <= p that does not continue (exists: position 0);
    // end(p) = first position > p that does not continue, or the number of valid bytes in the chunk
    u32 last_head = kNone, first_head = kNone;      // of this thread's positions (kNone: all four continue / are invalid)
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < cnt && !cont[i]) {
            last_head = 4 * t + i;
            if (first_head == kNone) first_head = 4 * t + i;
        }
    // invalid positions (beyond n) end every run: treat the first invalid position as a head for the backward scan
    const u32 valid_end = cnt < 4 ? 4 * t + cnt : kNone;
    u32 first_stop = first_head < valid_end ? first_head : valid_end;
    // exclusive forward max-scan of last_head (kNone = -1: use +1 encoding), exclusive backward min-scan of first_stop
    u32 fwd = last_head + 1u;      // 0 = none
    u32 bwd = first_stop;          // kNone = none
    const u32 lane = t & 63u, wave = t >> 6;
    u32 fi = fwd, bi = bwd;
    for (int o = 1; o < 64; o <<= 1) {
        const u32 up = __shfl_up(fi, o, 64), dn = __shfl_down(bi, o, 64);
        if ((int)lane >= o) fi = fi > up ? fi : up;
        if ((int)lane + o < 64) bi = bi < dn ? bi : dn;
    }
    if (lane == 63) lds[wave] = fi;
    if (lane == 0) lds[4 + wave] = bi;
    __syncthreads();
    u32 head_before = __shfl_up(fi, 1, 64);
    if (lane == 0) head_before = 0;
    for (u32 wv = 0; wv < wave; ++wv) head_before = head_before > lds[wv] ? head_before : lds[wv];
    u32 stop_after = __shfl_down(bi, 1, 64);
    if (lane == 63) stop_after = kNone;
    for (u32 wv = wave + 1; wv < kPackThreads / 64; ++wv) stop_after = stop_after < lds[4 + wv] ? stop_after : lds[4 + wv];
    __syncthreads();                 // lds is reused by the caller
    // walk my four positions
    Tokens tk;
    u32 head = head_before;          // +1 encoded position of the latest head before my first byte (0 only for thread 0)
    // end position for each of my bytes: the next stop after it
    u32 stop[4];
    u32 nxt = stop_after;            // first stop strictly after my last position (kNone: none in the chunk)
    const u32 chunk_valid = (u32)(n - chunk0 < (u64)kChunk ? n - chunk0 : (u64)kChunk);
    if (nxt == kNone || nxt > chunk_valid) nxt = chunk_valid;
#pragma unroll
    for (int i = 3; i >= 0; --i) {
        stop[i] = nxt;
        if (i < cnt && !cont[i]) nxt = 4 * t + i;
        if (i >= cnt) nxt = nxt < 4 * t + (u32)i ? nxt : 4 * t + (u32)i;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u32 p = 4 * t + i;
        tk.lit[i] = kNone;
        tk.piece[i] = 0;
        tk.first[i] = 0;
        if (i >= cnt) continue;
        tk.lit[i] = b[i];
        if (!cont[i]) {
            head = p + 1;
            continue;
        }
        const u32 k = p - (head - 1);            // >= 1: bytes since the head
        const u32 rem = stop[i] - p;             // continuing bytes left including this one
        const u32 o = (k - 1) % 258u;
        const u32 piece = rem + o;
        tk.piece[i] = piece < 258u ? piece : 258u;
        tk.first[i] = o == 0;
    }
    return tk;
}

// length 3..258 -> length symbol, extra bits (RFC 1951 3.2.5), branch-free enough
__device__ __forceinline__ void length_symbol(u32 length, u32 &symbol, u32 &extra_bits, u32 &extra)
{
    if (length == 258) {
        symbol = 285;
        extra_bits = 0;
        extra = 0;
        return;
    }
    const u32 l = length - 3;                         // 0..254
    if (l < 8) {
        symbol = 257 + l;
        extra_bits = 0;
        extra = 0;
        return;
    }
    const u32 e = 29 - __clz(l);                      // extra bits: l in [8,16) -> 1, [16,32) -> 2, ...
    symbol = 257 + 4 * e + 4 + ((l >> e) & 3u);
    extra_bits = e;
    extra = l & ((1u << e) - 1u);
}

// pass 1: histograms of the literal / length symbols the tokens use, for each candidate match threshold at once (a short
// run of a byte whose literal costs one bit is cheaper as literals than as a match; which threshold pays is decided on
// the host from the exact stream sizes the histograms imply).  A byte's token is the same under every candidate unless
// it lies in a run piece of kMatchThreshold[0] .. kMatchThreshold[last] - 1 bytes: those go to a per-candidate histogram,
// everything else to ONE common histogram (slot kMatchThresholds) that the host adds to each.
__global__ __launch_bounds__(kPackThreads) void k_token_hist(const u8 *__restrict__ src, u64 n, unsigned long long *__restrict__ hist)
{
    constexpr int kCopies = 4, kHists = kMatchThresholds + 1;
    __shared__ u32 scan[16];
    __shared__ u32 h[kHists][kDeflateSymbols * kCopies];      // copies value-major (a handful of symbols dominate)
    for (int i = threadIdx.x; i < kHists * kDeflateSymbols * kCopies; i += kPackThreads) (&h[0][0])[i] = 0;
    __syncthreads();
    const u32 copy = threadIdx.x & (kCopies - 1);
    auto add = [&](int which, u32 s) {
        u32 sym = s;
        if (s & 0x80000000u) {
            u32 eb, ex;
            length_symbol(s & 0xFFFFu, sym, eb, ex);
        }
        atomicAdd(&h[which][sym * kCopies + copy], 1u);
    };
    for (u64 chunk = blockIdx.x; chunk * kChunk < n; chunk += gridDim.x) {
        const Tokens tk = tokens_of_thread(src, n, chunk * kChunk, scan);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool valid = tk.lit[i] != kNone;
            const bool common = tk.piece[i] < kMatchThreshold[0] || tk.piece[i] >= kMatchThreshold[kMatchThresholds - 1];
            const u32 s = token_symbol(tk, i, kMatchThreshold[0]);       // what every candidate emits when `common`
            // literal zeros are most of a residual grid: count them per wave with a ballot, one add per wave
            const unsigned long long zeros = __ballot(valid && common && s == 0u);
            if (zeros && (threadIdx.x & 63u) == (u32)__ffsll((long long)zeros) - 1u)
                atomicAdd(&h[kMatchThresholds][0 * kCopies + copy], (u32)__popcll(zeros));
            if (!valid) continue;
            if (common) {
                if (s != kNone && s != 0u) add(kMatchThresholds, s);
                continue;
            }
#pragma unroll
            for (int v = 0; v < kMatchThresholds; ++v) {
                const u32 sv = token_symbol(tk, i, kMatchThreshold[v]);
                if (sv != kNone) add(v, sv);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kHists * kDeflateSymbols; i += kPackThreads) {
        const int v = i / kDeflateSymbols, sym = i - v * kDeflateSymbols;
        u32 sum = 0;
#pragma unroll
        for (int k = 0; k < kCopies; ++k) sum += h[v][sym * kCopies + k];
        if (sum) atomicAdd(&hist[i], (unsigned long long)sum);
    }
}

// the bits of this thread's tokens (table[s] = reversed code | length << 16 for the 286 symbols; dist = the one distance
// code used, same packing): at most 3 x 15 + (15 + 5 + 1) = 66 bits in 4 bytes -> a 128-bit accumulator
__device__ __forceinline__ void pack_tokens(const Tokens &tk, u32 min_match, const u32 *stab, u32 dist, unsigned __int128 &val, u32 &bits)
{
    val = 0;
    bits = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u32 s = token_symbol(tk, i, min_match);
        if (s == kNone) continue;
        if (s & 0x80000000u) {
            u32 sym, eb, ex;
            length_symbol(s & 0xFFFFu, sym, eb, ex);
            const u32 e = stab[sym];
            val |= (unsigned __int128)(e & 0xFFFFu) << bits;
            bits += e >> 16;
            val |= (unsigned __int128)ex << bits;
            bits += eb;
            val |= (unsigned __int128)(dist & 0xFFFFu) << bits;
            bits += dist >> 16;
        } else {
            const u32 e = stab[s];
            val |= (unsigned __int128)(e & 0xFFFFu) << bits;
            bits += e >> 16;
        }
    }
}

// pass 2: bits per chunk
__global__ __launch_bounds__(kPackThreads) void k_token_count(const u8 *__restrict__ src, u64 n, const u32 *__restrict__ table, u32 dist,
                                                              u32 min_match, u32 *__restrict__ chunk_bits)
{
    __shared__ u32 stab[kDeflateSymbols];
    __shared__ u32 scan[16];
    __shared__ u32 wsum[kPackThreads / 64];
    for (int i = threadIdx.x; i < kDeflateSymbols; i += kPackThreads) stab[i] = table[i];
    const Tokens tk = tokens_of_thread(src, n, (u64)blockIdx.x * kChunk, scan);      // (its barriers publish stab too)
    unsigned __int128 val;
    u32 bits;
    pack_tokens(tk, min_match, stab, dist, val, bits);
    for (int o = 32; o > 0; o >>= 1) bits += __shfl_down(bits, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = bits;
    __syncthreads();
    if (threadIdx.x == 0) chunk_bits[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of the chunk sizes into bit offsets; one workgroup walks the chunks 1024 at a time (coalesced reads and
// writes; shuffle scan inside each wave, the sixteen wave totals through LDS, a running total carried between blocks)
__global__ __launch_bounds__(1024) void k_huff_scan(const u32 *__restrict__ chunk_bits, u64 *__restrict__ chunk_off, u32 nchunks,
                                                    u64 *__restrict__ total)
{
    __shared__ u64 wtot[16];
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    u64 carry = 0;                      // every thread keeps its own copy of the running total
    constexpr u32 kAhead = 8;           // blocks of 1024 whose loads are issued together (the loop is a latency chain otherwise)
    for (u32 super = 0; super < nchunks; super += kAhead * 1024) {
        u32 mine[kAhead];
#pragma unroll
        for (u32 k = 0; k < kAhead; ++k) {
            const u32 i = super + k * 1024 + threadIdx.x;
            mine[k] = i < nchunks ? chunk_bits[i] : 0;
        }
#pragma unroll
        for (u32 k = 0; k < kAhead; ++k) {
            const u32 i = super + k * 1024 + threadIdx.x;
            u64 incl = mine[k];
            for (int o = 1; o < 64; o <<= 1) {
                const u64 up = __shfl_up(incl, o, 64);
                if ((int)lane >= o) incl += up;
            }
            if (lane == 63) wtot[wave] = incl;
            __syncthreads();
            u64 before = 0, all = 0;
            for (u32 wv = 0; wv < 16; ++wv) {
                const u64 t = wtot[wv];
                before += wv < wave ? t : 0;
                all += t;
            }
            if (i < nchunks) chunk_off[i] = carry + before + incl - mine[k];
            carry += all;
            __syncthreads();            // wtot is rewritten by the next block
        }
    }
    if (threadIdx.x == 0) *total = carry;
}

// pass 3: the chunk's tokens are OR-ed into an LDS image of the chunk's part of the stream (<= 1024 x 21 bits), which then
// goes out as whole words -- plain coalesced stores for the words the chunk owns alone, atomicOr only for its first and
// last word, which it shares with its neighbours (the stream was zeroed)
__global__ __launch_bounds__(kPackThreads) void k_token_pack(const u8 *__restrict__ src, u64 n, const u32 *__restrict__ table, u32 dist,
                                                             u32 min_match, const u64 *__restrict__ chunk_off, u64 base_bits,
                                                             u32 *__restrict__ out)
{
    constexpr int kWords = (kChunk * 21 + 31) / 32 + 3;      // worst case: every byte a 15-bit literal; matches are rarer
    __shared__ u32 stab[kDeflateSymbols];
    __shared__ u32 scan[16];
    __shared__ u32 wsum[kPackThreads / 64];
    __shared__ u32 img[kWords];
    for (int i = threadIdx.x; i < kDeflateSymbols; i += kPackThreads) stab[i] = table[i];
    for (int i = threadIdx.x; i < kWords; i += kPackThreads) img[i] = 0;
    const Tokens tk = tokens_of_thread(src, n, (u64)blockIdx.x * kChunk, scan);      // (its barriers publish stab and img)
    unsigned __int128 val;
    u32 bits;
    pack_tokens(tk, min_match, stab, dist, val, bits);
    // exclusive scan of `bits` over the workgroup: inside the wave by shuffles, across the four waves through LDS
    u32 incl = bits;
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const u32 up = __shfl_up(incl, o, 64);
        if ((int)lane >= o) incl += up;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    u32 before = incl - bits;
    for (u32 wv = 0; wv < wave; ++wv) before += wsum[wv];
    const u32 chunk_total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    const u64 pos0 = base_bits + chunk_off[blockIdx.x];      // the chunk's first bit in the stream
    const u32 sh0 = (u32)(pos0 & 31u);                       // ... and where that is inside its first word
    if (bits) {
        const u32 at = sh0 + before, sh = at & 31u;
        const unsigned __int128 shifted = val << sh;          // <= 66 + 31 bits
        const u32 o0 = (u32)shifted, o1 = (u32)(shifted >> 32), o2 = (u32)(shifted >> 64), o3 = (u32)(shifted >> 96);
        u32 *dst = img + (at >> 5);
        if (o0) atomicOr(dst, o0);
        if (o1) atomicOr(dst + 1, o1);
        if (o2) atomicOr(dst + 2, o2);
        if (o3) atomicOr(dst + 3, o3);
    }
    __syncthreads();
    if (!chunk_total) return;
    const u32 nwords = (sh0 + chunk_total + 31u) >> 5;
    u32 *dst = out + (pos0 >> 5);
    for (u32 w = threadIdx.x; w < nwords; w += kPackThreads) {
        const u32 v = img[w];
        if (w == 0 || w == nwords - 1) {
            if (v) atomicOr(dst + w, v);
        } else {
            dst[w] = v;
        }
    }
}

}  // namespace

u32 huffman_chunks(u64 n) { return (u32)((n + kChunk - 1) / kChunk); }

hipError_t launch_token_histogram(const uint8_t *src, uint64_t n, unsigned long long *d_hist, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(d_hist, 0, (kMatchThresholds + 1) * kDeflateSymbols * sizeof(unsigned long long), s);
    if (e != hipSuccess || n == 0) return e;
    u32 blocks = huffman_chunks(n);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_token_hist, dim3(blocks), dim3(kPackThreads), 0, s, src, n, d_hist);
    return hipGetLastError();
}

hipError_t launch_huffman_pack(const uint8_t *src, uint64_t n, const uint32_t *d_table, uint32_t dist_code, uint32_t min_match,
                               uint32_t *d_chunk_bits, uint64_t *d_chunk_off, uint64_t *d_total, uint64_t base_bits, uint32_t *d_out,
                               hipStream_t s)
{
    const u32 nchunks = huffman_chunks(n);
    if (nchunks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_token_count, dim3(nchunks), dim3(kPackThreads), 0, s, src, n, d_table, dist_code, min_match, d_chunk_bits);
    hipLaunchKernelGGL(k_huff_scan, dim3(1), dim3(1024), 0, s, d_chunk_bits, d_chunk_off, nchunks, d_total);
    hipLaunchKernelGGL(k_token_pack, dim3(nchunks), dim3(kPackThreads), 0, s, src, n, d_table, dist_code, min_match, d_chunk_off, base_bits, d_out);
    return hipGetLastError();
}

}  // namespace hgi
