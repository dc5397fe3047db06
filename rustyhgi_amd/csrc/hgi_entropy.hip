// Entropy stage on the device: raw DEFLATE (RFC 1951) of a residual grid as ONE dynamic-Huffman block of literals.
//
// The reference serialises a grid by handing its bincode image to flate2's DEFLATE at the best level
// (src/archive.rs:34-40), on the CPU, one thread: 8.5 ms for a 1920 x 1080 grid against 7 us for the encode that made it
// (profiles/r02_bench_cpp.txt).  Residual grids are noise around zero: LZ77 matches find almost nothing in them (zlib
// level 9 on LENA / Medium: 16 036 B; Huffman only: 14 496 B -- SMALLER), so the stage that matters is the Huffman code,
// and that parallelises: histogram (hgi_kernels.hip) -> code lengths and block header on the host (a few
// hundred symbols) -> code lengths summed per chunk, scanned, and every chunk's codes OR-ed into place by the device.
// The stream is ordinary DEFLATE: flate2 / zlib / miniz inflate it; `Archive::deserialize_from_reader`
// (src/archive.rs:43-55) reads archives written this way unchanged.
//
// This file: the host-side planner (length-limited canonical Huffman code, RFC 1951 block header) and the three kernels.
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
    std::vector<int> used;
    for (int i = 0; i < n; ++i) {
        len[i] = 0;
        if (freq[i]) used.push_back(i);
    }
    const int m = (int)used.size();
    if (m == 0) return;
    if (m == 1) {
        len[used[0]] = 1;
        return;
    }
    std::sort(used.begin(), used.end(), [&](int a, int b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
    // nodes 0..m-1: leaves in ascending frequency; m..2m-2: internal nodes in order of creation (also ascending)
    std::vector<u64> w(2 * (size_t)m - 1);
    std::vector<int> parent(2 * (size_t)m - 1, -1);
    for (int i = 0; i < m; ++i) w[(size_t)i] = freq[used[(size_t)i]];
    int leaf = 0, inner = m, next = m;
    auto take = [&]() {
        if (leaf < m && (inner >= next || w[(size_t)leaf] <= w[(size_t)inner])) return leaf++;
        return inner++;
    };
    for (; next < 2 * m - 1; ++next) {
        const int a = take(), b = take();
        w[(size_t)next] = w[(size_t)a] + w[(size_t)b];
        parent[(size_t)a] = parent[(size_t)b] = next;
    }
    std::vector<int> depth(2 * (size_t)m - 1, 0);
    for (int i = 2 * m - 3; i >= 0; --i) depth[(size_t)i] = depth[(size_t)parent[(size_t)i]] + 1;
    // how many codes of each length; fold what is too long into maxlen and repair the Kraft sum
    std::vector<int> count((size_t)std::max(maxlen, m) + 2, 0);
    for (int i = 0; i < m; ++i) ++count[(size_t)std::min(depth[(size_t)i], maxlen)];
    u64 kraft = 0;                                       // in units of 2^-maxlen
    for (int l = 1; l <= maxlen; ++l) kraft += (u64)count[(size_t)l] << (maxlen - l);
    while (kraft > ((u64)1 << maxlen)) {
        // take one code of the longest length away with one of the next shorter length that exists: the shorter one
        // becomes two codes one bit longer, and one code of length maxlen disappears into that pair
        --count[(size_t)maxlen];
        for (int l = maxlen - 1; l > 0; --l)
            if (count[(size_t)l]) {
                --count[(size_t)l];
                count[(size_t)l + 1] += 2;
                break;
            }
        --kraft;
    }
    // longest codes to the rarest symbols
    int at = 0;
    for (int l = maxlen; l >= 1; --l)
        for (int k = 0; k < count[(size_t)l]; ++k) len[used[(size_t)at++]] = (u8)l;
}

// canonical codes (RFC 1951 3.2.2), returned bit-reversed: DEFLATE packs codes starting from their most significant
// bit into a stream that fills bytes from the least significant bit, so a reversed code can simply be OR-ed in
void canonical_codes(const u8 *len, int n, int maxlen, uint16_t *code)
{
    std::vector<u32> count((size_t)maxlen + 1, 0), next((size_t)maxlen + 2, 0);
    for (int i = 0; i < n; ++i) ++count[len[i]];
    count[0] = 0;
    u32 c = 0;
    for (int l = 1; l <= maxlen; ++l) {
        c = (c + count[(size_t)l - 1]) << 1;
        next[(size_t)l] = c;
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
    std::vector<u8> bytes;
    size_t bits = 0;
    void put(u32 value, int n)      // n <= 24 bits, least significant first
    {
        for (int i = 0; i < n; ++i, ++bits) {
            if ((bits & 7) == 0) bytes.push_back(0);
            bytes.back() |= (u8)(((value >> i) & 1u) << (bits & 7));
        }
    }
};

}  // namespace

// Code for the 257 literal / end-of-block symbols from their frequencies (hist[256] = end of block, normally 1) and the
// header of the one block that carries them: BFINAL = 1, BTYPE = dynamic, 257 literal/length codes, two distance codes
// of one bit each (never used; a complete distance code is what every inflate accepts), the code lengths themselves
// Huffman-coded with zero runs folded (RFC 1951 3.2.7).  Returns the header's length in bits, 0 if it does not fit.
size_t huffman_plan(const uint64_t hist[257], uint8_t lens[257], uint16_t codes[257], uint8_t *header, size_t header_cap)
{
    code_lengths(hist, 257, 15, lens);
    canonical_codes(lens, 257, 15, codes);
    // the 259 code lengths to transmit, zero runs as symbols 17 (3..10) / 18 (11..138)
    u8 seq[259];
    std::memcpy(seq, lens, 257);
    seq[257] = seq[258] = 1;
    struct Item {
        u8 sym, extra_bits;
        uint16_t extra;
    };
    std::vector<Item> items;
    for (int i = 0; i < 259;) {
        int run = 1;
        while (i + run < 259 && seq[i + run] == seq[i]) ++run;
        if (seq[i] == 0 && run >= 3) {
            const int r = std::min(run, 138);
            if (r <= 10)
                items.push_back({17, 3, (uint16_t)(r - 3)});
            else
                items.push_back({18, 7, (uint16_t)(r - 11)});
            i += r;
        } else {
            items.push_back({seq[i], 0, 0});
            ++i;
        }
    }
    u64 clfreq[19] = {0};
    for (const Item &it : items) ++clfreq[it.sym];
    u8 cllen[19];
    uint16_t clcode[19];
    code_lengths(clfreq, 19, 7, cllen);
    canonical_codes(cllen, 19, 7, clcode);
    static const u8 order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && cllen[order[hclen - 1]] == 0) --hclen;
    BitWriter bw;
    bw.put(1, 1);                 // BFINAL
    bw.put(2, 2);                 // BTYPE = 10: dynamic Huffman
    bw.put(257 - 257, 5);         // HLIT
    bw.put(2 - 1, 5);             // HDIST
    bw.put((u32)(hclen - 4), 4);  // HCLEN
    for (int i = 0; i < hclen; ++i) bw.put(cllen[order[i]], 3);
    for (const Item &it : items) {
        bw.put(clcode[it.sym], cllen[it.sym]);
        if (it.extra_bits) bw.put(it.extra, it.extra_bits);
    }
    if (bw.bytes.size() > header_cap) return 0;
    std::memcpy(header, bw.bytes.data(), bw.bytes.size());
    return bw.bits;
}

// ---------------------------------------------------------------------------------------------------------------
// device: sum, scan, pack
// ---------------------------------------------------------------------------------------------------------------
namespace {

constexpr int kPackThreads = 256;
constexpr int kBytesPerThread = 4;                       // <= 60 code bits: one 64-bit accumulator, no arrays
constexpr int kChunk = kPackThreads * kBytesPerThread;   // bytes per workgroup

// table[v] = reversed code | length << 16
__device__ __forceinline__ void pack4(const u8 *__restrict__ src, u64 n, u64 at, const u32 *stab, u64 &val, u32 &bits)
{
    val = 0;
    bits = 0;
    if (at >= n) return;
    u32 w = 0;
    if (at + 4 <= n) {
        __builtin_memcpy(&w, src + at, 4);
    } else {
        for (u64 i = at; i < n; ++i) w |= (u32)src[i] << (8 * (i - at));
    }
    const int cnt = at + 4 <= n ? 4 : (int)(n - at);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        if (b < cnt) {
            const u32 e = stab[(w >> (8 * b)) & 255u];
            val |= (u64)(e & 0xFFFFu) << bits;
            bits += e >> 16;
        }
    }
}

__global__ __launch_bounds__(kPackThreads) void k_huff_count(const u8 *__restrict__ src, u64 n, const u32 *__restrict__ table,
                                                             u32 *__restrict__ chunk_bits)
{
    __shared__ u32 stab[256];
    __shared__ u32 wsum[kPackThreads / 64];
    stab[threadIdx.x] = table[threadIdx.x];
    __syncthreads();
    u64 val;
    u32 bits;
    pack4(src, n, (u64)blockIdx.x * kChunk + threadIdx.x * kBytesPerThread, stab, val, bits);
    for (int o = 32; o > 0; o >>= 1) bits += __shfl_down(bits, o, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = bits;
    __syncthreads();
    if (threadIdx.x == 0) chunk_bits[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of the chunk sizes into bit offsets; one workgroup
__global__ __launch_bounds__(1024) void k_huff_scan(const u32 *__restrict__ chunk_bits, u64 *__restrict__ chunk_off, u32 nchunks,
                                                    u64 *__restrict__ total)
{
    __shared__ u64 part[1024];
    const u32 per = (nchunks + 1023u) / 1024u, lo = threadIdx.x * per, hi = lo + per < nchunks ? lo + per : nchunks;
    u64 sum = 0;
    for (u32 i = lo; i < hi; ++i) sum += chunk_bits[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (u32 o = 1; o < 1024; o <<= 1) {      // Hillis-Steele inclusive scan
        const u64 add = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    u64 run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    for (u32 i = lo; i < hi; ++i) {
        chunk_off[i] = run;
        run += chunk_bits[i];
    }
    if (threadIdx.x == 1023) *total = part[1023];
}

// every thread ORs the codes of its four bytes into the (zeroed) stream at its bit position
__global__ __launch_bounds__(kPackThreads) void k_huff_pack(const u8 *__restrict__ src, u64 n, const u32 *__restrict__ table,
                                                            const u64 *__restrict__ chunk_off, u64 base_bits, u32 *__restrict__ out)
{
    __shared__ u32 stab[256];
    __shared__ u32 wsum[kPackThreads / 64];
    stab[threadIdx.x] = table[threadIdx.x];
    __syncthreads();
    u64 val;
    u32 bits;
    pack4(src, n, (u64)blockIdx.x * kChunk + threadIdx.x * kBytesPerThread, stab, val, bits);
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
    if (!bits) return;
    const u64 pos = base_bits + chunk_off[blockIdx.x] + before;
    const u32 sh = (u32)(pos & 31u);
    u32 *dst = out + (pos >> 5);
    const u32 v0 = (u32)val, v1 = (u32)(val >> 32);
    const u32 o0 = v0 << sh;
    const u32 o1 = (sh ? v0 >> (32 - sh) : 0u) | (v1 << sh);
    const u32 o2 = sh ? v1 >> (32 - sh) : 0u;
    if (o0) atomicOr(dst, o0);
    if (o1) atomicOr(dst + 1, o1);
    if (o2) atomicOr(dst + 2, o2);
}

}  // namespace

u32 huffman_chunks(u64 n) { return (u32)((n + kChunk - 1) / kChunk); }

hipError_t launch_huffman_pack(const uint8_t *src, uint64_t n, const uint32_t *d_table, uint32_t *d_chunk_bits, uint64_t *d_chunk_off,
                               uint64_t *d_total, uint64_t base_bits, uint32_t *d_out, hipStream_t s)
{
    const u32 nchunks = huffman_chunks(n);
    if (nchunks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_huff_count, dim3(nchunks), dim3(kPackThreads), 0, s, src, n, d_table, d_chunk_bits);
    hipLaunchKernelGGL(k_huff_scan, dim3(1), dim3(1024), 0, s, d_chunk_bits, d_chunk_off, nchunks, d_total);
    hipLaunchKernelGGL(k_huff_pack, dim3(nchunks), dim3(kPackThreads), 0, s, src, n, d_table, d_chunk_off, base_bits, d_out);
    return hipGetLastError();
}

}  // namespace hgi
