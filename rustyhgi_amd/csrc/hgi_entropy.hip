// Entropy stage on the device: raw DEFLATE (RFC 1951) of a residual grid as ONE dynamic-Huffman block of literals and
// run matches.
//
// The reference serialises a grid by handing its bincode image to flate2's DEFLATE at the best level
// (src/archive.rs:34-40), on the CPU, one thread: 8 ms for a 1920 x 1080 grid against 7 us for the encode that made it
// (profiles/r02_bench_cpp.txt), seconds for a 4096 x 4096 one.  Residual grids are noise around zero: what LZ77 finds in
// them is runs (of zeros, in smooth regions) and nothing else -- zlib's run-length-only strategy (Z_RLE) is within 1 % of
// its level 9 on them, and where there are no runs the plain Huffman code alone is SMALLER than level 9 (LENA / Medium:
// 14.0 against 16.0 kB with run matches, 14.5 with literals alone).  Both parallelise:
//   tokens     a byte equal to its predecessor continues a run; a run's bytes after its first are covered by matches of
//              distance 1 and length 3..258 (leftovers of 1-2 bytes stay literals).  Runs are cut at 1 KiB chunk
//              boundaries, so every token is decided inside one wave (cost: one extra literal per KiB of run).
//   histogram  of the 286 literal / length symbols the tokens use, for four match thresholds   (device, pass 1)
//   code       for each threshold: length-limited canonical Huffman code, exact stream size;    (host, a few hundred symbols)
//              the smallest wins (short runs of a byte whose literal costs 1 bit are cheaper as literals) + block header
//   bits       per chunk (pass 2), exclusive scan over the chunks, and every token OR-ed into place (pass 3)
// The stream is ordinary DEFLATE: flate2 / zlib / miniz inflate it; `Archive::deserialize_from_reader`
// (src/archive.rs:43-55) reads archives written this way unchanged.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hgi_kernels.h"
#include "hgi_knobs.h"

namespace hgi {

namespace {

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

}  // namespace

// (the host side -- code construction, block header, per-frame plan -- is plain C++ in hgi_huffman_host.h, so that it also
// builds and fuzzes under g++ -fsanitize=address,undefined: tests/test_sanitizers.py)

// ---------------------------------------------------------------------------------------------------------------
// device: tokens, histogram, sum, scan, pack
// ---------------------------------------------------------------------------------------------------------------
// One WAVE per 1 KiB chunk, sixteen bytes a lane: everything a chunk needs -- who heads a run, how long the run still
// goes -- is two shuffle scans inside the wave plus bit tricks on the lane's own 16-bit masks; no workgroup barrier
// exists in any of the loops.  A workgroup is four such waves that share the read-only tables (and, in pass 1, the
// histogram) in LDS and walk their own chunks, the next chunk's bytes already loading while this one is coded.
// grid.y = the frame, so one launch covers a whole batch.
namespace {

constexpr int kWaves = 4;
constexpr int kPackThreads = 64 * kWaves;
constexpr int kLaneBytes = 16;
constexpr int kChunk = 64 * kLaneBytes;         // bytes per wave pass: runs never cross a chunk boundary
constexpr u32 kNone = 0xFFFFFFFFu;
__device__ constexpr u32 kMatchThreshold[kMatchThresholds] = {3, 4, 6, 10};      // = kMatchThresholdHost (hgi_kernels.h)

// a wave's LDS instructions execute in order: lanes exchange data through LDS without a barrier, the compiler only has
// to keep the accesses where they are
#define WAVE_LDS_ORDER() asm volatile("" ::: "memory")

// where a frame's stream starts in the output buffer of the launch (the host lays the streams out: one slot per frame,
// or back to back for the packed entry point)
__device__ __forceinline__ u64 plan_out_off(const DeflatePlan &p) { return ((u64)p.out_off[1] << 32) | p.out_off[0]; }

// Wave-wide scans on the DPP lanes of the VALU (no LDS crossbar round trips): four row_shr / row_shl steps scan each row
// of sixteen lanes, then the rows are joined -- forward by the row broadcasts, backward through three scalar reads.
template <int kCtrl, int kRows = 0xF>
__device__ __forceinline__ u32 dpp(u32 fill, u32 v)      // lanes the pattern gives no source keep `fill`
{
    return (u32)__builtin_amdgcn_update_dpp((int)fill, (int)v, kCtrl, kRows, 0xF, false);
}
constexpr int kRowShr = 0x110, kRowShl = 0x100, kWaveShl1 = 0x130, kWaveShr1 = 0x138, kRowBcast15 = 0x142, kRowBcast31 = 0x143;

__device__ __forceinline__ u32 umax(u32 a, u32 b) { return a > b ? a : b; }
__device__ __forceinline__ u32 umin(u32 a, u32 b) { return a < b ? a : b; }

__device__ __forceinline__ u32 wave_scan_max(u32 v)      // inclusive, lanes 0 -> 63; identity 0
{
    v = umax(v, dpp<kRowShr + 1>(0u, v));
    v = umax(v, dpp<kRowShr + 2>(0u, v));
    v = umax(v, dpp<kRowShr + 4>(0u, v));
    v = umax(v, dpp<kRowShr + 8>(0u, v));
    v = umax(v, dpp<kRowBcast15, 0xA>(0u, v));
    v = umax(v, dpp<kRowBcast31, 0xC>(0u, v));
    return v;
}

__device__ __forceinline__ u32 wave_scan_add(u32 v)      // inclusive, lanes 0 -> 63
{
    v += dpp<kRowShr + 1>(0u, v);
    v += dpp<kRowShr + 2>(0u, v);
    v += dpp<kRowShr + 4>(0u, v);
    v += dpp<kRowShr + 8>(0u, v);
    v += dpp<kRowBcast15, 0xA>(0u, v);
    v += dpp<kRowBcast31, 0xC>(0u, v);
    return v;
}

__device__ __forceinline__ u32 wave_rscan_min(u32 v, u32 lane)      // inclusive, lanes 63 -> 0; identity 0xFFFFFFFF
{
    v = umin(v, dpp<kRowShl + 1>(0xFFFFFFFFu, v));
    v = umin(v, dpp<kRowShl + 2>(0xFFFFFFFFu, v));
    v = umin(v, dpp<kRowShl + 4>(0xFFFFFFFFu, v));
    v = umin(v, dpp<kRowShl + 8>(0xFFFFFFFFu, v));
    const u32 r1 = (u32)__builtin_amdgcn_readlane((int)v, 16), r2 = (u32)__builtin_amdgcn_readlane((int)v, 32),
              r3 = (u32)__builtin_amdgcn_readlane((int)v, 48);
    const u32 row = lane >> 4;
    const u32 later = row == 0 ? umin(r1, umin(r2, r3)) : row == 1 ? umin(r2, r3) : row == 2 ? r3 : 0xFFFFFFFFu;
    return umin(v, later);
}

struct Lane {
    u32 w[4];      // the lane's sixteen bytes, little endian
    u32 cnt;       // how many of them exist (16 except at the end of the data)
};

__device__ __forceinline__ Lane load_lane(const u8 *__restrict__ src, u64 n, u64 chunk0, u32 lane)
{
    Lane L;
    const u64 at = chunk0 + (u64)lane * kLaneBytes;
    L.w[0] = L.w[1] = L.w[2] = L.w[3] = 0;
    if (at + kLaneBytes <= n) {
        __builtin_memcpy(L.w, src + at, kLaneBytes);
        L.cnt = kLaneBytes;
    } else {
        L.cnt = at >= n ? 0u : (u32)(n - at);
#pragma unroll
        for (int i = 0; i < kLaneBytes; ++i)
            if ((u32)i < L.cnt) L.w[i >> 2] |= (u32)src[at + i] << (8 * (i & 3));
    }
    return L;
}

__device__ __forceinline__ u32 byte_of(const Lane &L, int i) { return (L.w[i >> 2] >> (8 * (i & 3))) & 255u; }

// Positions inside the chunk: p = 16 lane + i.  A byte CONTINUES when it equals its predecessor inside the chunk; the
// others HEAD a run (or stand alone).  For a continuing byte: head(p) = last head before it (exists: position 0),
// stop(p) = first position after it that does not continue (a head, or the end of the data).
struct Runs {
    u32 valid, cont, head, stop;      // 16-bit masks over the lane's bytes (stop = head | ~valid)
    u32 head_before;                  // 1 + position of the last head in earlier lanes (0: none, only for lane 0)
    u32 stop_after;                   // first stop in later lanes, or the number of valid bytes in the chunk
};

__device__ __forceinline__ Runs find_runs(const Lane &L, u32 lane, u32 chunk_valid)
{
    Runs R;
    R.valid = (1u << L.cnt) - 1u;
    u32 prev = dpp<kWaveShr1>(0u, L.w[3]) >> 24;      // the previous lane's last byte
    if (lane == 0) prev = 0x100u;                     // the chunk's first byte never continues
    u32 cont = 0;
#pragma unroll
    for (int i = 0; i < kLaneBytes; ++i) {
        const u32 b = byte_of(L, i);
        cont |= (u32)(b == prev) << i;
        prev = b;
    }
    R.cont = cont & R.valid;
    R.head = R.valid & ~R.cont;
    R.stop = (R.head | ~R.valid) & 0xFFFFu;
    const u32 fi = wave_scan_max(R.head ? kLaneBytes * lane + (31u - (u32)__clz(R.head)) + 1u : 0u);
    const u32 bi = wave_rscan_min(R.stop ? kLaneBytes * lane + (u32)__builtin_ctz(R.stop) : kNone, lane);
    R.head_before = dpp<kWaveShr1>(0u, fi);           // lane 0: none
    u32 sa = dpp<kWaveShl1>(kNone, bi);               // lane 63: none
    if (sa > chunk_valid) sa = chunk_valid;
    R.stop_after = sa;
    return R;
}

// The continuing bytes after a run's head are cut into PIECES of 258 (the longest DEFLATE match); a piece of at least
// min_match bytes is coded as one match (distance 1) at its first byte, the bytes of a shorter piece stay literals.  What a
// lane has to know about the pieces that touch its sixteen bytes:
struct Pieces {
    u32 starts;         // its bytes at which a piece starts (mask): a run's second byte, or 258 x m bytes further on
    u32 inh_cover;      // its leading bytes that belong to a piece started in an earlier lane (mask; 0: none) ...
    u32 inh_len;        // ... and that piece's whole length
};

__device__ __forceinline__ u32 mod258(u32 o /* <= 1022 */) { return o - 258u * ((u32)(o >= 258u) + (u32)(o >= 516u) + (u32)(o >= 774u)); }

__device__ __forceinline__ Pieces find_pieces(const Runs &R, u32 lane)
{
    Pieces P;
    const u32 p0 = kLaneBytes * lane;
    P.starts = R.cont & ~(R.cont << 1) & ~1u;                       // runs whose head is one of my bytes
    P.inh_cover = 0;
    P.inh_len = 0;
    const u32 lead = (u32)__builtin_ctz(~R.cont | 0x10000u);        // my leading bytes that continue a run from an earlier lane
    if (lead) {
        const u32 o0 = mod258(p0 - R.head_before);                  // byte 0's offset inside its piece (head at head_before - 1)
        const u32 to_next = o0 ? 258u - o0 : 0u;                    // bytes until the next piece starts
        if (to_next < lead) P.starts |= 1u << to_next;
        if (o0) {
            const u32 sp0 = lead < (u32)kLaneBytes ? p0 + lead : R.stop_after;
            const u32 len = sp0 - p0 + o0;
            P.inh_len = len < 258u ? len : 258u;
            P.inh_cover = (1u << (lead < to_next ? lead : to_next)) - 1u;
        }
    }
    return P;
}

// the piece that starts at my byte i: its length and which of my bytes it covers
__device__ __forceinline__ void piece_at(const Runs &R, u32 lane, u32 i, u32 &len, u32 &cover)
{
    const u32 p = kLaneBytes * lane + i;
    const u32 above = R.stop >> (i + 1u);
    const u32 sp = above ? p + 1u + (u32)__builtin_ctz(above) : R.stop_after;
    const u32 l = sp - p;
    len = l < 258u ? l : 258u;
    const u32 here = len < kLaneBytes - i ? len : kLaneBytes - i;
    cover = ((1u << here) - 1u) << i;
}

__device__ __forceinline__ u32 byte_at(const Lane &L, u32 i)      // i not known at compile time
{
    const u64 lo = ((u64)L.w[1] << 32) | L.w[0], hi = ((u64)L.w[3] << 32) | L.w[2];
    return (u32)((i < 8u ? lo : hi) >> (8u * (i & 7u))) & 255u;
}

// length 3..258 -> length symbol, extra bits (RFC 1951 3.2.5)
__device__ __forceinline__ void length_symbol(u32 length, u32 &symbol, u32 &extra_bits, u32 &extra)
{
    const u32 l = length - 3u;                        // 0..255
    const u32 e = l < 8u ? 0u : 29u - (u32)__clz(l);  // extra bits: l in [8,16) -> 1, [16,32) -> 2, ...
    symbol = l < 8u ? 257u + l : 257u + 4u * e + 4u + ((l >> e) & 3u);
    extra_bits = e;
    extra = l & ((1u << e) - 1u);
    if (length == 258u) {
        symbol = 285u;
        extra_bits = 0;
        extra = 0;
    }
}

// pass 1: histograms of the literal / length symbols the tokens use, for each candidate match threshold at once (a short
// run of a byte whose literal costs one bit is cheaper as literals than as a match; which threshold pays is decided on
// the host from the exact stream sizes the histograms imply).  A byte's token is the same under every candidate unless
// it lies in a piece of kMatchThreshold[0] .. kMatchThreshold[last] - 1 bytes: those go to a per-candidate histogram,
// everything else to ONE common histogram (slot kMatchThresholds) that the host adds to each.
__global__ __launch_bounds__(kPackThreads) void k_token_hist(const u8 *__restrict__ src, u64 n, u64 stride, u32 nchunks,
                                                             unsigned long long *__restrict__ hist)
{
    constexpr int kCopies = 4, kHists = kMatchThresholds + 1;
    constexpr u32 kLo = kMatchThreshold[0], kHi = kMatchThreshold[kMatchThresholds - 1];
    __shared__ u32 h[kHists][kDeflateSymbols * kCopies];      // copies value-major (a handful of symbols dominate)
    for (int i = threadIdx.x; i < kHists * kDeflateSymbols * kCopies; i += kPackThreads) (&h[0][0])[i] = 0;
    __syncthreads();
    src += (u64)blockIdx.y * stride;
    hist += (u64)blockIdx.y * (kHists * kDeflateSymbols);
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const u32 copy = lane & (kCopies - 1);
    u32 zeros = 0;                                            // literal zeros are most of a residual grid: counted privately
    auto add = [&](int which, u32 sym, u32 count) { atomicAdd(&h[which][sym * kCopies + copy], count); };
    // a piece of `len` bytes of value b, `mine` of them in this lane, `first`: it starts here
    auto piece = [&](u32 len, u32 b, u32 mine, bool first) {
        u32 sym, eb, ex;
        length_symbol(len, sym, eb, ex);
        if (len >= kHi) {
            if (first) add(kMatchThresholds, sym, 1u);
            return;
        }
#pragma unroll
        for (int v = 0; v < kMatchThresholds; ++v) {
            if (len >= kMatchThreshold[v]) {
                if (first) add(v, sym, 1u);
            } else {
                add(v, b, mine);
            }
        }
    };
    const u32 step = gridDim.x * kWaves;
    u32 chunk = blockIdx.x * kWaves + wave;
    Lane cur = {};
    if (chunk < nchunks) cur = load_lane(src, n, (u64)chunk * kChunk, lane);
    while (chunk < nchunks) {
        const u32 next = chunk + step;
        Lane nxt = {};
        if (next < nchunks) nxt = load_lane(src, n, (u64)next * kChunk, lane);
        const u64 left = n - (u64)chunk * kChunk;
        const Runs R = find_runs(cur, lane, left < (u64)kChunk ? (u32)left : (u32)kChunk);
        const Pieces P = find_pieces(R, lane);
        u32 covered = 0;                                      // my bytes in pieces of >= kLo: not literals under every candidate
        if (P.inh_len >= kLo) {
            covered = P.inh_cover;
            piece(P.inh_len, byte_of(cur, 0), (u32)__popc(P.inh_cover), false);
        }
        for (u32 s = P.starts; s; s &= s - 1u) {
            const u32 i = (u32)__builtin_ctz(s);
            u32 len, cover;
            piece_at(R, lane, i, len, cover);
            if (len < kLo) continue;
            covered |= cover;
            piece(len, byte_at(cur, i), (u32)__popc(cover), true);
        }
        const u32 lit = R.valid & ~covered;
#pragma unroll
        for (int i = 0; i < kLaneBytes; ++i) {
            if (!((lit >> i) & 1u)) continue;
            const u32 b = byte_of(cur, i);
            if (b == 0u)
                ++zeros;
            else
                add(kMatchThresholds, b, 1u);
        }
        cur = nxt;
        chunk = next;
    }
    if (zeros) atomicAdd(&h[kMatchThresholds][copy], zeros);
    __syncthreads();
    for (int i = threadIdx.x; i < kHists * kDeflateSymbols; i += kPackThreads) {
        const int v = i / kDeflateSymbols, sym = i - v * kDeflateSymbols;
        u32 sum = 0;
#pragma unroll
        for (int k = 0; k < kCopies; ++k) sum += h[v][sym * kCopies + k];
        if (sum) atomicAdd(&hist[i], (unsigned long long)sum);
    }
}

// a match of `len` bytes under the frame's code: its length code, the length's extra bits, the one distance code
// (<= 15 + 5 + 1 bits), as bits << 24 | code like the table's entries
__device__ __forceinline__ u32 match_token(u32 len, const u32 *stab, u32 dist)
{
    u32 sym, eb, ex;
    length_symbol(len, sym, eb, ex);
    const u32 e = stab[sym];
    u32 code = e & 0xFFFFFFu, nb = e >> 24;
    code |= ex << nb;
    nb += eb;
    code |= (dist & 0xFFFFFFu) << nb;
    nb += dist >> 24;
    return code | (nb << 24);
}

// pass 2: bits per chunk
__global__ __launch_bounds__(kPackThreads) void k_token_count(const u8 *__restrict__ src, u64 n, u64 stride, u32 nchunks,
                                                              const DeflatePlan *__restrict__ plans, u32 dist, u32 *__restrict__ chunk_bits)
{
    __shared__ u32 stab[kDeflateSymbols];
    const DeflatePlan &plan = plans[blockIdx.y];
    for (int i = threadIdx.x; i < kDeflateSymbols; i += kPackThreads) stab[i] = plan.table[i];
    const u32 min_match = plan.min_match;
    __syncthreads();
    src += (u64)blockIdx.y * stride;
    chunk_bits += (u64)blockIdx.y * nchunks;
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const u32 step = gridDim.x * kWaves;
    u32 chunk = blockIdx.x * kWaves + wave;
    Lane cur = {};
    if (chunk < nchunks) cur = load_lane(src, n, (u64)chunk * kChunk, lane);
    while (chunk < nchunks) {
        const u32 next = chunk + step;
        Lane nxt = {};
        if (next < nchunks) nxt = load_lane(src, n, (u64)next * kChunk, lane);
        const u64 left = n - (u64)chunk * kChunk;
        const Runs R = find_runs(cur, lane, left < (u64)kChunk ? (u32)left : (u32)kChunk);
        const Pieces P = find_pieces(R, lane);
        u32 covered = P.inh_len >= min_match ? P.inh_cover : 0u;
        u32 bits = 0;
        for (u32 s = P.starts; s; s &= s - 1u) {
            u32 len, cover;
            piece_at(R, lane, (u32)__builtin_ctz(s), len, cover);
            if (len < min_match) continue;
            covered |= cover;
            bits += match_token(len, stab, dist) >> 24;
        }
        const u32 lit = R.valid & ~covered;
#pragma unroll
        for (int i = 0; i < kLaneBytes; ++i) {
            const u32 e = stab[byte_of(cur, i)];
            bits += (lit >> i) & 1u ? e >> 24 : 0u;
        }
        bits = wave_scan_add(bits);
        if (lane == 63) chunk_bits[chunk] = bits;
        cur = nxt;
        chunk = next;
    }
}

// exclusive scan of a frame's chunk sizes into bit offsets (grid.x = the frame); one workgroup walks the chunks 1024 at a
// time (coalesced reads and writes; shuffle scan inside each wave, the sixteen wave totals through LDS, a running total
// carried between blocks).  It also clears the stream's words in which a chunk starts or the tokens end: those are the
// words two writers share in pass 3 (they OR into them); every other word has one owner who stores it whole, so the
// stream needs no clearing beyond this.
__global__ __launch_bounds__(1024) void k_huff_scan(const u32 *__restrict__ chunk_bits, u64 *__restrict__ chunk_off, u32 nchunks,
                                                    u64 *__restrict__ totals, const DeflatePlan *__restrict__ plans, u8 *__restrict__ outs)
{
    __shared__ u64 wtot[16];
    chunk_bits += (u64)blockIdx.x * nchunks;
    chunk_off += (u64)blockIdx.x * nchunks;
    u32 *out = reinterpret_cast<u32 *>(outs + plan_out_off(plans[blockIdx.x]));
    const u64 base = plans[blockIdx.x].base_bits;
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
            if (i < nchunks) {
                const u64 off = carry + before + incl - mine[k];
                chunk_off[i] = off;
                out[(base + off) >> 5] = 0;
            }
            carry += all;
            __syncthreads();            // wtot is rewritten by the next block
        }
    }
    if (threadIdx.x == 0) {
        totals[blockIdx.x] = carry;
        out[(base + carry) >> 5] = 0;
    }
}

// pass 3: a wave ORs its chunk's tokens into an LDS image of the chunk's part of the stream (<= 1024 x 15 bits), which
// then goes out as whole words -- plain coalesced stores for the words the chunk owns alone, atomicOr for its first word
// and, when the next chunk starts inside it, its last (k_huff_scan cleared exactly those).  The frame's first workgroup
// also writes what precedes the tokens, the wave that codes the last chunk what follows them: the stream leaves the
// device complete.
__global__ __launch_bounds__(kPackThreads) void k_token_pack(const u8 *__restrict__ src, u64 n, u64 stride, u32 nchunks,
                                                             const DeflatePlan *__restrict__ plans, u32 dist, const u64 *__restrict__ chunk_off,
                                                             u8 *__restrict__ outs)
{
    constexpr int kWords = (kChunk * 15 + 31) / 32 + 4;      // every byte a 15-bit literal (a match spends 21 bits on >= 3 bytes)
    constexpr int kMaxStarts = 9;                            // pieces that can start inside sixteen bytes
    __shared__ u32 stab[kDeflateSymbols];
    __shared__ u32 imgs[kWaves][kWords];
    __shared__ u32 mtoks[kWaves][kMaxStarts][64];            // a lane's match tokens, in the order its pieces start
    const DeflatePlan &plan = plans[blockIdx.y];
    for (int i = threadIdx.x; i < kDeflateSymbols; i += kPackThreads) stab[i] = plan.table[i];
    for (int i = threadIdx.x; i < kWaves * kWords; i += kPackThreads) (&imgs[0][0])[i] = 0;
    const u32 min_match = plan.min_match;
    const u64 base = plan.base_bits;
    __syncthreads();
    src += (u64)blockIdx.y * stride;
    chunk_off += (u64)blockIdx.y * nchunks;
    u32 *out = reinterpret_cast<u32 *>(outs + plan_out_off(plan));
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    u32 *img = imgs[wave];
    u32(*mtok)[64] = mtoks[wave];
    if (blockIdx.x == 0 && wave == 0) {
        // the front: whole words are this wave's alone; the word the tokens start in is shared with the first chunk
        const u32 own = (u32)(base >> 5);
        for (u32 w = lane; 4 * w < plan.front_bytes; w += 64) {
            u32 v;
            __builtin_memcpy(&v, plan.front + 4 * w, 4);
            if (w < own)
                out[w] = v;
            else if (v)
                atomicOr(out + w, v);
        }
    }
    const u32 step = gridDim.x * kWaves;
    u32 chunk = blockIdx.x * kWaves + wave;
    Lane cur = {};
    u64 off = 0;
    if (chunk < nchunks) {
        cur = load_lane(src, n, (u64)chunk * kChunk, lane);
        off = chunk_off[chunk];
    }
    while (chunk < nchunks) {
        const u32 next = chunk + step;
        Lane nxt = {};
        u64 off_next = 0;
        if (next < nchunks) {
            nxt = load_lane(src, n, (u64)next * kChunk, lane);
            off_next = chunk_off[next];
        }
        const u64 left = n - (u64)chunk * kChunk;
        const Runs R = find_runs(cur, lane, left < (u64)kChunk ? (u32)left : (u32)kChunk);
        const Pieces P = find_pieces(R, lane);
        // the pieces that become matches: their tokens into my column of mtok (slot = the piece's rank among my starts)
        u32 covered = P.inh_len >= min_match ? P.inh_cover : 0u, matches = 0, bits = 0;
        {
            u32 k = 0;
            for (u32 s = P.starts; s; s &= s - 1u, ++k) {
                const u32 i = (u32)__builtin_ctz(s);
                u32 len, cover;
                piece_at(R, lane, i, len, cover);
                if (len < min_match) continue;
                covered |= cover;
                matches |= 1u << i;
                const u32 t = match_token(len, stab, dist);
                mtok[k][lane] = t;
                bits += t >> 24;
            }
        }
        const u32 lit = R.valid & ~covered;
        u32 tok[kLaneBytes];
#pragma unroll
        for (int i = 0; i < kLaneBytes; ++i) {
            const u32 e = stab[byte_of(cur, i)];
            tok[i] = (lit >> i) & 1u ? e : 0u;
            bits += tok[i] >> 24;
        }
        const u32 incl = wave_scan_add(bits);
        const u32 chunk_total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        const u64 pos0 = base + off;                             // the chunk's first bit in the stream
        const u32 sh0 = (u32)(pos0 & 31u);                       // ... and where that is inside its first word
        WAVE_LDS_ORDER();
        {
            const u32 at = sh0 + incl - bits;
            u32 wi = at >> 5, nb = at & 31u;
            u64 acc = 0;
#pragma unroll
            for (int i = 0; i < kLaneBytes; ++i) {
                u32 t = tok[i];
                if ((matches >> i) & 1u) t = mtok[__popc(P.starts & ((1u << i) - 1u))][lane];
                if (!t) continue;
                acc |= (u64)(t & 0xFFFFFFu) << nb;
                nb += t >> 24;
                if (nb >= 32u) {
                    atomicOr(img + wi, (u32)acc);
                    ++wi;
                    acc >>= 32;
                    nb -= 32u;
                }
            }
            if (acc) atomicOr(img + wi, (u32)acc);
        }
        WAVE_LDS_ORDER();
        const u32 nwords = (sh0 + chunk_total + 31u) >> 5;
        const bool last_shared = ((sh0 + chunk_total) & 31u) != 0u;
        u32 *dst = out + (pos0 >> 5);
        for (u32 w = lane; w < nwords; w += 64) {
            const u32 v = img[w];
            img[w] = 0;
            if (w == 0 || (w == nwords - 1 && last_shared)) {
                if (v) atomicOr(dst + w, v);
            } else {
                dst[w] = v;
            }
        }
        WAVE_LDS_ORDER();
        if (chunk == nchunks - 1 && lane < 6) {
            // what follows the tokens: its first word is shared with the last chunk (or was cleared), the rest is new
            const u64 end = pos0 + chunk_total;
            const u32 sh = (u32)(end & 31u);
            const u32 hi = lane < 5 ? plan.tail[lane] : 0u, lo = lane ? plan.tail[lane - 1] : 0u;
            const u32 v = (u32)((((u64)hi << 32) | lo) >> (32u - sh));
            u32 *t = out + (end >> 5) + lane;
            if (lane == 0) {
                if (v) atomicOr(t, v);
            } else if (32u * lane < sh + plan.tail_bits) {
                *t = v;
            }
        }
        cur = nxt;
        off = off_next;
        chunk = next;
    }
}

// Workgroups per frame: all of a launch's workgroups resident at once (a second round would run at a fraction of the
// occupancy), and no more than there are chunks.  What fits is asked of the runtime once per kernel.
template <typename K>
u32 resident_workgroups(K kernel)
{
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kPackThreads, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount < 1) {
        (void)hipGetLastError();
        return 4u * 256u;
    }
    const int forced = HGI_KNOB(HGI_ENTROPY_WGS_PER_CU, 0);      // (knobs build: experiments)
    if (forced > 0) per_cu = forced;
    return (u32)per_cu * (u32)prop.multiProcessorCount;
}

u32 blocks_for(u32 nchunks, u32 frames, u32 resident)
{
    const u32 per_frame_max = (nchunks + kWaves - 1) / kWaves;
    u32 b = resident / frames;
    if (b > per_frame_max) b = per_frame_max;
    return b ? b : 1;
}

}  // namespace

u32 huffman_chunks(u64 n) { return (u32)((n + kChunk - 1) / kChunk); }

hipError_t launch_token_histogram(const uint8_t *src, uint64_t n, uint64_t stride, uint32_t frames, unsigned long long *d_hist, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(d_hist, 0, (size_t)frames * (kMatchThresholds + 1) * kDeflateSymbols * sizeof(unsigned long long), s);
    if (e != hipSuccess || n == 0 || frames == 0) return e;
    const u32 nchunks = huffman_chunks(n);
    static const u32 resident = resident_workgroups(k_token_hist);
    hipLaunchKernelGGL(k_token_hist, dim3(blocks_for(nchunks, frames, resident), frames), dim3(kPackThreads), 0, s, src, n, stride, nchunks, d_hist);
    return hipGetLastError();
}

hipError_t launch_huffman_pack(const uint8_t *src, uint64_t n, uint64_t stride, uint32_t frames, const void *d_plans, uint32_t dist_code,
                               uint32_t *d_chunk_bits, uint64_t *d_chunk_off, uint64_t *d_totals, uint8_t *d_outs, hipStream_t s)
{
    const u32 nchunks = huffman_chunks(n);
    if (nchunks == 0 || frames == 0) return hipSuccess;
    const DeflatePlan *plans = static_cast<const DeflatePlan *>(d_plans);
    static const u32 resident_count = resident_workgroups(k_token_count), resident_pack = resident_workgroups(k_token_pack);
    const dim3 grid_count(blocks_for(nchunks, frames, resident_count), frames), grid(blocks_for(nchunks, frames, resident_pack), frames);
    hipLaunchKernelGGL(k_token_count, grid_count, dim3(kPackThreads), 0, s, src, n, stride, nchunks, plans, dist_code, d_chunk_bits);
    hipLaunchKernelGGL(k_huff_scan, dim3(frames), dim3(1024), 0, s, d_chunk_bits, d_chunk_off, nchunks, d_totals, plans, d_outs);
    hipLaunchKernelGGL(k_token_pack, grid, dim3(kPackThreads), 0, s, src, n, stride, nchunks, plans, dist_code, d_chunk_off, d_outs);
    return hipGetLastError();
}

}  // namespace hgi
