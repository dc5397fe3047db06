// Internal launch interface between hgi_capi.hip (host logic) and hgi_kernels.hip (gfx950 kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "hgi_huffman_host.h"

namespace hgi {

constexpr int kInterpLeftTop = 0;
constexpr int kInterpCrossed = 1;

// Fused tile geometry (see DESIGN.md "Kernels").  Three builds of the same source live in the library:
// 128 x 64 tiles (throughput: batches, large frames), 128 x 32 tiles (latency: four times the waves and
// half the chain per wave when a call has too few tiles to fill the GPU) and 128 x 16 tiles (single small frames:
// the launch ends when its slowest wave does).  hgi_capi.hip picks per call.
constexpr int kTileW = 128;   // pixels per tile row: 8 lanes x 16 B = one 128-B line
constexpr int kFusedMaxLevels = 6;        // 2^k <= 64: deepest pyramid a 64-row tile can hold
constexpr int kFusedMaxLevelsSmall = 5;   // ... a 32-row tile
constexpr int kFusedMaxLevelsTiny = 4;    // ... and a 16-row tile
constexpr int kSeededMinLevels = 4;       // a seeded launch gives each lattice point of a tile's halo frame a lane: (128 >> k) + 2 by (64 >> k) + 2 <= 64
constexpr int kThreads = 64;  // ONE wave owns a tile: no workgroup barriers anywhere
// A plain encode of this many GiB per plane and more is dealt to the XCDs as contiguous eighths (hgi_fused_impl.h, xcd_mode()) and
// runs on 64-row tiles (hgi_capi.hip, use_tile_rows()); below, whole bands round-robin and 32-row tiles.  One number for both.
constexpr int kEncodeEighthsFromGiB = 4;

// 256-entry quantizer table passed BY VALUE in the kernarg segment: no device-side table to
// keep alive, nothing to synchronise, capturable.
struct Lut256 {
    uint32_t w[64];
};

struct Frames {
    uint32_t width, height;
    uint64_t frame_stride;  // bytes between frames
    uint32_t batch;
};

// What a tile launch of a pyramid deeper than its fused depth k starts from.
//   up == 0: compact planes holding the stride-2^k lattice, coded by earlier launches: rec = reconstructed values, q = grid
//            (residual) values (encode only), both sw x sh per frame, `stride` bytes apart.
//   up >= 1 (k == 4 only): the tile kernel rebuilds the `up` (<= 4) levels above the tile itself (hgi_fused_impl.h, cone_*).
//            rec == nullptr: the pyramid has 4 + up levels, the cone's base are the frame's own base samples.
//            rec != nullptr: the pyramid is deeper still; the planes hold the stride-2^(4 + up) lattice -- the cone's base.
struct Seeds {
    const uint8_t *rec;
    const uint8_t *q;
    uint32_t sw, sh;
    uint64_t stride;
    uint32_t up;
};

#ifdef HGI_TIMELINE
extern uint64_t *g_timeline;   // experiment builds only (tools/timeline.py): where the tile kernels log their blocks
#endif

// ---- level-wise path: one launch per level, straight global-memory stencil -----------------
hipError_t launch_seed(const uint8_t *src, uint8_t *dst, const Frames &f, uint32_t levels,
                       hipStream_t s);
hipError_t launch_decode_level(const uint8_t *grid, uint8_t *img, const Frames &f, uint32_t log2sub,
                               int interp, hipStream_t s);
hipError_t launch_encode_level(uint8_t *rec, uint8_t *grid, const Frames &f, uint32_t log2sub,
                               int interp, const Lut256 &lut, hipStream_t s);

// ---- fused path: the last k <= kFusedMaxLevels levels of every tile in one launch ------------
// (_64 / _32 = tile rows; k <= kFusedMaxLevels resp. kFusedMaxLevelsSmall; row_limit: pixel rows (multiple of 64)
// above which tile rows are launched, 0 = all -- bands of a frame that is still being uploaded; resident_tiles: tiles per CU a
// decode launch is held to, -1 = the library's policy (hgi_fused_impl.h) -- the placement probe asks for 0 = all the LDS allows)
#define HGI_DECLARE_FUSED(TH)                                                                                      \
    hipError_t launch_decode_fused_##TH(const uint8_t *grid, uint8_t *img, const Frames &f, uint32_t k, int interp, \
                                        const Seeds *seeds, hipStream_t s, uint32_t row_limit, int resident_tiles); \
    hipError_t launch_encode_fused_##TH(const uint8_t *img, uint8_t *grid, const Frames &f, uint32_t k, int interp, \
                                        const Lut256 &lut, bool lut_is_identity, const Seeds *seeds, hipStream_t s, \
                                        uint32_t row_limit);
HGI_DECLARE_FUSED(64)
HGI_DECLARE_FUSED(32)
HGI_DECLARE_FUSED(16)
#undef HGI_DECLARE_FUSED

// dst[f][j][i] = src[f][j << k][i << k]  (the stride-2^k lattice as a dense plane)
hipError_t launch_gather_lattice(const uint8_t *src, const Frames &f, uint32_t k, uint8_t *dst,
                                 uint32_t sw, uint32_t sh, uint64_t dst_stride, hipStream_t s);

// The whole upper pyramid (levels above the fused depth k) of every frame in ONE launch, one workgroup per frame, when
// the lattice plane is small enough (lattice_pyramid_fits): gathers the lattice from `src` (image when encoding, grid
// when decoding), codes `up` levels in LDS, writes the seed planes (out_q only when encoding).
bool lattice_pyramid_fits(uint32_t sw, uint32_t sh, size_t batch);
hipError_t launch_lattice_pyramid(const uint8_t *src, const Frames &f, uint32_t k, uint32_t up, int interp, const Lut256 &lut,
                                  bool lut_is_identity, bool encode, uint8_t *out_q, uint8_t *out_rec, uint32_t sw,
                                  uint32_t sh, uint64_t dst_stride, hipStream_t s);

// ---- harness kernels --------------------------------------------------------------------------
hipError_t launch_synth(int kind, uint64_t seed, uint64_t first_frame, uint8_t *out, const Frames &f,
                        hipStream_t s);
hipError_t launch_copy(const uint8_t *src, uint8_t *dst, size_t n, hipStream_t s);
// hist[f][v] = number of pixels of frame f equal to v (uint64); the entropy front end of SURVEY 8(f4)
hipError_t launch_histogram(const uint8_t *src, const Frames &f, unsigned long long *hist, hipStream_t s);
hipError_t launch_diff_stats(const uint8_t *a, const uint8_t *b, const Frames &f,
                             unsigned long long *out, hipStream_t s);

// ---- entropy stage (hgi_entropy.hip): raw DEFLATE of a grid as one dynamic-Huffman block of literals + run matches ----
// (symbol counts, the plan block and the host-side code construction: hgi_huffman_host.h, plain C++)
// device, pass 1, for `frames` grids `stride` bytes apart: d_hist[frame][kMatchThresholds + 1][kDeflateSymbols] = how often
// the tokens of the frame's n bytes use each symbol (runs -> distance-1 matches, cut at 1 KiB chunk boundaries): slot
// v < kMatchThresholds counts the tokens that depend on the candidate threshold v, slot kMatchThresholds those that do
// not -- the histogram of candidate v is the sum of the two
hipError_t launch_token_histogram(const uint8_t *src, uint64_t n, uint64_t stride, uint32_t frames, unsigned long long *d_hist,
                                  hipStream_t s);
// device, passes 2 + 3: each frame's complete stream (front, tokens, tail) written to d_outs + its plan's out_off, which
// need not be cleared (dist_code = reversed code | length << 24 of the one distance code); d_totals[frame] = the tokens'
// bits.  Scratch: frames * huffman_chunks(n) u32 + as many u64.
uint32_t huffman_chunks(uint64_t n);
hipError_t launch_huffman_pack(const uint8_t *src, uint64_t n, uint64_t stride, uint32_t frames, const void *d_plans, uint32_t dist_code,
                               uint32_t *d_chunk_bits, uint64_t *d_chunk_off, uint64_t *d_totals, uint8_t *d_outs, hipStream_t s);

}  // namespace hgi
