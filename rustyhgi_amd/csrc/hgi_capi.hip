// C ABI of libhgi_hip.so (include/hgi.h), part 1: contexts, argument checking, scratch management, level scheduling, the
// device / host / host-batch / banded entry points of the codec, harness helpers.  (Entropy stage: hgi_entropy_host.hip;
// plane placement: hgi_planes.hip.)  No CPU implementation lives here: every encode/decode goes to the gfx950 kernels or fails.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <cstring>
#include <thread>
#include <vector>

#include "hgi_host.h"

using namespace hgi;
using namespace hgi::host;

namespace {
thread_local char g_err[512] = "";
}

namespace hgi {
namespace host {

hgi_status fail(hgi_status st, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return st;
}

}  // namespace host
}  // namespace hgi

namespace {

// Compile-time constants of the release library (hgi_knobs.h: the knobs build reads the same names from the environment)
#ifndef HGI_ENC_L1_TILE_ROWS
#define HGI_ENC_L1_TILE_ROWS 16    // tile rows of a plain one-level encode (the finest pass alone) on large calls: use_tile_rows()
#endif
#ifndef HGI_TILE16_MAX
#define HGI_TILE16_MAX 600    // an ENCODE of at most this many 32-row tiles runs on 16-row tiles instead (profiles/r03_sizes.txt: 1920 x 1080 is 510)
#endif
// The shallowest pyramid that runs as four fused levels + cone (split_pyramid()).  Six: a lone frame then keeps the small tiles
// and the short chain of a four-level launch (1920 x 1080 level 6: 10.6 / 6.5 -> 7.3 / 5.6 us), a batch is unchanged
// (64 x 4096^2: +0.4 / -0.8 %); at five levels nothing is gained in either (profiles/r03_cone_levels.txt).
constexpr uint32_t kConeMinLevels = 6;

}  // namespace

namespace hgi {
namespace host {

// Scratch is a bump allocator over one device buffer; it only grows between calls.
hgi_status ws_ensure(hgi_ctx *c, size_t bytes)
{
    if (bytes <= c->ws_bytes) return HGI_OK;
    // growing means freeing memory that queued work may still use
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->ws) HIP_TRY(hipFree(c->ws));
    c->ws = nullptr;
    c->ws_bytes = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->ws), bytes));
    c->ws_bytes = bytes;
    return HGI_OK;
}

uint8_t *ws_take(hgi_ctx *c, size_t bytes)
{
    size_t off = align_up(c->ws_used, 256);
    if (off + bytes > c->ws_bytes) return nullptr;
    c->ws_used = off + bytes;
    return c->ws + off;
}

struct SubGeom {
    uint32_t sw, sh;
    uint64_t stride;
};

SubGeom sub_geom(uint32_t w, uint32_t h, uint32_t k)
{
    SubGeom g;
    g.sw = (uint32_t)((((uint64_t)w - 1) >> k) + 1);
    g.sh = (uint32_t)((((uint64_t)h - 1) >> k) + 1);
    g.stride = align_up((size_t)g.sw * g.sh, 256);
    return g;
}

// How a pyramid is split between the tile kernel and what has to be coded in front of it.
//   levels <= 5: the tile holds the pyramid: k = levels, nothing else.
//   6 ... 8:     ONE launch at four fused levels that rebuilds the 2 ... 4 levels above a tile for itself (hgi_fused_impl.h,
//                cone_*): k = 4, up = levels - 4.  No lattice planes, no scratch, no launch in front.
//   deeper:      k = 4, up = 4, and the stride-256 lattice -- an HGI image with levels - 8 levels of its own (same OOB rule:
//                x < W <=> x >> 8 < ceil(W / 256)) -- is coded first; its planes are the cone's base.
// (Frames coded band by band from host memory split differently -- six fused levels on seed planes of the stride-64 lattice:
// host_banded() says why.  Every split gives the same bytes.)
struct Split {
    uint32_t k, up, shift;      // fused levels, cone levels, log2 of the lattice coded in front (0: none)
};

Split split_pyramid(uint32_t levels)
{
    if (levels < kConeMinLevels) return {levels, 0u, 0u};
    if (levels <= 8u) return {4u, levels - 4u, 0u};
    return {4u, 4u, 8u};
}

// Scratch bytes one encode (or decode) of this shape takes, recursion included.  The bump allocator only resets
// between calls, so this mirrors encode_impl / decode_impl plane for plane: a decode with a lattice in front takes two
// planes per recursion level; an encode takes three -- and then both encodes AND decodes its lattice (the
// reconstruction is what the tile kernel starts from), each of which recurses on its own.
size_t plane_bytes(const SubGeom &g, size_t batch) { return align_up(batch * g.stride, 256) + 256; }

size_t ws_need_decode(uint32_t w, uint32_t h, uint32_t levels, size_t batch)
{
    const Split sp = split_pyramid(levels);
    if (!sp.shift) return 0;
    const SubGeom g = sub_geom(w, h, sp.shift);
    return 2 * plane_bytes(g, batch) + ws_need_decode(g.sw, g.sh, levels - sp.shift, batch);
}

size_t ws_need_encode(uint32_t w, uint32_t h, uint32_t levels, size_t batch)
{
    const Split sp = split_pyramid(levels);
    if (!sp.shift) return 0;
    const SubGeom g = sub_geom(w, h, sp.shift);
    return 3 * plane_bytes(g, batch) + ws_need_encode(g.sw, g.sh, levels - sp.shift, batch) + ws_need_decode(g.sw, g.sh, levels - sp.shift, batch);
}

size_t ws_need(const hgi_ctx *c, uint32_t w, uint32_t h, uint32_t levels, size_t batch, size_t stride)
{
    if (levels == 0) return 0;
    if (c->path == HGI_PATH_LEVELWISE) return align_up(batch * stride, 256) + 256;
    const size_t e = ws_need_encode(w, h, levels, batch), d = ws_need_decode(w, h, levels, batch);
    return e > d ? e : d;
}

// Tile geometry of a fused launch.  128 x 64 tiles are the throughput shape; a call whose 64-row tiles would not
// fill the GPU (256 CUs x 20-32 resident waves) finishes sooner with 128 x 32 tiles -- four times the waves, half
// the dependent chain per wave.  Measured crossover on MI355X at level 4 (tools/size_sweep.py): equal at ~2000
// tiles; 32-row ahead by 15-35 % below ~1200, 64-row ahead by 12 % at 4000.  A single small frame (up to about two
// waves per CU of 32-row tiles) ends when its slowest wave does, and that wave's chain is mostly its own VALU work:
// 128 x 16 tiles halve the finest level's share of it (profiles/r03_sizes.txt).
// Round 4: a PLAIN encode (the tile holds the pyramid: no seeds) keeps the 32-row tiles on large calls too, on rows up to 8 192
// pixels and below four GiB per plane (kEncodeEighthsFromGiB).  The encoder needs its 20 resident tiles per CU (fewer: -2 ... -6 %), and with 64-row tiles
// those read 5.5 MB at a time per XCD -- more than its 4 MB L2; with 32-row tiles it is half (the decoder gets the same effect
// from holding ten 64-row tiles, hgi_fused_impl.h launch_decode_fused).  Measured, one process per setting on the knobs build
// (profiles/r04_enc_tile_rows.txt): 64 x 4096^2 level 4 356.0 -> 341.9 us (-4.0 %), 128 x -3.3 %, 256 x -2.7 %, levels 1 / 2 / 5
// -2.6 / -3.7 / -5.6 %, 16 x 8192^2 -1.5 %, 1 / 4 x 4096^2 -9 / -6 %, 16 x 1920 x 1080 -10 %; no change at 2 and 8 x 4096^2.  Not on
// 16384-wide rows (+7 %), not for encodes that rebuild levels in the kernel (the cone: +12 ... +19 %), and not from four GiB per
// plane, where the launch is dealt to the XCDs as contiguous eighths and the 64-row tiles stay ahead on every box sampled
// (2.65-2.78 against 2.73-2.80 ms per 512 frames; 1.329 against 1.340 ms per 256, 1.661 against 1.675 per 320, 80 x 8192^2 1.673
// against 1.732: profiles/r04_mid_sizes.txt).
// (Knobs build: HGI_TILE_H = 16 | 32 | 64 forces one where the pyramid fits -- the test suite runs every geometry on every
// shape; HGI_TILE16_MAX moves the lower crossover.)
uint32_t use_tile_rows(uint32_t w, uint32_t h, uint32_t k, size_t batch, bool encode, bool plain)
{
    const int forced = HGI_KNOB(HGI_TILE_H, 0);
    const uint64_t tiny_max = (uint64_t)HGI_KNOB(HGI_TILE16_MAX, HGI_TILE16_MAX);
    if (k > (uint32_t)kFusedMaxLevelsSmall) return 64;
    const bool fits16 = k <= (uint32_t)kFusedMaxLevelsTiny;
    if (forced == 16 && fits16) return 16;
    if (forced == 32 || (forced == 16 && !fits16)) return 32;
    if (forced == 64) return 64;
    const uint64_t tx = (w + kTileW - 1) / kTileW;
    const uint64_t tiles64 = tx * ((h + 63) / 64) * batch, tiles32 = tx * ((h + 31) / 32) * batch;
    if (encode && fits16 && tiles32 <= tiny_max) return 16;      // (decode sits on the launch floor with 32-row tiles already)
    if (tiles64 < 1536) return 32;
    // one level (the finest pass alone, P_fine): a tile has next to no halo there, and still smaller tiles keep still less in
    // flight: 16 rows, at any size (64 x 4096^2: 339 -> 333 us; 512 x: 2.69 ms whatever the planes' classes, where the 64-row
    // tiles dealt as eighths give 2.64 ... 2.89; profiles/r04_pfine_sweep.txt)
    if (encode && plain && k == 1 && w <= 8192) return HGI_ENC_L1_TILE_ROWS;
    // (the same measure as xcd_mode(): the bytes of the 64-row tiles that lie entirely inside the frames)
    const uint64_t interior64 = (uint64_t)(w / kTileW) * (h / 64) * batch * kTileW * 64;
    if (encode && plain && w <= 8192 && interior64 < ((uint64_t)kEncodeEighthsFromGiB << 30)) return 32;
    return 64;
}

// Pyramids deeper than eight levels: the one-workgroup-per-frame kernel codes the lattice plane when it is small (at most
// 8 192 points per frame: hgi_kernels.hip); a larger plane is an image in its own right and goes through encode_impl /
// decode_impl.  (Knobs build: HGI_NO_LATTICE_KERNEL sends small planes down the second route too, so that the test suite
// reaches it without a 23 000 x 23 000 frame.)
bool use_lattice_kernel(const SubGeom &g, size_t batch)
{
    return !HGI_SWITCH(HGI_NO_LATTICE_KERNEL) && lattice_pyramid_fits(g.sw, g.sh, batch);
}

hipError_t launch_encode_fused(const uint8_t *img, uint8_t *grid, const Frames &f, uint32_t k, int interp,
                               const Lut256 &lut, bool ident, const Seeds *seeds, hipStream_t s, uint32_t row_limit = 0)
{
    const uint32_t rows = row_limit && row_limit < f.height ? row_limit : f.height;   // what this launch really covers
    switch (use_tile_rows(f.width, rows, k, f.batch, true, seeds == nullptr)) {
    case 16: return launch_encode_fused_16(img, grid, f, k, interp, lut, ident, seeds, s, row_limit);
    case 32: return launch_encode_fused_32(img, grid, f, k, interp, lut, ident, seeds, s, row_limit);
    default: return launch_encode_fused_64(img, grid, f, k, interp, lut, ident, seeds, s, row_limit);
    }
}

hipError_t launch_decode_fused(const uint8_t *grid, uint8_t *img, const Frames &f, uint32_t k, int interp,
                               const Seeds *seeds, hipStream_t s, uint32_t row_limit = 0, int resident_tiles = -1)
{
    const uint32_t rows = row_limit && row_limit < f.height ? row_limit : f.height;
    switch (use_tile_rows(f.width, rows, k, f.batch, false, seeds == nullptr)) {
    case 16: return launch_decode_fused_16(grid, img, f, k, interp, seeds, s, row_limit, resident_tiles);
    case 32: return launch_decode_fused_32(grid, img, f, k, interp, seeds, s, row_limit, resident_tiles);
    default: return launch_decode_fused_64(grid, img, f, k, interp, seeds, s, row_limit, resident_tiles);
    }
}

Lut256 pack_lut(const uint8_t lut[256])
{
    Lut256 l;
    memcpy(l.w, lut, 256);
    return l;
}

bool is_identity(const uint8_t lut[256])
{
    for (int i = 0; i < 256; ++i)
        if (lut[i] != i) return false;
    return true;
}

hgi_status encode_impl(hgi_ctx *c, const uint8_t *img, uint32_t w, uint32_t h, uint32_t levels, int interp,
                       const uint8_t lut[256], uint8_t *grid, size_t batch, size_t stride)
{
    Frames f = {w, h, (uint64_t)stride, (uint32_t)batch};
    if (levels == 0) {   // src/encoder.rs:28-36 with step 1: the grid is the image
        for (size_t b = 0; b < batch; ++b)
            HIP_TRY(launch_copy(img + b * stride, grid + b * stride, (size_t)w * h, c->stream));
        return HGI_OK;
    }
    Lut256 l = pack_lut(lut);
    if (c->path == HGI_PATH_LEVELWISE) {
        uint8_t *rec = ws_take(c, batch * stride);
        if (!rec) return fail(HGI_ENOMEM, "scratch exhausted (level-wise reconstruction plane)");
        HIP_TRY(launch_copy(img, rec, batch * stride, c->stream));
        HIP_TRY(launch_seed(img, grid, f, levels, c->stream));
        for (uint32_t level = 0; level < levels; ++level)   // src/encoder.rs:45, sequential
            HIP_TRY(launch_encode_level(rec, grid, f, levels - level - 1, interp, l, c->stream));
        return HGI_OK;
    }
    const Split sp = split_pyramid(levels);
    if (sp.shift) {
        // the lattice = 0 (mod 2^shift) first: its reconstruction and residuals are what the tile launch starts from
        const SubGeom g = sub_geom(w, h, sp.shift);
        uint8_t *sub_img = ws_take(c, batch * g.stride);
        uint8_t *sub_grid = ws_take(c, batch * g.stride);
        uint8_t *sub_rec = ws_take(c, batch * g.stride);
        if (!sub_img || !sub_grid || !sub_rec) return fail(HGI_ENOMEM, "scratch exhausted (lattice planes)");
        if (use_lattice_kernel(g, batch)) {   // small planes: gather + all upper levels + both planes in one launch
            HIP_TRY(launch_lattice_pyramid(img, f, sp.shift, levels - sp.shift, interp, l, is_identity(lut), true, sub_grid, sub_rec,
                                           g.sw, g.sh, g.stride, c->stream));
        } else {                              // an image in its own right: gather it, code it, decode it (each may recurse)
            HIP_TRY(launch_gather_lattice(img, f, sp.shift, sub_img, g.sw, g.sh, g.stride, c->stream));
            HGI_TRY(encode_impl(c, sub_img, g.sw, g.sh, levels - sp.shift, interp, lut, sub_grid, batch, g.stride));
            HGI_TRY(decode_impl(c, sub_grid, g.sw, g.sh, levels - sp.shift, interp, sub_rec, batch, g.stride));
        }
        const Seeds sd = {sub_rec, sub_grid, g.sw, g.sh, g.stride, sp.up};
        HIP_TRY(launch_encode_fused(img, grid, f, sp.k, interp, l, is_identity(lut), &sd, c->stream));
    } else if (sp.up) {
        const Seeds sd = {nullptr, nullptr, 0, 0, 0, sp.up};
        HIP_TRY(launch_encode_fused(img, grid, f, sp.k, interp, l, is_identity(lut), &sd, c->stream));
    } else {
        HIP_TRY(launch_encode_fused(img, grid, f, sp.k, interp, l, is_identity(lut), nullptr, c->stream));
    }
    return HGI_OK;
}

hgi_status decode_impl(hgi_ctx *c, const uint8_t *grid, uint32_t w, uint32_t h, uint32_t levels, int interp,
                       uint8_t *img, size_t batch, size_t stride)
{
    Frames f = {w, h, (uint64_t)stride, (uint32_t)batch};
    if (levels == 0) {
        for (size_t b = 0; b < batch; ++b)
            HIP_TRY(launch_copy(grid + b * stride, img + b * stride, (size_t)w * h, c->stream));
        return HGI_OK;
    }
    if (c->path == HGI_PATH_LEVELWISE) {
        HIP_TRY(launch_seed(grid, img, f, levels, c->stream));   // src/decoder.rs:22-28
        for (uint32_t level = 0; level < levels; ++level)        // src/decoder.rs:30
            HIP_TRY(launch_decode_level(grid, img, f, levels - level - 1, interp, c->stream));
        return HGI_OK;
    }
    const Split sp = split_pyramid(levels);
    if (sp.shift) {
        const SubGeom g = sub_geom(w, h, sp.shift);
        uint8_t *sub_grid = ws_take(c, batch * g.stride);
        uint8_t *sub_rec = ws_take(c, batch * g.stride);
        if (!sub_grid || !sub_rec) return fail(HGI_ENOMEM, "scratch exhausted (lattice planes)");
        if (use_lattice_kernel(g, batch)) {
            HIP_TRY(launch_lattice_pyramid(grid, f, sp.shift, levels - sp.shift, interp, Lut256{}, true, false, nullptr, sub_rec, g.sw,
                                           g.sh, g.stride, c->stream));
        } else {
            HIP_TRY(launch_gather_lattice(grid, f, sp.shift, sub_grid, g.sw, g.sh, g.stride, c->stream));
            HGI_TRY(decode_impl(c, sub_grid, g.sw, g.sh, levels - sp.shift, interp, sub_rec, batch, g.stride));
        }
        const Seeds sd = {sub_rec, nullptr, g.sw, g.sh, g.stride, sp.up};
        HIP_TRY(launch_decode_fused(grid, img, f, sp.k, interp, &sd, c->stream));
    } else if (sp.up) {
        const Seeds sd = {nullptr, nullptr, 0, 0, 0, sp.up};
        HIP_TRY(launch_decode_fused(grid, img, f, sp.k, interp, &sd, c->stream));
    } else {
        HIP_TRY(launch_decode_fused(grid, img, f, sp.k, interp, nullptr, c->stream, 0, c->probe_resident_tiles));
    }
    return HGI_OK;
}

hgi_status check_common(hgi_ctx *c, const void *a, const void *b, uint32_t levels, int interp, size_t batch,
                        size_t stride, uint32_t w, uint32_t h)
{
    if (!c) return fail(HGI_EINVAL, "ctx is NULL");
    if (levels > 31) return fail(HGI_EINVAL, "levels %u out of range 0..=31", levels);
    if (interp != HGI_INTERP_LEFTTOP && interp != HGI_INTERP_CROSSED)
        return fail(HGI_EUNSUPPORTED, "interpolator %d not implemented (0 = LeftTop, 1 = Crossed)", interp);
    if (w == 0 || h == 0 || batch == 0) return HGI_OK;
    if (!a || !b) return fail(HGI_EINVAL, "NULL buffer");
    if (batch > 1 && stride < (size_t)w * h) return fail(HGI_EINVAL, "frame_stride %zu < width*height", stride);
    if (batch > 0x7fffffffu) return fail(HGI_EINVAL, "batch too large");
    // No input frame may share a byte with an output frame: the kernels read a tile's halo from frames the neighbouring
    // tiles are writing.  (The reference consumes its input by value, src/encoder.rs:39: aliasing cannot happen there.)
    // Frames are w*h bytes every `stride` bytes, so two frame trains that interleave inside one allocation (in = base,
    // out = base + w*h, stride = 2*w*h) are fine; what is refused is a pair of frames i, j with
    // |pa + i*stride - pb - j*stride| < w*h.  Host and device pointers are compared alike -- under unified addressing
    // they share one space.
    const uintptr_t pa = reinterpret_cast<uintptr_t>(a), pb = reinterpret_cast<uintptr_t>(b);
    const uintptr_t n = (uintptr_t)w * h, delta = pa > pb ? pa - pb : pb - pa;
    bool overlap;
    if (batch == 1) {
        overlap = delta < n;
    } else {
        // the frame-index difference m that brings the trains closest is delta / stride or one more
        const uintptr_t q = delta / stride, r = delta % stride;
        overlap = (q <= batch - 1 && r < n) || (q + 1 <= batch - 1 && stride - r < n);
    }
    if (overlap)
        return fail(HGI_EINVAL, "an input frame and an output frame overlap (%zu bytes each, %zu apart): they must not alias",
                    (size_t)n, (size_t)delta);
    return HGI_OK;
}

// Host-pointer calls on ONE frame (hgi_encode_u8 / hgi_decode_u8): are they banded, and how much scratch do they take?  One
// formula for host_roundtrip / host_banded AND hgi_ctx_reserve, so that a reserved ctx really does not re-allocate.
bool host_call_is_banded(const hgi_ctx *c, uint32_t w, uint32_t h, uint32_t levels)
{
    // large frames with a pyramid at least one level deep: band the frame so that its upload and download overlap
    return (size_t)w * h >= (4u << 20) && levels >= 1 && c->path != HGI_PATH_LEVELWISE && h >= 256 && !HGI_SWITCH(HGI_NO_BANDS);
}

// banded calls code pyramids deeper than a tile at six fused levels on seed planes of the stride-64 lattice (host_banded)
size_t banded_lattice_need(uint32_t w, uint32_t h, uint32_t levels)
{
    const uint32_t k = levels < (uint32_t)kFusedMaxLevels ? levels : (uint32_t)kFusedMaxLevels;
    if (levels <= k) return 0;
    const SubGeom g = sub_geom(w, h, k);
    return 3 * plane_bytes(g, 1) + ws_need_encode(g.sw, g.sh, levels - k, 1) + ws_need_decode(g.sw, g.sh, levels - k, 1);
}

size_t host_call_need(const hgi_ctx *c, uint32_t w, uint32_t h, uint32_t levels)
{
    const size_t n = (size_t)w * h, slot = align_up(n, 256) + 256;      // two staging slots: the frame in, the frame out
    if (host_call_is_banded(c, w, h, levels)) return 2 * slot + banded_lattice_need(w, h, levels) + 1024;
    return 2 * slot + ws_need(c, w, h, levels, 1, n);
}

hgi_status pipe_ensure(hgi_ctx *c)
{
    if (c->have_pipe) return HGI_OK;
    bool ok = hipStreamCreateWithFlags(&c->pipe[0], hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&c->pipe[1], hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; i < 3 && ok; ++i)
        ok = hipEventCreateWithFlags(&c->ev_up[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_free[i], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 16 && ok; ++i) ok = hipEventCreateWithFlags(&c->ev_band[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) return fail(HGI_EDEVICE, "stream/event creation failed");   // (a partial set is reclaimed with the process)
    c->have_pipe = true;
    return HGI_OK;
}

hgi_status pin_ensure(hgi_ctx *c, size_t bytes)
{
    if (bytes <= c->pin_bytes) return HGI_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->pin) HIP_TRY(hipHostFree(c->pin));
    c->pin = nullptr;
    c->pin_bytes = 0;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c->pin), bytes, hipHostMallocDefault));
    c->pin_bytes = bytes;
    return HGI_OK;
}

}  // namespace host
}  // namespace hgi

extern "C" {

const char *hgi_last_error(void) { return g_err; }
#ifdef HGI_KNOBS_ENV
const char *hgi_version(void) { return "hgi-hip 0.2.0 (gfx950; KNOBS build: tuning constants and test switches from the environment)"; }
#else
const char *hgi_version(void) { return "hgi-hip 0.2.0 (gfx950)"; }
#endif

hgi_status hgi_ctx_create(int device, hgi_ctx **out)
{
    if (!out) return fail(HGI_EINVAL, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(HGI_EDEVICE, "no usable HIP device (%s); this library has no CPU path",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= n) return fail(HGI_EDEVICE, "device %d not in 0..%d", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    hgi_ctx *c = new (std::nothrow) hgi_ctx();
    if (!c) return fail(HGI_ENOMEM, "host allocation failed");
    c->device = device;
    c->path = HGI_PATH_FUSED;
    c->ws = nullptr;
    c->ws_bytes = c->ws_used = 0;
    c->have_pipe = false;
    c->pin = nullptr;
    c->pin_bytes = 0;
    c->probe_resident_tiles = -1;
    c->planes_report[0] = 0;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_hist[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_hist[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&c->ev_probe[0]) != hipSuccess || hipEventCreate(&c->ev_probe[1]) != hipSuccess) {
        delete c;      // (a partial set is reclaimed with the process)
        return fail(HGI_EDEVICE, "stream/event creation failed");
    }
    c->stream = c->own_stream;
    *out = c;
    return HGI_OK;
}

void hgi_ctx_destroy(hgi_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->ws) (void)hipFree(c->ws);
    if (c->pin) (void)hipHostFree(c->pin);
    (void)hipEventDestroy(c->ev0);
    (void)hipEventDestroy(c->ev1);
    for (int i = 0; i < 2; ++i) {
        (void)hipEventDestroy(c->ev_hist[i]);
        (void)hipEventDestroy(c->ev_probe[i]);
    }
    (void)hipStreamDestroy(c->own_stream);
    if (c->have_pipe) {
        (void)hipStreamDestroy(c->pipe[0]);
        (void)hipStreamDestroy(c->pipe[1]);
        for (int i = 0; i < 3; ++i) {
            (void)hipEventDestroy(c->ev_up[i]);
            (void)hipEventDestroy(c->ev_free[i]);
        }
        for (int i = 0; i < 16; ++i) (void)hipEventDestroy(c->ev_band[i]);
    }
    delete c;
}

hgi_status hgi_ctx_set_stream(hgi_ctx *c, void *hip_stream)
{
    if (!c) return fail(HGI_EINVAL, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));   // scratch hand-over between streams
    c->stream = static_cast<hipStream_t>(hip_stream);
    return HGI_OK;
}

hgi_status hgi_ctx_use_own_stream(hgi_ctx *c)
{
    if (!c) return fail(HGI_EINVAL, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = c->own_stream;
    return HGI_OK;
}

hgi_status hgi_ctx_set_path(hgi_ctx *c, hgi_path path)
{
    if (!c) return fail(HGI_EINVAL, "ctx is NULL");
    if (path != HGI_PATH_AUTO && path != HGI_PATH_LEVELWISE && path != HGI_PATH_FUSED)
        return fail(HGI_EINVAL, "unknown path %d", (int)path);
    c->path = path == HGI_PATH_AUTO ? HGI_PATH_FUSED : path;
    return HGI_OK;
}

hgi_status hgi_ctx_reserve(hgi_ctx *c, uint32_t w, uint32_t h, uint32_t levels, size_t batch)
{
    if (!c) return fail(HGI_EINVAL, "ctx is NULL");
    if (levels > 31) return fail(HGI_EINVAL, "levels %u out of range 0..=31", levels);
    HIP_TRY(hipSetDevice(c->device));
    // the device-pointer calls on a batch of this shape, or the host-pointer calls on one frame of it, whichever takes more
    const size_t dev = ws_need(c, w, h, levels, batch, (size_t)w * h), host = host_call_need(c, w, h, levels);
    return ws_ensure(c, dev > host ? dev : host);
}

hgi_status hgi_ctx_scratch_bytes(hgi_ctx *c, size_t *bytes)
{
    if (!c || !bytes) return fail(HGI_EINVAL, "NULL argument");
    *bytes = c->ws_bytes;
    return HGI_OK;
}

hgi_status hgi_sync(hgi_ctx *c)
{
    if (!c) return fail(HGI_EINVAL, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return HGI_OK;
}

// src/quantizator.rs:41-63
hgi_status hgi_linear_lut(int level, uint8_t lut[256], uint8_t *max_err)
{
    static const uint8_t errs[4] = {0, 10, 20, 30};
    if (!lut) return fail(HGI_EINVAL, "lut is NULL");
    if (level < 0 || level > 3) return fail(HGI_EINVAL, "quantization level %d not in 0..3", level);
    const unsigned error = errs[level], scale = 2 * error + 1;
    for (unsigned i = 0; i < 256; ++i) lut[i] = (uint8_t)(((i + error) / scale) * scale);
    if (max_err) *max_err = (uint8_t)error;
    return HGI_OK;
}

// src/quantizator.rs:26-29
void hgi_noop_lut(uint8_t lut[256])
{
    if (lut)
        for (unsigned i = 0; i < 256; ++i) lut[i] = (uint8_t)i;
}

hgi_status hgi_encode_u8_dev(hgi_ctx *c, const void *d_img, uint32_t w, uint32_t h, uint32_t levels,
                             hgi_interp interp, const uint8_t lut[256], void *d_grid, size_t batch,
                             size_t frame_stride)
{
    HGI_TRY(check_common(c, d_img, d_grid, levels, interp, batch, frame_stride, w, h));
    if (!lut) return fail(HGI_EINVAL, "lut is NULL");
    if (w == 0 || h == 0 || batch == 0) return HGI_OK;
    if (batch == 1 && frame_stride < (size_t)w * h) frame_stride = (size_t)w * h;
    HIP_TRY(hipSetDevice(c->device));
    HGI_TRY(ws_ensure(c, ws_need(c, w, h, levels, batch, frame_stride)));
    c->ws_used = 0;
    return encode_impl(c, static_cast<const uint8_t *>(d_img), w, h, levels, interp, lut,
                       static_cast<uint8_t *>(d_grid), batch, frame_stride);
}

hgi_status hgi_decode_u8_dev(hgi_ctx *c, const void *d_grid, uint32_t w, uint32_t h, uint32_t levels,
                             hgi_interp interp, void *d_img, size_t batch, size_t frame_stride)
{
    HGI_TRY(check_common(c, d_grid, d_img, levels, interp, batch, frame_stride, w, h));
    if (w == 0 || h == 0 || batch == 0) return HGI_OK;
    if (batch == 1 && frame_stride < (size_t)w * h) frame_stride = (size_t)w * h;
    HIP_TRY(hipSetDevice(c->device));
    HGI_TRY(ws_ensure(c, ws_need(c, w, h, levels, batch, frame_stride)));
    c->ws_used = 0;
    return decode_impl(c, static_cast<const uint8_t *>(d_grid), w, h, levels, interp,
                       static_cast<uint8_t *>(d_img), batch, frame_stride);
}

static hgi_status host_banded(hgi_ctx *c, const uint8_t *in, uint8_t *out, uint32_t w, uint32_t h, uint32_t levels,
                              hgi_interp interp, const uint8_t *lut, bool encode);

// Host-pointer forms: stage through device scratch (PCIe-bound; never the number that is benchmarked).
static hgi_status host_roundtrip(hgi_ctx *c, const uint8_t *in, uint8_t *out, uint32_t w, uint32_t h,
                                 uint32_t levels, hgi_interp interp, const uint8_t *lut, bool encode)
{
    HGI_TRY(check_common(c, in, out, levels, interp, 1, (size_t)w * h, w, h));
    if (encode && !lut) return fail(HGI_EINVAL, "lut is NULL");
    if (w == 0 || h == 0) return HGI_OK;
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = (size_t)w * h;
    if (host_call_is_banded(c, w, h, levels)) return host_banded(c, in, out, w, h, levels, interp, lut, encode);
    HGI_TRY(ws_ensure(c, host_call_need(c, w, h, levels)));
    c->ws_used = 0;
    uint8_t *d_in = ws_take(c, n), *d_out = ws_take(c, n);
    if (!d_in || !d_out) return fail(HGI_ENOMEM, "scratch exhausted (host staging)");
    HIP_TRY(hipMemcpyAsync(d_in, in, n, hipMemcpyHostToDevice, c->stream));
    if (encode)
        HGI_TRY(encode_impl(c, d_in, w, h, levels, interp, lut, d_out, 1, n));
    else
        HGI_TRY(decode_impl(c, d_in, w, h, levels, interp, d_out, 1, n));
    HIP_TRY(hipMemcpyAsync(out, d_out, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return HGI_OK;
}

// One LARGE frame in host memory: the same overlap inside the frame.  A tile only depends on input pixels of its own
// rows and of the halo rows at offsets 0 .. 2^k <= 64 below them (65 rows), so a band of tile rows can be coded as soon as its rows and the next band
// are on the device, and downloaded while the bands further down still upload.  No kernel change: each band is a
// launch of the standard kernel on a row-shifted view of the frame (true remaining height, so the out-of-image rule is
// exact) limited to the band's tile rows.  Pyramids deeper than a tile need the stride-64 lattice of the WHOLE frame
// first: that is gathered on the host (65 536 bytes for 16384^2) and uploaded ahead of the bands.
static hgi_status host_banded(hgi_ctx *c, const uint8_t *in, uint8_t *out, uint32_t w, uint32_t h, uint32_t levels,
                              hgi_interp interp, const uint8_t *lut, bool encode)
{
    const size_t n = (size_t)w * h;
    HGI_TRY(pipe_ensure(c));
    // Pyramids deeper than a tile: the stride-2^k lattice (every 64th pixel of every 64th row) is gathered on the host --
    // it is tiny -- and goes up first; its seeds are ready long before the first band is.  The split is six levels + seed
    // planes here, not split_pyramid()'s: a tile that rebuilt the levels above it for itself (the cone) would read rows
    // far below its band, which have not been uploaded yet.  This path is bound by PCIe anyway (6 ms for 16384^2 against
    // 0.1 ms of kernels), and six keeps the host-side gather at w*h / 4096 bytes.
    const uint32_t k = levels < (uint32_t)kFusedMaxLevels ? levels : (uint32_t)kFusedMaxLevels;
    const bool deep = levels > k;
    const SubGeom g = deep ? sub_geom(w, h, k) : SubGeom{0, 0, 0};
    HGI_TRY(ws_ensure(c, host_call_need(c, w, h, levels)));
    c->ws_used = 0;
    uint8_t *d_in = ws_take(c, n), *d_out = ws_take(c, n);
    if (!d_in || !d_out) return fail(HGI_ENOMEM, "scratch exhausted (host staging)");
    HIP_TRY(hipStreamSynchronize(c->stream));
    uint8_t *sub_src = nullptr, *sub_grid = nullptr, *sub_rec = nullptr;
    std::vector<uint8_t> lattice;
    if (deep) {
        sub_src = ws_take(c, g.stride);
        sub_grid = encode ? ws_take(c, g.stride) : sub_src;     // decoding: the gathered plane IS the lattice's grid
        sub_rec = ws_take(c, g.stride);
        if (!sub_src || !sub_grid || !sub_rec) return fail(HGI_ENOMEM, "scratch exhausted (lattice planes)");
        lattice.resize((size_t)g.sw * g.sh);
        for (uint32_t sy = 0; sy < g.sh; ++sy) {
            const uint8_t *row = in + ((size_t)sy << k) * w;
            uint8_t *dstp = lattice.data() + (size_t)sy * g.sw;
            for (uint32_t sx = 0; sx < g.sw; ++sx) dstp[sx] = row[(size_t)sx << k];
        }
    }
    // bands of about 4 MiB (per-band fixed costs: two copies, an event, a launch), whole multiples of 64 rows, 2..16 bands
    uint32_t band = (uint32_t)(((4u << 20) / w + 63) / 64 * 64);
    if (band < 64) band = 64;
    if (band >= h) band = (h / 2 + 63) / 64 * 64;
    while ((h + band - 1) / band > 16) band += 64;
    const uint32_t nb = (h + band - 1) / band;
    const bool reg_in = hipHostRegister(const_cast<uint8_t *>(in), n, hipHostRegisterDefault) == hipSuccess;
    const bool reg_out = hipHostRegister(out, n, hipHostRegisterDefault) == hipSuccess;
    (void)hipGetLastError();
    hipStream_t up = c->pipe[0], down = c->pipe[1];
    const Lut256 l = encode ? pack_lut(lut) : Lut256{};
    const bool ident = encode && is_identity(lut);
    hipError_t e = hipSuccess;
    hgi_status st = HGI_OK;
    if (deep) {     // lattice -> seeds, ordered before the first band's kernel.  The upload goes on the upload stream:
                    // a stream that has issued a copy in one direction tends to keep its copies on that DMA engine, and
                    // downloads queued behind the sixteen band uploads would wait for all of them (measured: 10.4 ms
                    // instead of 6.4 ms for 16384^2)
        e = hipMemcpyAsync(sub_src, lattice.data(), lattice.size(), hipMemcpyHostToDevice, up);
        if (e == hipSuccess) e = hipEventRecord(c->ev_up[0], up);
        if (e == hipSuccess) e = hipStreamWaitEvent(down, c->ev_up[0], 0);
        if (e == hipSuccess) {
            hipStream_t saved = c->stream;
            c->stream = down;
            if (encode) st = encode_impl(c, sub_src, g.sw, g.sh, levels - k, interp, lut, sub_grid, 1, g.stride);
            if (st == HGI_OK) st = decode_impl(c, sub_grid, g.sw, g.sh, levels - k, interp, sub_rec, 1, g.stride);
            c->stream = saved;
        }
    }
    // Upload stream: band b goes up together with its halo rows -- the tiles of its last tile row read input rows down to
    // offset 2^k <= 64 below the band INCLUSIVE (halo row TH + 64 at k = 6), i.e. 65 rows of band b + 1 -- so that its
    // kernel waits for nothing else; band b + 1 then starts below them.
    // Knobs build only (HGI_TEST_BAND_HOLD; constant false, and the branches below gone, in the release library): d_in is
    // poisoned with 0xFF first and band b + 1's upload is held until band b's kernel has finished, so a kernel that read a
    // row its own upload did not cover would see poison, deterministically.
    const bool hold = HGI_SWITCH(HGI_TEST_BAND_HOLD);
    constexpr size_t kHaloRows = 65;
    auto upload = [&](uint32_t b) {
        const size_t y0 = b ? (size_t)b * band + kHaloRows : 0;
        size_t y1 = (size_t)(b + 1) * band + kHaloRows;
        if (y1 > h) y1 = h;
        hipError_t r = hipSuccess;
        if (y1 > y0) r = hipMemcpyAsync(d_in + y0 * w, in + y0 * w, (y1 - y0) * w, hipMemcpyHostToDevice, up);
        if (r == hipSuccess) r = hipEventRecord(c->ev_band[b], up);
        return r;
    };
    // compute + download stream
    auto code = [&](uint32_t b) {
        const size_t y0 = (size_t)b * band, rows = y0 + band <= h ? band : h - y0;
        hipError_t r = hipStreamWaitEvent(down, c->ev_band[b], 0);
        if (r != hipSuccess) return r;
        const Frames f = {w, (uint32_t)(h - y0), (uint64_t)((size_t)(h - y0) * w), 1};
        const uint32_t limit = b + 1 < nb ? band : 0;
        // the view starts y0 rows down (a multiple of 64 = 2^6 >= 2^k): its seeds start y0 >> k lattice rows down
        const size_t ly = y0 >> k;
        const Seeds sd = {deep ? sub_rec + ly * g.sw : nullptr, deep && encode ? sub_grid + ly * g.sw : nullptr, g.sw,
                          deep ? (uint32_t)(g.sh - ly) : 0u, g.stride, 0u};
        r = encode ? launch_encode_fused(d_in + y0 * w, d_out + y0 * w, f, k, interp, l, ident, deep ? &sd : nullptr, down, limit)
                   : launch_decode_fused(d_in + y0 * w, d_out + y0 * w, f, k, interp, deep ? &sd : nullptr, down, limit);
        if (r == hipSuccess && hold) r = hipEventRecord(c->ev_free[b % 3], down);
        if (r == hipSuccess) r = hipMemcpyAsync(out + y0 * w, d_out + y0 * w, rows * w, hipMemcpyDeviceToHost, down);
        return r;
    };
    if (hold) {
        if (e == hipSuccess) e = hipMemsetAsync(d_in, 0xFF, n, up);
        for (uint32_t b = 0; b < nb && e == hipSuccess && st == HGI_OK; ++b) {
            if (b) e = hipStreamWaitEvent(up, c->ev_free[(b - 1) % 3], 0);
            if (e == hipSuccess) e = upload(b);
            if (e == hipSuccess) e = code(b);
        }
    } else {
        for (uint32_t b = 0; b < nb && e == hipSuccess; ++b) e = upload(b);
        for (uint32_t b = 0; b < nb && e == hipSuccess && st == HGI_OK; ++b) e = code(b);
    }
    const hipError_t e0 = hipStreamSynchronize(up), e1 = hipStreamSynchronize(down);
    if (reg_in) (void)hipHostUnregister(const_cast<uint8_t *>(in));
    if (reg_out) (void)hipHostUnregister(out);
    c->ws_used = 0;
    if (st != HGI_OK) return st;
    if (e != hipSuccess || e0 != hipSuccess || e1 != hipSuccess)
        return fail(HGI_EDEVICE, "%s", hipGetErrorString(e != hipSuccess ? e : e0 != hipSuccess ? e0 : e1));
    return HGI_OK;
}

// Host-pointer BATCH forms.  The frames go through the device in chunks, as a directional pipeline: one stream only
// uploads, the other runs the kernels of a chunk and downloads it, three device slots decouple them -- so the upload
// of chunk j + 1 (and j + 2) runs while chunk j downloads.  PCIe is full duplex (tools/pcie.hip: 47 GB/s each way
// concurrently, 55 GB/s one way); the kernels disappear behind the transfers.
static hgi_status host_batch(hgi_ctx *c, const uint8_t *in, uint8_t *out, uint32_t w, uint32_t h, uint32_t levels,
                             hgi_interp interp, const uint8_t *lut, size_t batch, size_t frame_stride, bool encode)
{
    HGI_TRY(check_common(c, in, out, levels, interp, batch, frame_stride, w, h));
    if (encode && !lut) return fail(HGI_EINVAL, "lut is NULL");
    if (w == 0 || h == 0 || batch == 0) return HGI_OK;
    const size_t n = (size_t)w * h;
    if (batch == 1) frame_stride = n;
    HIP_TRY(hipSetDevice(c->device));
    constexpr int kSlots = 3;
    HGI_TRY(pipe_ensure(c));
    // chunks of about 8 MiB (at least one frame), and at least two chunks when there are two frames to overlap
    size_t fpc = (8u << 20) / n;
    if (fpc < 1) fpc = 1;
    if (fpc > (batch + 1) / 2) fpc = (batch + 1) / 2;
    const size_t slot = align_up(fpc * n, 256) + 256, planes = ws_need(c, w, h, levels, fpc, n), per = 2 * slot + planes + 512;
    HGI_TRY(ws_ensure(c, kSlots * per));
    HIP_TRY(hipStreamSynchronize(c->stream));          // ordered after whatever the caller queued on the ctx stream
    // hipMemcpyAsync on pageable memory blocks the host, which would serialise the pipeline: register the caller's
    // buffers for the duration of the call (about a microsecond here whatever the size; a buffer the caller registered
    // already, or a refusal, just leaves that side as it is)
    const size_t span = (batch - 1) * frame_stride + n;
    const bool reg_in = hipHostRegister(const_cast<uint8_t *>(in), span, hipHostRegisterDefault) == hipSuccess;
    const bool reg_out = hipHostRegister(out, span, hipHostRegisterDefault) == hipSuccess;
    (void)hipGetLastError();
    hipStream_t saved = c->stream, up = c->pipe[0], down = c->pipe[1];
    hgi_status st = HGI_OK;
    hipError_t e = hipSuccess;
    for (size_t j = 0, first = 0; first < batch && st == HGI_OK && e == hipSuccess; ++j, first += fpc) {
        const size_t frames = batch - first < fpc ? batch - first : fpc;
        const int sl = (int)(j % kSlots);
        uint8_t *base = c->ws + sl * per, *d_in = base, *d_out = base + slot;
        const uint8_t *src = in + first * frame_stride;
        uint8_t *dst = out + first * frame_stride;
        // upload stream: the slot's previous chunk must have been consumed by its kernels
        if (j >= (size_t)kSlots) e = hipStreamWaitEvent(up, c->ev_free[sl], 0);
        if (frame_stride == n) {
            if (e == hipSuccess) e = hipMemcpyAsync(d_in, src, frames * n, hipMemcpyHostToDevice, up);
        } else {
            for (size_t f = 0; f < frames && e == hipSuccess; ++f)
                e = hipMemcpyAsync(d_in + f * n, src + f * frame_stride, n, hipMemcpyHostToDevice, up);
        }
        if (e == hipSuccess) e = hipEventRecord(c->ev_up[sl], up);
        // compute + download stream (in-order: the slot's previous download precedes these kernels)
        if (e == hipSuccess) e = hipStreamWaitEvent(down, c->ev_up[sl], 0);
        if (e != hipSuccess) break;
        c->stream = down;                               // the launches below (and their scratch planes) belong to this slot
        c->ws_used = (size_t)(base + 2 * slot - c->ws);
        st = encode ? encode_impl(c, d_in, w, h, levels, interp, lut, d_out, frames, n)
                    : decode_impl(c, d_in, w, h, levels, interp, d_out, frames, n);
        c->stream = saved;
        if (st != HGI_OK) break;
        e = hipEventRecord(c->ev_free[sl], down);
        if (frame_stride == n) {
            if (e == hipSuccess) e = hipMemcpyAsync(dst, d_out, frames * n, hipMemcpyDeviceToHost, down);
        } else {
            for (size_t f = 0; f < frames && e == hipSuccess; ++f)
                e = hipMemcpyAsync(dst + f * frame_stride, d_out + f * n, n, hipMemcpyDeviceToHost, down);
        }
    }
    const hipError_t e0 = hipStreamSynchronize(up), e1 = hipStreamSynchronize(down);
    if (reg_in) (void)hipHostUnregister(const_cast<uint8_t *>(in));
    if (reg_out) (void)hipHostUnregister(out);
    c->ws_used = 0;
    if (st != HGI_OK) return st;
    if (e != hipSuccess || e0 != hipSuccess || e1 != hipSuccess)
        return fail(HGI_EDEVICE, "%s", hipGetErrorString(e != hipSuccess ? e : e0 != hipSuccess ? e0 : e1));
    return HGI_OK;
}

hgi_status hgi_encode_u8_batch(hgi_ctx *c, const uint8_t *imgs, uint32_t w, uint32_t h, uint32_t levels, hgi_interp interp,
                               const uint8_t lut[256], uint8_t *grids_out, size_t batch, size_t frame_stride)
{
    return host_batch(c, imgs, grids_out, w, h, levels, interp, lut, batch, frame_stride, true);
}

hgi_status hgi_decode_u8_batch(hgi_ctx *c, const uint8_t *grids, uint32_t w, uint32_t h, uint32_t levels, hgi_interp interp,
                               uint8_t *imgs_out, size_t batch, size_t frame_stride)
{
    return host_batch(c, grids, imgs_out, w, h, levels, interp, nullptr, batch, frame_stride, false);
}

hgi_status hgi_encode_u8(hgi_ctx *c, const uint8_t *img, uint32_t w, uint32_t h, uint32_t levels,
                         hgi_interp interp, const uint8_t lut[256], uint8_t *grid_out)
{
    return host_roundtrip(c, img, grid_out, w, h, levels, interp, lut, true);
}

hgi_status hgi_decode_u8(hgi_ctx *c, const uint8_t *grid, uint32_t w, uint32_t h, uint32_t levels,
                         hgi_interp interp, uint8_t *img_out)
{
    return host_roundtrip(c, grid, img_out, w, h, levels, interp, nullptr, false);
}

hgi_status hgi_synth_u8_dev(hgi_ctx *c, hgi_synth_kind kind, uint64_t seed, uint64_t first_frame, uint32_t w,
                            uint32_t h, void *d_out, size_t batch, size_t frame_stride)
{
    if (!c) return fail(HGI_EINVAL, "ctx is NULL");
    if ((int)kind < 0 || (int)kind > 2) return fail(HGI_EINVAL, "unknown synthetic kind %d", (int)kind);
    if (w == 0 || h == 0 || batch == 0) return HGI_OK;
    if (!d_out) return fail(HGI_EINVAL, "NULL buffer");
    if (batch > 1 && frame_stride < (size_t)w * h) return fail(HGI_EINVAL, "frame_stride < width*height");
    HIP_TRY(hipSetDevice(c->device));
    Frames f = {w, h, (uint64_t)frame_stride, (uint32_t)batch};
    HIP_TRY(launch_synth((int)kind, seed, first_frame, static_cast<uint8_t *>(d_out), f, c->stream));
    return HGI_OK;
}

hgi_status hgi_copy_u8_dev(hgi_ctx *c, const void *d_src, void *d_dst, size_t n)
{
    if (!c) return fail(HGI_EINVAL, "ctx is NULL");
    if (n == 0) return HGI_OK;
    if (!d_src || !d_dst) return fail(HGI_EINVAL, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_copy(static_cast<const uint8_t *>(d_src), static_cast<uint8_t *>(d_dst), n, c->stream));
    return HGI_OK;
}

hgi_status hgi_histogram_u8_dev(hgi_ctx *c, const void *d_grid, uint32_t w, uint32_t h, size_t batch, size_t frame_stride,
                                void *d_hist)
{
    if (!c) return fail(HGI_EINVAL, "ctx is NULL");
    if (batch == 0) return HGI_OK;
    if (!d_hist || ((w && h) && !d_grid)) return fail(HGI_EINVAL, "NULL buffer");
    if (batch > 1 && frame_stride < (size_t)w * h) return fail(HGI_EINVAL, "frame_stride < width*height");
    if (batch > 0xFFFFFFFFull) return fail(HGI_EINVAL, "batch too large");
    HIP_TRY(hipSetDevice(c->device));
    Frames f = {w, h, (uint64_t)frame_stride, (uint32_t)batch};
    HIP_TRY(launch_histogram(static_cast<const uint8_t *>(d_grid), f, static_cast<unsigned long long *>(d_hist), c->stream));
    return HGI_OK;
}

hgi_status hgi_diff_stats_dev(hgi_ctx *c, const void *d_before, const void *d_after, uint32_t w, uint32_t h,
                              size_t batch, size_t frame_stride, void *d_out)
{
    if (!c) return fail(HGI_EINVAL, "ctx is NULL");
    if (batch == 0) return HGI_OK;
    if (!d_out || ((w && h) && (!d_before || !d_after))) return fail(HGI_EINVAL, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    Frames f = {w, h, (uint64_t)frame_stride, (uint32_t)batch};
    HIP_TRY(launch_diff_stats(static_cast<const uint8_t *>(d_before), static_cast<const uint8_t *>(d_after), f,
                              static_cast<unsigned long long *>(d_out), c->stream));
    return HGI_OK;
}


hgi_status hgi_timer_start(hgi_ctx *c)
{
    if (!c) return fail(HGI_EINVAL, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    return HGI_OK;
}

hgi_status hgi_timer_stop(hgi_ctx *c, float *elapsed_ms)
{
    if (!c || !elapsed_ms) return fail(HGI_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(elapsed_ms, c->ev0, c->ev1));
    return HGI_OK;
}

}  // extern "C"

#ifdef HGI_TIMELINE
// Experiment builds only (make VARIANT=_tl EXTRA=-DHGI_TIMELINE; tools/timeline.py): a device buffer of four u64 per
// block into which the tile kernels log their interior blocks.  Not part of include/hgi.h, never in the shipped library.
namespace hgi {
uint64_t *g_timeline = nullptr;
}
extern "C" HGI_API void hgi_debug_timeline(void *device_buffer) { hgi::g_timeline = static_cast<uint64_t *>(device_buffer); }
#endif
