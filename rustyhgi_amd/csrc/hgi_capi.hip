// C ABI of libhgi_hip.so (include/hgi.h): argument checking, scratch management, level scheduling.
// No CPU implementation lives here: every encode/decode goes to the gfx950 kernels or fails.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/hgi.h"
#include "hgi_kernels.h"

using namespace hgi;

namespace {

thread_local char g_err[512] = "";

hgi_status fail(hgi_status st, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return st;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(e_ == hipErrorOutOfMemory ? HGI_ENOMEM : HGI_EDEVICE, "%s: %s", #expr, \
                        hipGetErrorString(e_));                                             \
    } while (0)

#define HGI_TRY(expr)                  \
    do {                               \
        hgi_status s_ = (expr);        \
        if (s_ != HGI_OK) return s_;   \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

#ifndef HGI_ENTROPY_GROUP_MIB_DEFAULT
#define HGI_ENTROPY_GROUP_MIB_DEFAULT 256
#endif
// The shallowest pyramid that runs as four fused levels + cone (split_pyramid()).  Six: a lone frame then keeps the small tiles and
// the short chain of a four-level launch (1920 x 1080 level 6: 10.6 / 6.5 -> 7.3 / 5.6 us), a batch is unchanged (64 x 4096^2:
// +0.4 / -0.8 %); at five levels nothing is gained in either (profiles/r03_cone_levels.txt).
#ifndef HGI_CONE_MIN_ENC_DEFAULT
#define HGI_CONE_MIN_ENC_DEFAULT 6
#endif
#ifndef HGI_CONE_MIN_DEC_DEFAULT
#define HGI_CONE_MIN_DEC_DEFAULT 6
#endif
#ifndef HGI_TILE16_MAX_DEFAULT
#define HGI_TILE16_MAX_DEFAULT 600    // an ENCODE of at most this many 32-row tiles runs on 16-row tiles instead (profiles/r03_sizes.txt: 1920 x 1080 is 510)
#endif

}  // namespace

struct hgi_ctx {
    int device;
    hipStream_t own_stream, stream;
    hgi_path path;
    uint8_t *ws;
    size_t ws_bytes, ws_used;
    hipEvent_t ev0, ev1;      // hgi_timer_start / hgi_timer_stop, nothing else
    hipEvent_t ev_hist[2];    // entropy stage: "histograms of group set k are down"
    hipEvent_t ev_probe[2];   // placement probe
    // host-pointer batch calls (created on first use): pipe[0] uploads, pipe[1] runs the kernels and downloads;
    // three device slots, per slot one event "uploaded" and one "kernels done, input slot free"
    hipStream_t pipe[2];
    hipEvent_t ev_up[3], ev_free[3];
    hipEvent_t ev_band[16];   // banded single-frame calls: "band uploaded"
    bool have_pipe;
    uint8_t *pin;             // pinned host memory (entropy stage: histograms and stream sizes come down without stalling the host)
    size_t pin_bytes;
};

namespace {

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
//   levels <= 5 (HGI_CONE_MIN_* - 1): the tile holds the pyramid: k = levels, nothing else.
//   6 ... 8: ONE launch at four fused levels that rebuilds the 2 ... 4 levels above a tile for itself (hgi_fused_impl.h,
//            cone_*): k = 4, up = levels - 4.  No lattice planes, no scratch, no launch in front.
//   deeper:  k = 4, up = 4, and the stride-256 lattice -- an HGI image with levels - 8 levels of its own (same OOB rule:
//            x < W <=> x >> 8 < ceil(W / 256)) -- is coded first; its planes are the cone's base.
//   HGI_CONE=0 in the environment (tests, experiments): no cone; the tile kernel takes six levels (or HGI_DEEP_K_ENC /
//            HGI_DEEP_K_DEC = 4 | 5) and the stride-2^k lattice is coded first.  Every split gives the same bytes.
struct Split {
    uint32_t k, up, shift;      // fused levels, cone levels, log2 of the lattice coded in front (0: none)
};

Split split_pyramid(uint32_t levels, bool encode)
{
    static const bool off = getenv("HGI_CONE") && atoi(getenv("HGI_CONE")) == 0;
    static const int lo[2] = {getenv("HGI_CONE_MIN_DEC") ? atoi(getenv("HGI_CONE_MIN_DEC")) : HGI_CONE_MIN_DEC_DEFAULT,
                              getenv("HGI_CONE_MIN_ENC") ? atoi(getenv("HGI_CONE_MIN_ENC")) : HGI_CONE_MIN_ENC_DEFAULT};
    static const int forced[2] = {getenv("HGI_DEEP_K_DEC") ? atoi(getenv("HGI_DEEP_K_DEC")) : 0,
                                  getenv("HGI_DEEP_K_ENC") ? atoi(getenv("HGI_DEEP_K_ENC")) : 0};
    const int m = lo[encode ? 1 : 0] < 5 ? 5 : lo[encode ? 1 : 0];
    if (!off && levels >= (uint32_t)m) {
        if (levels <= 8u) return {4u, levels - 4u, 0u};
        return {4u, 4u, 8u};
    }
    if (levels <= (uint32_t)kFusedMaxLevels) return {levels, 0u, 0u};
    const int f = forced[encode ? 1 : 0];
    const uint32_t k = (f >= kSeededMinLevels && f <= kFusedMaxLevels) ? (uint32_t)f : (uint32_t)kFusedMaxLevels;
    return {k, 0u, k};
}

// Scratch bytes one encode (or decode) of this shape takes, recursion included.  The bump allocator only resets
// between calls, so this mirrors encode_impl / decode_impl plane for plane: a decode with a lattice in front takes two
// planes per recursion level; an encode takes three -- and then both encodes AND decodes its lattice (the
// reconstruction is what the tile kernel starts from), each of which recurses on its own.
size_t plane_bytes(const SubGeom &g, size_t batch) { return align_up(batch * g.stride, 256) + 256; }

size_t ws_need_decode(uint32_t w, uint32_t h, uint32_t levels, size_t batch)
{
    const Split sp = split_pyramid(levels, false);
    if (!sp.shift) return 0;
    const SubGeom g = sub_geom(w, h, sp.shift);
    return 2 * plane_bytes(g, batch) + ws_need_decode(g.sw, g.sh, levels - sp.shift, batch);
}

size_t ws_need_encode(uint32_t w, uint32_t h, uint32_t levels, size_t batch)
{
    const Split sp = split_pyramid(levels, true);
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
// 128 x 16 tiles halve the finest level's share of it (profiles/r03_sizes.txt).  HGI_TILE_H=16|32|64 in the
// environment forces one where the pyramid fits (experiments, tests); HGI_TILE16_MAX moves the lower crossover.
uint32_t use_tile_rows(uint32_t w, uint32_t h, uint32_t k, size_t batch, bool encode)
{
    static const int forced = [] {
        const char *e = getenv("HGI_TILE_H");
        return e ? atoi(e) : 0;
    }();
    static const uint64_t tiny_max = [] {
        const char *e = getenv("HGI_TILE16_MAX");
        return e ? (uint64_t)atoll(e) : (uint64_t)HGI_TILE16_MAX_DEFAULT;
    }();
    if (k > (uint32_t)kFusedMaxLevelsSmall) return 64;
    const bool fits16 = k <= (uint32_t)kFusedMaxLevelsTiny;
    if (forced == 16 && fits16) return 16;
    if (forced == 32 || (forced == 16 && !fits16)) return 32;
    if (forced == 64) return 64;
    const uint64_t tx = (w + kTileW - 1) / kTileW;
    const uint64_t tiles64 = tx * ((h + 63) / 64) * batch, tiles32 = tx * ((h + 31) / 32) * batch;
    if (encode && fits16 && tiles32 <= tiny_max) return 16;      // (decode sits on the launch floor with 32-row tiles already)
    return tiles64 < 1536 ? 32 : 64;
}

// Deep pyramids: the one-workgroup-per-frame kernel for the levels above the fused depth, when the lattice plane is
// small (hgi_kernels.hip).  HGI_NO_LATTICE_KERNEL in the environment keeps the recursive path (tests).
bool use_lattice_kernel(const SubGeom &g, size_t batch)
{
    static const bool off = getenv("HGI_NO_LATTICE_KERNEL") != nullptr;
    return !off && lattice_pyramid_fits(g.sw, g.sh, batch);
}

hipError_t launch_encode_fused(const uint8_t *img, uint8_t *grid, const Frames &f, uint32_t k, int interp,
                               const Lut256 &lut, bool ident, const Seeds *seeds, hipStream_t s, uint32_t row_limit = 0)
{
    const uint32_t rows = row_limit && row_limit < f.height ? row_limit : f.height;   // what this launch really covers
    switch (use_tile_rows(f.width, rows, k, f.batch, true)) {
    case 16: return launch_encode_fused_16(img, grid, f, k, interp, lut, ident, seeds, s, row_limit);
    case 32: return launch_encode_fused_32(img, grid, f, k, interp, lut, ident, seeds, s, row_limit);
    default: return launch_encode_fused_64(img, grid, f, k, interp, lut, ident, seeds, s, row_limit);
    }
}

hipError_t launch_decode_fused(const uint8_t *grid, uint8_t *img, const Frames &f, uint32_t k, int interp,
                               const Seeds *seeds, hipStream_t s, uint32_t row_limit = 0)
{
    const uint32_t rows = row_limit && row_limit < f.height ? row_limit : f.height;
    switch (use_tile_rows(f.width, rows, k, f.batch, false)) {
    case 16: return launch_decode_fused_16(grid, img, f, k, interp, seeds, s, row_limit);
    case 32: return launch_decode_fused_32(grid, img, f, k, interp, seeds, s, row_limit);
    default: return launch_decode_fused_64(grid, img, f, k, interp, seeds, s, row_limit);
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

hgi_status decode_impl(hgi_ctx *c, const uint8_t *grid, uint32_t w, uint32_t h, uint32_t levels, int interp,
                       uint8_t *img, size_t batch, size_t stride);

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
    const Split sp = split_pyramid(levels, true);
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
    const Split sp = split_pyramid(levels, false);
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
        HIP_TRY(launch_decode_fused(grid, img, f, sp.k, interp, nullptr, c->stream));
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

}  // namespace

extern "C" {

const char *hgi_last_error(void) { return g_err; }
const char *hgi_version(void) { return "hgi-hip 0.1.0 (gfx950)"; }

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
    size_t stride = (size_t)w * h;
    size_t need = ws_need(c, w, h, levels, batch, stride);
    size_t host = 2 * (align_up(stride, 256) + 256);   // staging of the host-pointer entry points
    return ws_ensure(c, need + host);
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
    const size_t n = (size_t)w * h, slot = align_up(n, 256) + 256;
    // large frames with a pyramid one tile deep: band the frame so that its upload and download overlap
    if (n >= (4u << 20) && levels >= 1 && c->path != HGI_PATH_LEVELWISE && h >= 256 && !getenv("HGI_NO_BANDS"))
        return host_banded(c, in, out, w, h, levels, interp, lut, encode);
    HGI_TRY(ws_ensure(c, ws_need(c, w, h, levels, 1, n) + 2 * slot));
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

static hgi_status pipe_ensure(hgi_ctx *c)
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
    const size_t slot = align_up(n, 256) + 256;
    // Pyramids deeper than a tile: the stride-2^k lattice (every 64th pixel of every 64th row) is gathered on the host --
    // it is tiny -- and goes up first; its seeds are ready long before the first band is.  The split is six levels + seed
    // planes here, not split_pyramid()'s: a tile that rebuilt the levels above it for itself (the cone) would read rows
    // far below its band, which have not been uploaded yet.  This path is bound by PCIe anyway (6 ms for 16384^2 against
    // 0.1 ms of kernels), and six keeps the host-side gather at w*h / 4096 bytes.
    const uint32_t k = levels < (uint32_t)kFusedMaxLevels ? levels : (uint32_t)kFusedMaxLevels;
    const bool deep = levels > k;
    SubGeom g = {0, 0, 0};
    size_t lattice_need = 0;
    if (deep) {
        g = sub_geom(w, h, k);
        lattice_need = 3 * plane_bytes(g, 1) + ws_need_encode(g.sw, g.sh, levels - k, 1) + ws_need_decode(g.sw, g.sh, levels - k, 1);
    }
    HGI_TRY(ws_ensure(c, 2 * slot + lattice_need + 1024));
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
    // HGI_TEST_BAND_HOLD (tests): d_in is poisoned with 0xFF first and band b + 1's upload is held until band b's kernel
    // has finished, so a kernel that read a row its own upload did not cover would see poison, deterministically.
    static const bool hold = getenv("HGI_TEST_BAND_HOLD") != nullptr;
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

// ---- entropy stage --------------------------------------------------------------------------------------------
hgi_status hgi_huffman_plan(const uint64_t hist[286], uint8_t lens[286], uint16_t codes[286], uint8_t *header, size_t header_cap,
                            size_t *header_bits)
{
    if (!hist || !lens || !codes || !header || !header_bits) return fail(HGI_EINVAL, "NULL argument");
    uint64_t any = 0;
    for (int i = 0; i < kDeflateSymbols; ++i) any |= hist[i];
    if (!any) return fail(HGI_EINVAL, "empty histogram");
    *header_bits = huffman_plan(hist, lens, codes, header, header_cap);
    if (!*header_bits) return fail(HGI_EINVAL, "header buffer too small (%zu bytes)", header_cap);
    return HGI_OK;
}

// ---- the stage itself: `batch` grids, phase by phase, so that the host waits three times per GROUP of frames, not per frame
namespace {

constexpr size_t kHistBytes = (kMatchThresholds + 1) * kDeflateSymbols * 8;      // per frame, contiguous (one download)

struct DeflateGeom {
    uint64_t n;
    uint32_t nchunks;
    size_t dev_cap;       // bytes of stream buffer per frame on the device
    size_t group;         // frames whose stream buffers live in scratch at once
    size_t need;          // scratch bytes
};

DeflateGeom deflate_geom(uint64_t n, size_t batch)
{
    DeflateGeom g;
    g.n = n;
    g.nchunks = huffman_chunks(n);
    g.dev_cap = align_up((size_t)(n + n / 4) + 4096, 256);      // an optimal code averages < 9 bits per byte
    // a group's stream buffers: 256 MiB by default (two groups are in flight: one being packed, one being downloaded);
    // HGI_ENTROPY_GROUP_MIB in the environment sets another size (tools/entropy_packed_time.py: smaller groups shorten
    // the un-overlapped head and tail of the pipeline, more groups cost more synchronisations)
    static const size_t group_mib = [] {
        const char *e = getenv("HGI_ENTROPY_GROUP_MIB");
        const long v = e ? atol(e) : 0;
        return v > 0 ? (size_t)v : (size_t)HGI_ENTROPY_GROUP_MIB_DEFAULT;
    }();
    size_t group = (group_mib << 20) / g.dev_cap;
    if (group < 1) group = 1;
    if (group > batch) group = batch ? batch : 1;
    if (group > 256) group = 256;
    // equal groups: the last one is not a straggler
    const size_t ngroups = batch ? (batch + group - 1) / group : 1;
    if (batch) group = (batch + ngroups - 1) / ngroups;
    g.group = group;
    const size_t sets = ngroups > 1 ? 2 : 1;
    g.need = group * (sets * (kHistBytes + g.dev_cap) + kPlanBytes + (size_t)g.nchunks * 12 + 64) + (batch ? batch : 1) * 8 + 4096;
    return g;
}

using huff::FramePlan;

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

// The stage over `batch` grids, in groups of g.group frames, software-pipelined so that the device always has the next
// thing queued while the host builds codes or waits for a download:
//     device, c->stream :  hist(0) | hist(1) pack(0) | hist(2) pack(1) | ...
//     host              :          | plan(0)         | plan(1)         | ...      (several threads, one frame each)
//     device, pipe[1]   :                            | streams(0) down | streams(1) down ...
// hist = token histograms (one launch per group), plan = codes + headers, pack = count / scan / pack (three launches per
// group).  Stream sizes are known from the histograms, so the downloads are queued without waiting for the pack.
// offsets == nullptr: stream f goes to out + f * out_stride (cap = room per stream).  offsets != nullptr (packed): the
// streams of a group lie back to back on the device (64-byte aligned starts) and come down with ONE copy per group into
// out + offsets[f]; cap = room in `out` altogether.
hgi_status deflate_frames(hgi_ctx *c, const uint8_t *d_grids, uint32_t w, uint32_t h, size_t batch, size_t stride, uint8_t *out,
                          size_t out_stride, size_t cap, size_t *sizes, size_t *offsets = nullptr)
{
    const bool packed = offsets != nullptr;
    size_t packed_at = 0;                      // packed: where the next group starts in `out`
    const DeflateGeom g = deflate_geom((uint64_t)w * h, batch);
    const uint64_t n = g.n;
    // the bincode image of Grid { buffer: Vec<u8>, width: usize } (src/grid.rs:2-5): u64 length, the bytes, u64 width
    uint8_t prefix[8], suffix[8];
    for (int i = 0; i < 8; ++i) {
        prefix[i] = (uint8_t)(n >> (8 * i));
        suffix[i] = (uint8_t)((uint64_t)w >> (8 * i));
    }
    // Group boundaries: equal groups.  (Small first groups that double up to the full size -- to shorten the pipeline's
    // un-overlapped head, the first group's own histogram + plan + pack -- were tried: 9 groups instead of 6 for the C3
    // shard cost more in synchronisations than the head gave back: packed / strided 0.92 against 0.90, profiles/r03_entropy_groups.txt.)
    std::vector<size_t> starts;
    for (size_t at = 0; at < batch; at += g.group) starts.push_back(at);
    starts.push_back(batch);
    const size_t ngroups = starts.size() - 1;
    const bool piped = ngroups > 1;
    if (!n) {
        // nothing for the device to code: the front, then the tail, here
        std::vector<uint64_t> hist0((kMatchThresholds + 1) * kDeflateSymbols, 0);
        FramePlan p;
        if (!huff::plan_frame(reinterpret_cast<uint64_t (*)[kDeflateSymbols]>(hist0.data()), false, prefix, suffix, p))
            return fail(HGI_EDEVICE, "block header does not fit");
        const size_t total_bytes = (size_t)((p.exact_bits + 7) / 8), slot = align_up(total_bytes, 64);
        if (packed ? slot * batch > cap : total_bytes > cap)
            return fail(HGI_EINVAL, "output buffer too small: %zu bytes needed", packed ? slot * batch : total_bytes);
        for (size_t f = 0; f < batch; ++f) {
            uint8_t *dst = out + (packed ? f * slot : f * out_stride);
            if (packed) offsets[f] = f * slot;
            std::memset(dst, 0, total_bytes);
            std::memcpy(dst, p.block.front, p.block.front_bytes);
            uint64_t at = p.block.base_bits;
            const uint8_t *tail = reinterpret_cast<const uint8_t *>(p.block.tail);
            for (uint32_t i = 0; i < p.block.tail_bits; ++i, ++at) dst[at >> 3] |= (uint8_t)(((tail[i >> 3] >> (i & 7)) & 1u) << (at & 7));
            sizes[f] = total_bytes;
        }
        return HGI_OK;
    }
    HGI_TRY(ws_ensure(c, g.need));
    HGI_TRY(pin_ensure(c, 2 * g.group * kHistBytes + align_up(batch * 8, 256) + 2 * g.group * kPlanBytes));
    if (piped) HGI_TRY(pipe_ensure(c));
    c->ws_used = 0;
    uint8_t *d_hist[2], *d_outs[2];
    for (int k = 0; k < 2; ++k) d_hist[k] = (k == 0 || piped) ? ws_take(c, g.group * kHistBytes) : d_hist[0];
    uint8_t *d_plans = ws_take(c, g.group * kPlanBytes);
    uint64_t *d_totals = reinterpret_cast<uint64_t *>(ws_take(c, batch * 8));
    uint64_t *d_off = reinterpret_cast<uint64_t *>(ws_take(c, g.group * (size_t)g.nchunks * 8 + 8));
    uint32_t *d_cbits = reinterpret_cast<uint32_t *>(ws_take(c, g.group * (size_t)g.nchunks * 4 + 8));
    for (int k = 0; k < 2; ++k) d_outs[k] = (k == 0 || piped) ? ws_take(c, g.group * g.dev_cap) : d_outs[0];
    c->ws_used = 0;
    if (!d_hist[0] || !d_hist[1] || !d_plans || !d_totals || !d_off || !d_cbits || !d_outs[0] || !d_outs[1])
        return fail(HGI_ENOMEM, "scratch exhausted (entropy stage)");
    uint64_t *h_hist[2] = {reinterpret_cast<uint64_t *>(c->pin), reinterpret_cast<uint64_t *>(c->pin + g.group * kHistBytes)};
    uint64_t *h_totals = reinterpret_cast<uint64_t *>(c->pin + 2 * g.group * kHistBytes);
    const uint32_t dist_code = 0u | (1u << 24);      // distance symbol 0 (= distance 1): the one-bit code "0"
    std::vector<FramePlan> plans(g.group);
    // The plan blocks go up from pinned memory, two sets: the copy is then truly asynchronous (the host plans group gi + 1
    // while the device still packs group gi), and a set is rewritten only after ev_hist of two groups later -- which the
    // stream reaches behind this set's upload -- has been waited for.
    DeflatePlan *h_plans[2];
    h_plans[0] = reinterpret_cast<DeflatePlan *>(c->pin + 2 * g.group * kHistBytes + align_up(batch * 8, 256));
    h_plans[1] = h_plans[0] + g.group;
    std::vector<uint64_t> promised(batch), fixed_bits(batch);      // per frame: the stream's bits, and those that are not tokens
    std::vector<size_t> group_at(ngroups, 0), group_bytes(ngroups, 0);      // packed: a group's place in `out` and its length
    hipStream_t down = piped ? c->pipe[1] : c->stream;
    hipEvent_t *ev_hist = c->ev_hist;
    auto first_of = [&](size_t gi) { return starts[gi]; };
    auto count_of = [&](size_t gi) { return starts[gi + 1] - starts[gi]; };
    auto queue_hist = [&](size_t gi) -> hipError_t {
        const int set = (int)(gi & 1);
        hipError_t e = launch_token_histogram(d_grids + first_of(gi) * stride, n, stride, (uint32_t)count_of(gi),
                                              reinterpret_cast<unsigned long long *>(d_hist[set]), c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(h_hist[set], d_hist[set], count_of(gi) * kHistBytes, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipEventRecord(ev_hist[set], c->stream);
        return e;
    };
    auto queue_downloads = [&](size_t gi) -> hipError_t {
        const int set = (int)(gi & 1);
        hipError_t e = hipSuccess;
        if (piped) e = hipStreamWaitEvent(down, c->ev_free[set], 0);
        if (packed) {      // the group's streams are contiguous on the device: one copy
            if (e == hipSuccess && group_bytes[gi])
                e = hipMemcpyAsync(out + group_at[gi], d_outs[set], group_bytes[gi], hipMemcpyDeviceToHost, down);
        } else {
            for (size_t f = 0; f < count_of(gi) && e == hipSuccess; ++f) {
                const size_t frame = first_of(gi) + f;
                e = hipMemcpyAsync(out + frame * out_stride, d_outs[set] + f * g.dev_cap, (size_t)((promised[frame] + 7) / 8), hipMemcpyDeviceToHost, down);
            }
        }
        if (piped && e == hipSuccess) e = hipEventRecord(c->ev_up[set], down);
        return e;
    };
    // anything that fails after work was queued: drain before the host buffers the queue refers to go away
    auto bail = [&](hgi_status st) {
        (void)hipStreamSynchronize(c->stream);
        if (piped) (void)hipStreamSynchronize(down);
        return st;
    };
#define DF_TRY(expr)                                                                                                             \
    do {                                                                                                                         \
        hipError_t e_ = (expr);                                                                                                  \
        if (e_ != hipSuccess) return bail(fail(HGI_EDEVICE, "%s: %s", #expr, hipGetErrorString(e_)));                           \
    } while (0)
    DF_TRY(queue_hist(0));
    for (size_t gi = 0; gi < ngroups; ++gi) {
        const int set = (int)(gi & 1);
        const size_t g0 = first_of(gi), cnt = count_of(gi);
        DF_TRY(hipEventSynchronize(ev_hist[set]));
        if (gi + 1 < ngroups) DF_TRY(queue_hist(gi + 1));              // the device has this to do while the host plans
        // codes on the host, a frame per thread
        {
            std::vector<int> status(cnt, 0);
            auto work = [&](size_t t, size_t nt) {
                for (size_t f = t; f < cnt; f += nt)
                    status[f] = huff::plan_frame(reinterpret_cast<uint64_t (*)[kDeflateSymbols]>(h_hist[set] + f * (kHistBytes / 8)), true, prefix,
                                                 suffix, plans[f]) ? 0 : 1;
            };
            size_t nt = cnt / 2;
            if (nt > 8) nt = 8;
            if (nt <= 1) {
                work(0, 1);
            } else {
                std::vector<std::thread> pool;
                for (size_t t = 1; t < nt; ++t) pool.emplace_back(work, t, nt);
                work(0, nt);
                for (auto &th : pool) th.join();
            }
            for (size_t f = 0; f < cnt; ++f)
                if (status[f]) return bail(fail(HGI_EDEVICE, "block header does not fit"));
        }
        // the histograms say exactly how long each stream will be: never start packing into a buffer it would overrun
        size_t dev_at = 0;                     // packed: running offset inside the group's device buffer
        for (size_t f = 0; f < cnt; ++f) {
            const FramePlan &p = plans[f];
            const size_t bytes = (size_t)((p.exact_bits + 7) / 8);
            if (p.exact_bits / 8 + 64 > g.dev_cap)
                return bail(fail(HGI_EDEVICE, "entropy stage: stream of %llu bytes exceeds its scratch", (unsigned long long)(p.exact_bits / 8)));
            if (!packed && bytes > cap) return bail(fail(HGI_EINVAL, "output buffer too small: %zu bytes needed", bytes));
            h_plans[set][f] = p.block;
            const uint64_t off = packed ? dev_at : f * g.dev_cap;
            h_plans[set][f].out_off[0] = (uint32_t)off;
            h_plans[set][f].out_off[1] = (uint32_t)(off >> 32);
            if (packed) {
                offsets[g0 + f] = packed_at + dev_at;
                dev_at += align_up(bytes, 64);
            }
            promised[g0 + f] = p.exact_bits;
            fixed_bits[g0 + f] = p.block.base_bits + p.block.tail_bits;
        }
        if (packed) {
            if (packed_at + dev_at > cap)
                return bail(fail(HGI_EINVAL, "output buffer too small: %zu bytes needed for the first %zu frames", packed_at + dev_at, g0 + cnt));
            group_at[gi] = packed_at;
            group_bytes[gi] = dev_at;
            packed_at += dev_at;
        }
        // one upload of the plans, count / scan / pack over the whole group (its stream buffers are free once the group
        // two back has been downloaded)
        if (piped && gi >= 2) DF_TRY(hipStreamWaitEvent(c->stream, c->ev_up[set], 0));
        DF_TRY(hipMemcpyAsync(d_plans, h_plans[set], cnt * kPlanBytes, hipMemcpyHostToDevice, c->stream));
        DF_TRY(launch_huffman_pack(d_grids + g0 * stride, n, stride, (uint32_t)cnt, d_plans, dist_code, d_cbits, d_off, d_totals + g0, d_outs[set],
                                   c->stream));
        if (piped) DF_TRY(hipEventRecord(c->ev_free[set], c->stream));
        // downloads lag one group behind, so that the device has hist(gi + 1) and pack(gi) queued while they run
        if (piped) {
            if (gi >= 1) DF_TRY(queue_downloads(gi - 1));
        } else {
            DF_TRY(queue_downloads(gi));
        }
    }
    if (piped) DF_TRY(queue_downloads(ngroups - 1));
    DF_TRY(hipMemcpyAsync(h_totals, d_totals, batch * 8, hipMemcpyDeviceToHost, c->stream));
    DF_TRY(hipStreamSynchronize(c->stream));
    if (piped) DF_TRY(hipStreamSynchronize(down));
#undef DF_TRY
    for (size_t f = 0; f < batch; ++f) {
        const uint64_t got = fixed_bits[f] + h_totals[f];
        if (got != promised[f])
            return fail(HGI_EDEVICE, "entropy stage: packed %llu bits where the histograms promised %llu", (unsigned long long)got,
                        (unsigned long long)promised[f]);
        sizes[f] = (size_t)((promised[f] + 7) / 8);
    }
    return HGI_OK;
}

}  // namespace

// host-pointer form (what pairs with hgi_encode_u8): the grid goes up into scratch behind the stage's own buffers
hgi_status hgi_deflate_grid(hgi_ctx *c, const uint8_t *grid, uint32_t w, uint32_t h, uint8_t *out, size_t cap, size_t *bytes)
{
    if (!c || !out || !bytes) return fail(HGI_EINVAL, "NULL argument");
    const size_t n = (size_t)w * h;
    if (n && !grid) return fail(HGI_EINVAL, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    const size_t front = align_up(deflate_geom(n, 1).need, 256);
    HGI_TRY(ws_ensure(c, front + n + 256));
    uint8_t *staged = c->ws + front;
    if (n) HIP_TRY(hipMemcpyAsync(staged, grid, n, hipMemcpyHostToDevice, c->stream));
    return deflate_frames(c, staged, w, h, 1, n, out, cap, cap, bytes);
}

hgi_status hgi_deflate_grid_dev(hgi_ctx *c, const void *d_grid, uint32_t w, uint32_t h, uint8_t *out, size_t cap, size_t *bytes)
{
    if (!c || !out || !bytes) return fail(HGI_EINVAL, "NULL argument");
    if ((uint64_t)w * h && !d_grid) return fail(HGI_EINVAL, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    return deflate_frames(c, static_cast<const uint8_t *>(d_grid), w, h, 1, (size_t)w * h, out, cap, cap, bytes);
}

hgi_status hgi_deflate_grids_dev(hgi_ctx *c, const void *d_grids, uint32_t w, uint32_t h, size_t batch, size_t frame_stride, uint8_t *out,
                                 size_t out_stride, size_t *sizes)
{
    if (!c || (batch && (!out || !sizes))) return fail(HGI_EINVAL, "NULL argument");
    if (batch == 0) return HGI_OK;
    const size_t n = (size_t)w * h;
    if (n && !d_grids) return fail(HGI_EINVAL, "NULL buffer");
    if (batch > 1 && frame_stride < n) return fail(HGI_EINVAL, "frame_stride %zu < width*height", frame_stride);
    HIP_TRY(hipSetDevice(c->device));
    return deflate_frames(c, static_cast<const uint8_t *>(d_grids), w, h, batch, frame_stride, out, out_stride, out_stride, sizes);
}

hgi_status hgi_deflate_grids_packed_dev(hgi_ctx *c, const void *d_grids, uint32_t w, uint32_t h, size_t batch, size_t frame_stride,
                                        uint8_t *out, size_t cap, size_t *offsets, size_t *sizes)
{
    if (!c || (batch && (!out || !sizes || !offsets))) return fail(HGI_EINVAL, "NULL argument");
    if (batch == 0) return HGI_OK;
    const size_t n = (size_t)w * h;
    if (n && !d_grids) return fail(HGI_EINVAL, "NULL buffer");
    if (batch > 1 && frame_stride < n) return fail(HGI_EINVAL, "frame_stride %zu < width*height", frame_stride);
    HIP_TRY(hipSetDevice(c->device));
    return deflate_frames(c, static_cast<const uint8_t *>(d_grids), w, h, batch, frame_stride, out, 0, cap, sizes, offsets);
}

// ---- plane placement ------------------------------------------------------------------------------------------
// Measured on MI355X (profiles/r02_modes*.txt): the HBM behind one device is served in large physical
// regions (the driver's buddy blocks of up to 64 GiB never straddle one), and a kernel that streams one buffer in while
// streaming another out runs 4-5 % faster when the two lie in DIFFERENT regions than when they share one -- for the
// tile kernels 0.365 against 0.382 ms per GiB, for a linear copy 1.5 %.  Physical addresses are not visible from user
// space, so the only way to tell is to run the stream: the probe below times the decode kernel from one buffer into the
// other (its time does not depend on the bytes).  hgi_planes_alloc uses it to hand out planes whose neighbours in the
// array lie in different regions: what an encode -> decode chain (image -> grid -> image) wants.
namespace {

// mean time of decode launches prev -> cand over min(bytes, 2 GiB), as one frame 4096 wide
hgi_status probe_pair_ms(hgi_ctx *c, const uint8_t *prev, uint8_t *cand, size_t bytes, float *ms)
{
    const uint32_t w = 4096;
    size_t rows = bytes / w;
    if (rows > (2u << 20) / 4) rows = (2u << 20) / 4;          // 2 GiB: every byte offset stays below 2^32
    rows &= ~(size_t)63;
    const uint32_t h = (uint32_t)rows;
    constexpr int kWarm = 2, kTimed = 4;
    for (int i = 0; i < kWarm + kTimed; ++i) {
        if (i == kWarm) HIP_TRY(hipEventRecord(c->ev_probe[0], c->stream));
        c->ws_used = 0;
        // LeftTop: the same memory-access structure as Crossed, and a kernel name of its own in profiles
        // (k_dec_tiles<0, ...>), so that probe launches are never counted among the workload's k_dec_tiles<1, ...>
        HGI_TRY(decode_impl(c, prev, w, h, 4, HGI_INTERP_LEFTTOP, cand, 1, (size_t)w * h));
    }
    HIP_TRY(hipEventRecord(c->ev_probe[1], c->stream));
    HIP_TRY(hipEventSynchronize(c->ev_probe[1]));
    HIP_TRY(hipEventElapsedTime(ms, c->ev_probe[0], c->ev_probe[1]));
    *ms /= kTimed;
    return HGI_OK;
}

constexpr size_t kProbeMinBytes = (size_t)512 << 20;   // below this the stream lives in the 256 MiB Infinity Cache: no signal
constexpr float kFastRatio = 0.96f;   // a pair counts as "different regions" when it streams in < 0.96 of the same-region
                                      // yardstick's time (measured: 0.93-0.94 across regions, 0.97-1.0 within or between
                                      // some pairs of blocks)
constexpr int kMaxCandidates = 10;

}  // namespace

hgi_status hgi_probe_pair_u8_dev(hgi_ctx *c, const void *d_src, void *d_dst, size_t bytes, float *ms)
{
    if (!c || !d_src || !d_dst || !ms) return fail(HGI_EINVAL, "NULL argument");
    if (bytes < 4096 * 64) return fail(HGI_EINVAL, "probe needs at least 256 KiB");
    const uintptr_t a = reinterpret_cast<uintptr_t>(d_src), b = reinterpret_cast<uintptr_t>(d_dst);
    if (a < b + bytes && b < a + bytes) return fail(HGI_EINVAL, "probe buffers overlap");
    HIP_TRY(hipSetDevice(c->device));
    HGI_TRY(ws_ensure(c, ws_need(c, 4096, 4096, 4, 1, (size_t)4096 * 4096)));
    return probe_pair_ms(c, static_cast<const uint8_t *>(d_src), static_cast<uint8_t *>(d_dst), bytes, ms);
}

hgi_status hgi_planes_alloc(hgi_ctx *c, size_t bytes, uint32_t count, void **planes, int *separated)
{
    if (!c || !planes) return fail(HGI_EINVAL, "NULL argument");
    if (separated) *separated = 0;
    for (uint32_t i = 0; i < count; ++i) planes[i] = nullptr;
    if (count == 0 || bytes == 0) return HGI_OK;
    // Planes of 128 ... 512 MiB: a launch that reads one and writes the next (2 x 256 MiB for a lone 16384^2 frame) no longer
    // fits the 256 MiB Infinity Cache, so placement matters to it (16384^2 level 8: encode 101 -> 98.3 us, decode 99.8 -> 97.5,
    // profiles/r03_c4_placement.txt) -- but a probe over less than 512 MiB would measure that cache, not the regions, and at
    // exactly 512 MiB the signal is too weak to call (separated = 0 in every run).  Such planes are allocated at 1 GiB, the
    // size the probe was calibrated on; the caller uses their first `bytes`.
    if (bytes >= ((size_t)128 << 20) && bytes < ((size_t)1 << 30) && count > 1 && !getenv("HGI_NO_PLACEMENT")) bytes = (size_t)1 << 30;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<void *> bufs, spacers;     // candidate planes; allocations that only push the driver onwards
    void *ref = nullptr;                   // one allocation whose two halves are the same-region yardstick
    auto release = [&](std::vector<void *> &v) {
        for (void *p : v)
            if (p) (void)hipFree(p);
        v.clear();
        (void)hipGetLastError();
    };
    auto bail = [&](hgi_status st) {
        release(bufs);
        release(spacers);
        if (ref) (void)hipFree(ref);
        for (uint32_t i = 0; i < count; ++i) planes[i] = nullptr;
        return st;
    };
    for (uint32_t i = 0; i < count; ++i) {
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) return bail(fail(HGI_ENOMEM, "hipMalloc of %zu bytes failed", bytes));
        bufs.push_back(p);
    }
    auto hand_out = [&](const std::vector<int> &order) {
        std::vector<char> used(bufs.size(), 0);
        for (uint32_t i = 0; i < count; ++i) {
            planes[i] = bufs[(size_t)order[i]];
            used[(size_t)order[i]] = 1;
        }
        for (size_t j = 0; j < bufs.size(); ++j)
            if (!used[j]) (void)hipFree(bufs[j]);
        bufs.clear();
        release(spacers);
        if (ref) (void)hipFree(ref);
        ref = nullptr;
        (void)hipGetLastError();
    };
    std::vector<int> plain(count);
    for (uint32_t i = 0; i < count; ++i) plain[i] = (int)i;
    const bool probing = bytes >= kProbeMinBytes && count > 1 && !getenv("HGI_NO_PLACEMENT");
    if (!probing) {
        hand_out(plain);
        return HGI_OK;
    }
    {
        const hgi_status st = ws_ensure(c, ws_need(c, 4096, 4096, 4, 1, (size_t)4096 * 4096));
        if (st != HGI_OK) return bail(st);
    }
    // The yardstick: what a stream costs when source and destination share a region.  A power-of-two request is served
    // as ONE block, and a block never straddles a region, so the two halves of such an allocation are a same-region pair
    // by construction.  (Comparisons among candidates alone cannot tell "all fast" from "all slow".)
    const size_t span = bytes < ((size_t)2 << 30) ? bytes : ((size_t)2 << 30);     // what a probe streams
    size_t ref_bytes = 1;
    while (ref_bytes < 2 * span) ref_bytes <<= 1;
    if (hipMalloc(&ref, ref_bytes) != hipSuccess) {
        (void)hipGetLastError();
        ref = nullptr;
        hand_out(plain);
        return HGI_OK;
    }
    uint8_t *ref_lo = static_cast<uint8_t *>(ref), *ref_hi = ref_lo + ref_bytes / 2;
    // Every pair is timed AGAINST the yardstick, interleaved with it, after the yardstick has stopped drifting: the
    // device's clocks fall back within milliseconds of idling (an allocation in between is enough) and ramp for ~25 ms
    // once work resumes (profiles/r02_ramp.txt), so absolute times taken at different moments do not compare.
    auto other_region = [&](int a, int b, bool *yes) -> hgi_status {
        float last = 0, ms = 0;
        HGI_TRY(probe_pair_ms(c, ref_lo, ref_hi, span, &last));
        for (int it = 0; it < 12; ++it) {
            HGI_TRY(probe_pair_ms(c, ref_lo, ref_hi, span, &ms));
            const bool steady = ms <= last * 1.007f && last <= ms * 1.007f;
            last = ms;
            if (steady) break;
        }
        float same = 0, pair = 0;
        for (int rep = 0; rep < 2; ++rep) {
            HGI_TRY(probe_pair_ms(c, ref_lo, ref_hi, span, &ms));
            same += ms;
            HGI_TRY(probe_pair_ms(c, static_cast<const uint8_t *>(bufs[(size_t)a]), static_cast<uint8_t *>(bufs[(size_t)b]), span, &ms));
            pair += ms;
        }
        *yes = pair < same * kFastRatio;
        return HGI_OK;
    };
    // Candidates are sorted into groups that share a region (a candidate joins the first group whose representative it
    // does NOT stream fast against).  `count` planes whose neighbours differ exist as soon as no group has to supply
    // more than every other plane.  Until then: one more candidate, behind a spacer.  The driver serves requests
    // buddy-style from blocks of up to 64 GiB, the smallest free piece that fits first, so candidates of one size tend to
    // come from one block until it is used up (profiles/r02_modes4.txt: runs of 16); spacers of `bytes`, 2 x, 4 x ...
    // take that block's free buddies.  Large allocations take the driver seconds (it clears them), hence the caps.
    std::vector<std::vector<int>> groups;
    size_t classified = 0;
    int spacer_shift = 0;
    std::vector<int> order;
    for (;;) {
        for (; classified < bufs.size(); ++classified) {
            bool placed = false;
            for (auto &g : groups) {
                bool other = false;
                const hgi_status st = other_region(g[0], (int)classified, &other);
                if (st != HGI_OK) return bail(st);
                if (!other) {
                    g.push_back((int)classified);
                    placed = true;
                    break;
                }
            }
            if (!placed) groups.push_back(std::vector<int>{(int)classified});
        }
        // greedy arrangement: always take from the largest remaining group that is not the one just used
        std::vector<size_t> left(groups.size());
        for (size_t g = 0; g < groups.size(); ++g) left[g] = groups[g].size();
        order.clear();
        int prev = -1;
        while (order.size() < count) {
            int pick = -1;
            for (size_t g = 0; g < groups.size(); ++g)
                if ((int)g != prev && left[g] > 0 && (pick < 0 || left[g] > left[(size_t)pick])) pick = (int)g;
            if (pick < 0) break;
            order.push_back(groups[(size_t)pick][groups[(size_t)pick].size() - left[(size_t)pick]]);
            --left[(size_t)pick];
            prev = pick;
        }
        if (order.size() == count) break;                                   // neighbours all in different regions
        if (bufs.size() >= (size_t)count + kMaxCandidates) break;           // give up
        size_t free_b = 0, total_b = 0;
        const size_t want = bytes << (spacer_shift < 5 ? spacer_shift : 5);      // 1, 2, 4, 8, 16, then 32 x bytes each time
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && want <= free_b / 3) {
            void *fill = nullptr;
            if (hipMalloc(&fill, want) == hipSuccess) spacers.push_back(fill);
            (void)hipGetLastError();
            ++spacer_shift;
        }
        void *cand = nullptr;
        if (hipMalloc(&cand, bytes) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        bufs.push_back(cand);
    }
    const bool ok = order.size() == count;
    if (!ok) {      // could not be established: still alternate between the two largest groups as far as they go
        std::vector<char> used(bufs.size(), 0);
        for (int j : order) used[(size_t)j] = 1;
        for (size_t j = 0; j < bufs.size() && order.size() < count; ++j)
            if (!used[j]) order.push_back((int)j);
    }
    hand_out(order);
    if (separated) *separated = ok ? 1 : 0;
    return HGI_OK;
}

hgi_status hgi_planes_free(hgi_ctx *c, uint32_t count, void **planes)
{
    if (!c || (!planes && count)) return fail(HGI_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    for (uint32_t i = 0; i < count; ++i) {
        if (planes[i]) HIP_TRY(hipFree(planes[i]));
        planes[i] = nullptr;
    }
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
