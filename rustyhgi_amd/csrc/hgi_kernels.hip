// gfx950 (MI355X / CDNA4) kernels of the HGI encode/decode core: level-wise path + harness kernels.
// (The fused LDS-tile path lives in hgi_fused_impl.h / hgi_fused_{dec,enc}.hip; shared device helpers in hgi_dev.h.)
//
// Algorithm (reference, paths relative to pl0q1n/RustyHGI):
//   src/encoder.rs:39-71   closed-loop encode: predict -> residual -> quantize -> overflow fallback
//   src/decoder.rs:18-46   decode: predict + residual (wrapping)
//   src/interpolator.rs    LeftTop (:15-28), Crossed (:57-90, prediction :41-55, OOB corner -> 0 :75-82)
//   src/utils.rs:12-41     per-level pixel set: coordinates = 0 (mod sub), not both = 0 (mod 2*sub)
//
// Level-wise: one launch per level, one thread per step-cell, global-memory stencil.  Simple and
// strided; kept as the on-device cross-check and for measuring single passes.
// All arithmetic is u8/integer; there is no MFMA-shaped work on this path.
#include <stdlib.h>
#include <atomic>

#include "hgi_dev.h"
#include "hgi_knobs.h"

namespace hgi {
namespace {

using namespace dev;

// ---------------------------------------------------------------------------------------------
// level-wise path
// ---------------------------------------------------------------------------------------------
// src/encoder.rs:26-37 / src/decoder.rs:22-28: copy the stride-2^levels lattice.
__global__ void k_seed(const u8 *__restrict__ src, u8 *__restrict__ dst, Frames f, u32 levels)
{
    u64 nbx = (((u64)f.width - 1) >> levels) + 1, nby = (((u64)f.height - 1) >> levels) + 1;
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbx * nby) return;
    u64 by = i / nbx, bx = i - by * nbx;
    u64 off = (by << levels) * f.width + (bx << levels);
    for (u32 fr = blockIdx.y; fr < f.batch; fr += gridDim.y)
        dst[fr * f.frame_stride + off] = src[fr * f.frame_stride + off];
}

// One thread per step-cell: the cell's three new pixels share one prediction.
template <int INTERP, bool ENCODE, bool IDENT>
__global__ void k_level(u8 *__restrict__ plane,   // decode: image being built; encode: reconstruction
                        u8 *__restrict__ grid,    // decode: residuals (read); encode: residuals (written)
                        Frames f, u32 log2sub, Lut256 lut)
{
    __shared__ u8 slut[256];
    if (ENCODE && !IDENT) {
        if (threadIdx.x < 64) reinterpret_cast<u32 *>(slut)[threadIdx.x] = lut.w[threadIdx.x];
        __syncthreads();
    }
    const u64 W = f.width, H = f.height;
    const u64 sub = 1ull << log2sub, step = sub << 1;
    const u64 ncx = (W + step - 1) / step, ncy = (H + step - 1) / step;
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncx * ncy) return;
    u64 cy = i / ncx, cx = i - cy * ncx;
    u64 x0 = cx * step, y0 = cy * step, x1 = x0 + step, y1 = y0 + step, xs = x0 + sub, ys = y0 + sub;
    for (u32 fr = blockIdx.y; fr < f.batch; fr += gridDim.y) {
        u8 *im = plane + fr * f.frame_stride;
        u8 *g = grid + fr * f.frame_stride;
        u32 lt = im[y0 * W + x0];
        u32 rt = y1 < H ? im[y1 * W + x0] : 0u;                 // src/interpolator.rs:75-82
        u32 lb = x1 < W ? im[y0 * W + x1] : 0u;
        u32 rb = (x1 < W && y1 < H) ? im[y1 * W + x1] : 0u;
        u32 p = pred1<INTERP>(lt, rt, lb, rb);
        u64 pos[3] = {y0 * W + xs, ys * W + x0, ys * W + xs};
        bool ok[3] = {xs < W, ys < H, xs < W && ys < H};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (!ok[j]) continue;
            if (ENCODE) {
                u32 q = quant1<IDENT>(im[pos[j]], p, slut);
                g[pos[j]] = (u8)q;                              // src/encoder.rs:62
                im[pos[j]] = (u8)(p + q);                       // :63-64
            } else {
                im[pos[j]] = (u8)(p + g[pos[j]]);               // src/decoder.rs:39-40
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// lattice gather, harness kernels
// ---------------------------------------------------------------------------------------------
__global__ void k_gather_lattice(const u8 *__restrict__ src, Frames f, u32 k, u8 *__restrict__ dst,
                                 u32 sw, u32 sh, u64 dst_stride)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (u64)sw * sh) return;
    u64 sy = i / sw, sx = i - sy * sw;
    for (u32 fr = blockIdx.y; fr < f.batch; fr += gridDim.y)
        dst[fr * dst_stride + i] = src[fr * f.frame_stride + (sy << k) * f.width + (sx << k)];
}

// ---------------------------------------------------------------------------------------------
// upper part of a deep pyramid, one workgroup per frame
// ---------------------------------------------------------------------------------------------
// The lattice = 0 (mod 2^k) of a frame is itself an HGI image with `up` = levels - k levels (same out-of-image rule).
// When it is small it is not worth three launches (gather, encode, decode -- the encoder needs the lattice's
// reconstruction as seeds): one workgroup gathers it straight from the frame, runs the `up` levels closed-loop in
// LDS (src/encoder.rs:45-68 / src/decoder.rs:30-43, cells of a level in parallel) and writes the residual and the
// reconstruction planes the tile kernels take as seeds.  LDS: [originals -> residuals][reconstruction][table].
template <int INTERP, bool ENCODE>
__global__ __launch_bounds__(1024) void k_lattice_pyramid(const u8 *__restrict__ src, Frames f, u32 k, u32 up, Lut256 lut,
                                                          u32 ident, u8 *__restrict__ out_q, u8 *__restrict__ out_rec,
                                                          u32 sw, u32 sh, u64 dst_stride)
{
    extern __shared__ __attribute__((aligned(16))) u8 lds[];
    const u32 n = sw * sh, n_al = (n + 15u) & ~15u;
    u8 *qa = lds, *rec = lds + n_al, *slut = lds + 2 * n_al;
    const u32 tid = threadIdx.x, nt = blockDim.x;
    const u8 *fr = src + (size_t)blockIdx.x * f.frame_stride;
    const u32 bmask = up >= 32 ? ~0u : (1u << up) - 1u;          // base lattice of the sub-image: x, y = 0 (mod 2^up)
    if (ENCODE && tid < 64) reinterpret_cast<u32 *>(slut)[tid] = lut.w[tid];
    for (u32 i = tid; i < n; i += nt) {
        const u32 y = i / sw, x = i - y * sw;
        const u8 v = fr[((size_t)y << k) * f.width + ((size_t)x << k)];
        qa[i] = v;
        rec[i] = (ENCODE || !((x | y) & bmask)) ? v : (u8)0;      // encode: originals; decode: base samples, rest built below
    }
    __syncthreads();
    for (u32 l = 0; l < up; ++l) {
        const u32 e = up - l, sub = 1u << (e - 1);
        if (sub >= sw && sub >= sh) continue;                      // no pixel of this level lies inside the plane (uniform)
        const u32 ncx = ((sw - 1) >> e) + 1, ncy = ((sh - 1) >> e) + 1;
        for (u32 c = tid; c < ncx * ncy; c += nt) {
            const u32 cy = c / ncx, cx = c - cy * ncx;
            const u32 x0 = cx << e, y0 = cy << e, x1 = x0 + (sub << 1), y1 = y0 + (sub << 1);
            const bool xi = x1 < sw, yi = y1 < sh;                 // src/interpolator.rs:75-82: corners outside read 0
            const u32 lt = rec[y0 * sw + x0], rt = yi ? rec[y1 * sw + x0] : 0u;
            const u32 lb = xi ? rec[y0 * sw + x1] : 0u, rb = (xi && yi) ? rec[y1 * sw + x1] : 0u;
            const u32 p = pred1<INTERP>(lt, rt, lb, rb);
            const bool xn = x0 + sub < sw, yn = y0 + sub < sh;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const bool on = j == 0 ? xn : j == 1 ? yn : (xn && yn);
                if (!on) continue;
                const u32 idx = (y0 + (j >= 1 ? sub : 0u)) * sw + x0 + (j != 1 ? sub : 0u);
                if (ENCODE) {
                    const u32 q = ident ? ((qa[idx] - p) & 255u) : quant1<false>(qa[idx], p, slut);
                    qa[idx] = (u8)q;
                    rec[idx] = (u8)(p + q);
                } else {
                    rec[idx] = (u8)(p + qa[idx]);
                }
            }
        }
        __syncthreads();
    }
    for (u32 i = tid; i < n; i += nt) {
        if (ENCODE) out_q[(size_t)blockIdx.x * dst_stride + i] = qa[i];
        out_rec[(size_t)blockIdx.x * dst_stride + i] = rec[i];
    }
}

__device__ __forceinline__ u64 mix64(u64 z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// SURVEY.md 8(d) formulas; XY is the reference's own generator (benches/bench.rs:26-28).
__global__ void k_synth(int kind, u64 seed, u64 first_frame, u8 *__restrict__ out, Frames f)
{
    u64 n = (u64)f.width * f.height;
    for (u32 fr = blockIdx.y; fr < f.batch; fr += gridDim.y) {
        u8 *o = out + fr * f.frame_stride;
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
            u32 y = (u32)(i / f.width), x = (u32)(i - (u64)y * f.width);
            u8 v;
            if (kind == 0) {
                v = (u8)((x * y) & 0xFFu);
            } else {
                u8 nz = (u8)(mix64(seed ^ ((first_frame + fr) << 40) ^ ((u64)y << 20) ^ (u64)x) >> 56);
                v = kind == 1 ? nz : (u8)((((3u * x + 5u * y) >> 4) + (nz & 0x0Fu)) & 0xFFu);
            }
            o[i] = v;
        }
    }
}

// Streaming copy in the shape that measured fastest on MI355X (tools/membw.hip: 6.2 TB/s, against 4.7 TB/s
// for hipMemcpyAsync D2D and for a grid-stride loop): 16 B per lane, one access per lane and block,
// non-temporal on both sides.  It is the bench's "same bytes, no arithmetic" yardstick.
typedef u32 copy_v4u __attribute__((ext_vector_type(4)));
__global__ void k_copy16(const copy_v4u *__restrict__ src, copy_v4u *__restrict__ dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
__global__ void k_copy1(const u8 *__restrict__ src, u8 *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

// per-frame byte histogram of a grid batch -- the front end of an entropy coder (or of a rate
// estimate) on the device.  Residual grids are dominated by a handful of values, so a plain LDS histogram would
// serialise on them: each block keeps kHistCopies copies, value-major (h[v][copy]: the copies of one value sit in
// different banks), a lane adds into copy (lane & 15), and the copies are folded when the block is done.
constexpr int kHistCopies = 16;
__global__ __launch_bounds__(256) void k_histogram(const u8 *__restrict__ src, Frames f, unsigned long long *__restrict__ hist)
{
    __shared__ u32 h[256 * kHistCopies];
    const u64 n = (u64)f.width * f.height;
    const u32 copy = threadIdx.x & (kHistCopies - 1);
    for (u32 fr = blockIdx.y; fr < f.batch; fr += gridDim.y) {
        for (int i = threadIdx.x; i < 256 * kHistCopies; i += blockDim.x) h[i] = 0;
        __syncthreads();
        const u8 *p = src + fr * f.frame_stride;
        const u64 chunks = n >> 4;
        for (u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x; c < chunks; c += (u64)gridDim.x * blockDim.x) {
            uint4 v;
            __builtin_memcpy(&v, p + (c << 4), 16);     // any alignment: one 16-B load
            const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
#pragma unroll
                for (int b = 0; b < 4; ++b) atomicAdd(&h[((w[d] >> (8 * b)) & 255u) * kHistCopies + copy], 1u);
            }
        }
        if (blockIdx.x == 0)                            // the last n % 16 bytes of the frame
            for (u64 i = (chunks << 4) + threadIdx.x; i < n; i += blockDim.x) atomicAdd(&h[(u32)p[i] * kHistCopies + copy], 1u);
        __syncthreads();
        if (threadIdx.x < 256) {
            u32 sum = 0;
#pragma unroll
            for (int k = 0; k < kHistCopies; ++k) sum += h[threadIdx.x * kHistCopies + ((k + threadIdx.x) & (kHistCopies - 1))];
            if (sum) atomicAdd(&hist[(size_t)fr * 256 + threadIdx.x], (unsigned long long)sum);
        }
        __syncthreads();
    }
}

// src/main.rs:84-92 per frame: sum of squared differences, max |diff|, count of differing pixels
__global__ void k_diff_stats(const u8 *__restrict__ a, const u8 *__restrict__ b, Frames f,
                             unsigned long long *__restrict__ out)
{
    u64 n = (u64)f.width * f.height;
    for (u32 fr = blockIdx.y; fr < f.batch; fr += gridDim.y) {
        const u8 *pa = a + fr * f.frame_stride, *pb = b + fr * f.frame_stride;
        unsigned long long sq = 0, cnt = 0, mx = 0;
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
            int d = (int)pa[i] - (int)pb[i];
            d = d < 0 ? -d : d;
            sq += (unsigned)(d * d);
            cnt += d != 0;
            mx = (unsigned)d > mx ? (unsigned)d : mx;
        }
        for (int o = 32; o > 0; o >>= 1) {
            sq += __shfl_down(sq, o, 64);
            cnt += __shfl_down(cnt, o, 64);
            unsigned long long m2 = __shfl_down(mx, o, 64);
            mx = m2 > mx ? m2 : mx;
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&out[3 * fr + 0], sq);
            atomicMax(&out[3 * fr + 1], mx);
            atomicAdd(&out[3 * fr + 2], cnt);
        }
    }
}

inline bool ptr16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline u32 batch_grid_y(const Frames &f) { return f.batch < 65535u ? f.batch : 65535u; }

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
hipError_t launch_seed(const uint8_t *src, uint8_t *dst, const Frames &f, uint32_t levels, hipStream_t s)
{
    u64 nbx = (((u64)f.width - 1) >> levels) + 1, nby = (((u64)f.height - 1) >> levels) + 1;
    u64 blocks = (nbx * nby + 255) / 256;
    hipLaunchKernelGGL(k_seed, dim3((u32)blocks, batch_grid_y(f)), dim3(256), 0, s, src, dst, f, levels);
    return hipGetLastError();
}

template <int INTERP>
static hipError_t launch_level_t(uint8_t *plane, uint8_t *grid, const Frames &f, uint32_t log2sub,
                                 bool encode, bool ident, const Lut256 &lut, hipStream_t s)
{
    const u64 step = 2ull << log2sub;
    u64 cells = ((f.width + step - 1) / step) * ((f.height + step - 1) / step);
    dim3 g((u32)((cells + 255) / 256), batch_grid_y(f)), b(256);
    if (!encode)
        hipLaunchKernelGGL((k_level<INTERP, false, true>), g, b, 0, s, plane, grid, f, log2sub, lut);
    else if (ident)
        hipLaunchKernelGGL((k_level<INTERP, true, true>), g, b, 0, s, plane, grid, f, log2sub, lut);
    else
        hipLaunchKernelGGL((k_level<INTERP, true, false>), g, b, 0, s, plane, grid, f, log2sub, lut);
    return hipGetLastError();
}

static bool lut_identity(const Lut256 &lut)
{
    for (int i = 0; i < 64; ++i) {
        u32 b = 4u * i;
        if (lut.w[i] != (b | ((b + 1) << 8) | ((b + 2) << 16) | ((b + 3) << 24))) return false;
    }
    return true;
}

hipError_t launch_decode_level(const uint8_t *grid, uint8_t *img, const Frames &f, uint32_t log2sub,
                               int interp, hipStream_t s)
{
    Lut256 none = {};
    uint8_t *g = const_cast<uint8_t *>(grid);   // read-only in the decode instantiation
    return interp == kInterpCrossed ? launch_level_t<kInterpCrossed>(img, g, f, log2sub, false, true, none, s)
                                    : launch_level_t<kInterpLeftTop>(img, g, f, log2sub, false, true, none, s);
}

hipError_t launch_encode_level(uint8_t *rec, uint8_t *grid, const Frames &f, uint32_t log2sub, int interp,
                               const Lut256 &lut, hipStream_t s)
{
    bool ident = lut_identity(lut);
    return interp == kInterpCrossed ? launch_level_t<kInterpCrossed>(rec, grid, f, log2sub, true, ident, lut, s)
                                    : launch_level_t<kInterpLeftTop>(rec, grid, f, log2sub, true, ident, lut, s);
}

hipError_t launch_gather_lattice(const uint8_t *src, const Frames &f, uint32_t k, uint8_t *dst, uint32_t sw,
                                 uint32_t sh, uint64_t dst_stride, hipStream_t s)
{
    u64 n = (u64)sw * sh;
    hipLaunchKernelGGL(k_gather_lattice, dim3((u32)((n + 255) / 256), batch_grid_y(f)), dim3(256), 0, s, src, f,
                       k, dst, sw, sh, dst_stride);
    return hipGetLastError();
}

bool lattice_pyramid_fits(uint32_t sw, uint32_t sh, size_t batch)
{
    // One workgroup per frame: worth it for small planes only (64 x 64 per 4K frame: three launches become one).  At
    // 256 x 256 -- a lone 16384^2 frame -- the single workgroup takes longer than the tile kernels it would replace
    // (measured 164 / 140 us against 139 / 114 us for the whole call), so larger planes keep the host recursion.
    const uint64_t limit = (uint64_t)HGI_KNOB(HGI_LATTICE_MAX, 8192);     // largest plane (points) the kernel takes
    return (uint64_t)sw * sh <= limit && (uint64_t)sw * sh <= 64 * 1024 && batch <= 0x7FFFFFFFull;
}

hipError_t launch_lattice_pyramid(const uint8_t *src, const Frames &f, uint32_t k, uint32_t up, int interp, const Lut256 &lut,
                                  bool ident, bool encode, uint8_t *out_q, uint8_t *out_rec, uint32_t sw, uint32_t sh,
                                  uint64_t dst_stride, hipStream_t s)
{
    const size_t n_al = ((size_t)sw * sh + 15) & ~(size_t)15, lds = 2 * n_al + 256;
    const dim3 grid(f.batch), block(1024);
    // more than 64 KiB of dynamic LDS has to be asked for once per kernel AND PER DEVICE (one process may drive several:
    // include/hgi.h, "distinct ctxs are independent"); a bit per device ordinal, set after the attribute call succeeded
#define HGI_LATTICE(I, E)                                                                                                  \
    do {                                                                                                                    \
        static std::atomic<unsigned long long> big_lds{0};                                                                  \
        int dev_ = 0;                                                                                                       \
        if (lds > 64 * 1024 && hipGetDevice(&dev_) == hipSuccess && !((big_lds.load() >> (dev_ & 63)) & 1ull)) {            \
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lattice_pyramid<I, E>),                   \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                    \
            if (e_ != hipSuccess) return e_;                                                                                \
            big_lds.fetch_or(1ull << (dev_ & 63));                                                                          \
        }                                                                                                                   \
        hipLaunchKernelGGL((k_lattice_pyramid<I, E>), grid, block, lds, s, src, f, k, up, lut, ident ? 1u : 0u, out_q, out_rec, \
                           sw, sh, dst_stride);                                                                             \
    } while (0)
    if (interp == kInterpCrossed) {
        if (encode) HGI_LATTICE(kInterpCrossed, true); else HGI_LATTICE(kInterpCrossed, false);
    } else {
        if (encode) HGI_LATTICE(kInterpLeftTop, true); else HGI_LATTICE(kInterpLeftTop, false);
    }
#undef HGI_LATTICE
    return hipGetLastError();
}

hipError_t launch_synth(int kind, uint64_t seed, uint64_t first_frame, uint8_t *out, const Frames &f, hipStream_t s)
{
    u64 n = (u64)f.width * f.height;
    u64 blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_synth, dim3((u32)blocks, batch_grid_y(f)), dim3(256), 0, s, kind, seed, first_frame, out, f);
    return hipGetLastError();
}

hipError_t launch_copy(const uint8_t *src, uint8_t *dst, size_t n, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    if (ptr16(src) && ptr16(dst) && n % 16 == 0) {
        size_t n16 = n / 16;
        size_t blocks = (n16 + 255) / 256;
        if (blocks > (1u << 30)) blocks = 1u << 30;
        hipLaunchKernelGGL(k_copy16, dim3((u32)blocks), dim3(256), 0, s, reinterpret_cast<const copy_v4u *>(src),
                           reinterpret_cast<copy_v4u *>(dst), n16);
    } else {
        size_t blocks = (n + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(k_copy1, dim3((u32)blocks), dim3(256), 0, s, src, dst, n);
    }
    return hipGetLastError();
}

hipError_t launch_histogram(const uint8_t *src, const Frames &f, unsigned long long *hist, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(hist, 0, 256 * sizeof(unsigned long long) * f.batch, s);
    if (e != hipSuccess) return e;
    const u64 n = (u64)f.width * f.height;
    if (n == 0) return hipSuccess;
    // a block sweeps >= 64 KiB so that zeroing and folding its copies stays small; at most ~8 blocks per CU in all
    u64 blocks = (n + 65535) / 65536;
    const u64 per_frame_cap = 2048 / (batch_grid_y(f) ? batch_grid_y(f) : 1) + 1;
    if (blocks > per_frame_cap) blocks = per_frame_cap;
    hipLaunchKernelGGL(k_histogram, dim3((u32)blocks, batch_grid_y(f)), dim3(256), 0, s, src, f, hist);
    return hipGetLastError();
}

hipError_t launch_diff_stats(const uint8_t *a, const uint8_t *b, const Frames &f, unsigned long long *out,
                             hipStream_t s)
{
    hipError_t e = hipMemsetAsync(out, 0, 3 * sizeof(unsigned long long) * f.batch, s);
    if (e != hipSuccess) return e;
    u64 n = (u64)f.width * f.height;
    u64 blocks = (n + 255) / 256;
    if (blocks == 0) return hipSuccess;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_diff_stats, dim3((u32)blocks, batch_grid_y(f)), dim3(256), 0, s, a, b, f, out);
    return hipGetLastError();
}

}  // namespace hgi
