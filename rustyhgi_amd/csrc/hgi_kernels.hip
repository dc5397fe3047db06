// gfx950 (MI355X / CDNA4) kernels of the HGI encode/decode core.
//
// Algorithm (reference, paths relative to pl0q1n/RustyHGI):
//   src/encoder.rs:39-71   closed-loop encode: predict -> residual -> quantize -> overflow fallback
//   src/decoder.rs:18-46   decode: predict + residual (wrapping)
//   src/interpolator.rs    LeftTop (:15-28), Crossed (:57-90, prediction :41-55, OOB corner -> 0 :75-82)
//   src/utils.rs:12-41     per-level pixel set: coordinates = 0 (mod sub), not both = 0 (mod 2*sub)
//
// Two device implementations:
//   * level-wise: one launch per level, one thread per step-cell, global-memory stencil.  Simple,
//     strided; kept as the on-device cross-check and for measuring single passes.
//   * fused: one 256-thread workgroup owns a 256x64 tile, stages it (plus a sparse one-sided
//     halo) in LDS with 16-B coalesced row loads, runs the last k <= 6 levels of the pyramid in
//     LDS, and streams the finest level (75 % of the pixels) from LDS through packed-u8 VALU
//     arithmetic (v_lerp_u8 / v_perm_b32 / v_alignbyte) straight to 16-B global stores.
//     Every image byte is read from HBM once and every output byte written once.
//
// All arithmetic is u8/integer; there is no MFMA-shaped work on this path.
#include "hgi_kernels.h"

namespace hgi {
namespace {

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

constexpr int TW = kTileW;
constexpr int TH = kTileH;
constexpr int NT = kThreads;
constexpr int CH = TW / 16;        // 16-B chunks per tile row
constexpr int HR = 8;              // reserved halo rows / columns (>= kFusedMaxLevels)
constexpr int S = TW + 16;         // LDS pitch, full-resolution plane (halo columns live at TW + idx)
constexpr int R = TH + HR;         // LDS rows (halo rows live at TH + idx)
constexpr int S2 = TW / 2 + 16;    // LDS pitch, half-resolution plane (encode: reconstruction lattice)
constexpr int R2 = TH / 2 + HR;

static_assert(TW == 256 && (TH & (TH - 1)) == 0 && (1 << kFusedMaxLevels) <= TH, "tile geometry");
static_assert(S % 16 == 0 && S2 % 8 == 0, "LDS pitches keep vector alignment");

// ---------------------------------------------------------------------------------------------
// predictors
// ---------------------------------------------------------------------------------------------
// Names follow src/interpolator.rs:84-89: lt=(x0,y0) rt=(x0,y0+step) lb=(x0+step,y0) rb=(x0+step,y0+step)
template <int INTERP>
__device__ __forceinline__ u32 pred1(u32 lt, u32 rt, u32 lb, u32 rb)
{
    if (INTERP == kInterpLeftTop) return lt;                     // src/interpolator.rs:26
    u32 left = (lt + lb + 1) >> 1, right = (rb + rt + 1) >> 1;   // :46-47
    u32 top = (rt + lt + 1) >> 1, bot = (rb + lb + 1) >> 1;      // :48-49
    return (left + right + top + bot) >> 2;                      // :51
}

// Four predictions at once, one per byte.  v_lerp_u8: D.b = (S0.b + S1.b + (S2.b & 1)) >> 1.
// (L+R+T+B)>>2 == lerp(lerp(L,R,0), lerp(T,B,0), (L^R)&(T^B)) -- the two discarded halves add
// up to one whole only when both pair sums are odd.
__device__ __forceinline__ u32 pred4_crossed(u32 lt, u32 rt, u32 lb, u32 rb)
{
    const u32 one = 0x01010101u;
    u32 l = __builtin_amdgcn_lerp(lt, lb, one), r = __builtin_amdgcn_lerp(rb, rt, one);
    u32 t = __builtin_amdgcn_lerp(rt, lt, one), b = __builtin_amdgcn_lerp(rb, lb, one);
    u32 u = __builtin_amdgcn_lerp(l, r, 0u), v = __builtin_amdgcn_lerp(t, b, 0u);
    return __builtin_amdgcn_lerp(u, v, (l ^ r) & (t ^ b));
}

// Eight cells of one row pair: c/f = corner bytes of the upper/lower lattice row (c.x byte i =
// corner of cell i, c8/f8 = ninth corner in byte 0).  P0 = predictions of cells 0-3, P1 = 4-7.
template <int INTERP>
__device__ __forceinline__ void pred8(uint2 c, u32 c8, uint2 f, u32 f8, u32 &P0, u32 &P1)
{
    if (INTERP == kInterpLeftTop) {
        P0 = c.x;
        P1 = c.y;
        return;
    }
    u32 cn0 = __builtin_amdgcn_alignbyte(c.y, c.x, 1), cn1 = __builtin_amdgcn_alignbyte(c8, c.y, 1);
    u32 fn0 = __builtin_amdgcn_alignbyte(f.y, f.x, 1), fn1 = __builtin_amdgcn_alignbyte(f8, f.y, 1);
    P0 = pred4_crossed(c.x, f.x, cn0, fn0);
    P1 = pred4_crossed(c.y, f.y, cn1, fn1);
}

// byte-wise add / sub modulo 256 on four packed bytes
__device__ __forceinline__ u32 add4(u32 a, u32 b)
{
    return ((a & 0x7f7f7f7fu) + (b & 0x7f7f7f7fu)) ^ ((a ^ b) & 0x80808080u);
}
__device__ __forceinline__ u32 sub4(u32 a, u32 b)
{
    return ((a | 0x80808080u) - (b & 0x7f7f7f7fu)) ^ ((a ^ ~b) & 0x80808080u);
}

// src/encoder.rs:53-60 for one pixel: residual, quantize, overflow fallback.
template <bool IDENT>
__device__ __forceinline__ u32 quant1(u32 a, u32 p, const u8 *slut)
{
    u32 d = (a - p) & 255u;                    // :53 wrapping_sub
    if (IDENT) return d;                       // identity table: q == d, fallback can never fire
    u32 q = slut[d];                           // :54
    bool overflow = (p + q) > 255u;            // :56
    bool expected = (p + d) > 255u;            // :57
    return overflow != expected ? d : q;       // :58-60
}

// Four pixels packed in a dword (a = originals, p = predictions).
template <bool IDENT>
__device__ __forceinline__ u32 quant4(u32 a, u32 p, const u8 *slut)
{
    if (IDENT) return sub4(a, p);
    u32 out = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        out |= quant1<false>((a >> (8 * i)) & 255u, (p >> (8 * i)) & 255u, slut) << (8 * i);
    return out;
}

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
// fused path: tile bookkeeping
// ---------------------------------------------------------------------------------------------
// Halo coordinates.  Beyond the tile only offsets {0, 4, 8, ..., 2^k} are ever touched (level
// `sub` reads corners at offset 2*sub and writes at offset sub, sub >= 4), so halo
// rows/columns are stored compactly at index hmap(offset).
__device__ __forceinline__ int hmap(int off) { return off ? 30 - __clz(off) : 0; }   // 4->1, 8->2 ...
__device__ __forceinline__ int hoff(int idx) { return idx ? 2 << idx : 0; }           // 1->4, 2->8 ...
__device__ __forceinline__ int lcol(int x) { return x < TW ? x : TW + hmap(x - TW); }
__device__ __forceinline__ int lrow(int y) { return y < TH ? y : TH + hmap(y - TH); }
__device__ __forceinline__ int lcol2(int x) { return x < TW ? x >> 1 : TW / 2 + hmap(x - TW); }
__device__ __forceinline__ int lrow2(int y) { return y < TH ? y >> 1 : TH / 2 + hmap(y - TH); }

struct Tile {
    u32 frame, X0, Y0;
};

// XCD-aware block -> tile map.  Workgroups are dealt round-robin over the 8 XCDs (b % 8), so
// giving XCD x the x-th contiguous eighth of the row-major tile list makes x-neighbours (which
// share halo lines) and consecutive rows land in the same XCD's L2.  Speed only, never correctness.
__device__ __forceinline__ Tile tile_of_block(u32 ntiles, u32 tiles_x, u32 tiles_y)
{
    u32 b = blockIdx.x;
    u32 q = ntiles >> 3, r = ntiles & 7u, xcd = b & 7u, i = b >> 3;
    u32 t = xcd * q + (xcd < r ? xcd : r) + i;
    u32 tpf = tiles_x * tiles_y;
    Tile tl;
    tl.frame = t / tpf;
    u32 tt = t - tl.frame * tpf;
    u32 ty = tt / tiles_x;
    tl.X0 = (tt - ty * tiles_x) * TW;
    tl.Y0 = ty * TH;
    return tl;
}

// 16 image bytes at (gx, gy); zero beyond the image (src/interpolator.rs:75-82).
__device__ __forceinline__ uint4 load16(const u8 *__restrict__ fr, u32 W, u32 H, u32 gx, u32 gy,
                                        bool aligned)
{
    uint4 v = make_uint4(0, 0, 0, 0);
    if (gy < H && gx < W) {
        const u8 *p = fr + (size_t)gy * W + gx;
        if (aligned) {
            v = *reinterpret_cast<const uint4 *>(p);
        } else {
            u32 w[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (gx + j < W) w[j >> 2] |= (u32)p[j] << (8 * (j & 3));
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    return v;
}

__device__ __forceinline__ void store16(u8 *__restrict__ fr, u32 W, u32 gx, u32 gy, uint4 v,
                                        bool aligned)
{
    u8 *p = fr + (size_t)gy * W + gx;
    if (aligned) {
        *reinterpret_cast<uint4 *>(p) = v;
    } else {
        u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (gx + j < W) p[j] = (u8)(w[j >> 2] >> (8 * (j & 3)));
    }
}

// Stage tile + halo of `src` in LDS.  nh = number of halo rows/columns in use.
__device__ __forceinline__ void stage_tile(u8 *buf, const u8 *__restrict__ fr, u32 W, u32 H, Tile tl,
                                           int nh, bool aligned)
{
    const int tid = threadIdx.x;
    // body rows: TH x CH chunks, all loads of a thread in flight before the first LDS write
    uint4 v[TH * CH / NT];
#pragma unroll
    for (int j = 0; j < TH * CH / NT; ++j) {
        int i = tid + j * NT;
        v[j] = load16(fr, W, H, tl.X0 + 16 * (i & (CH - 1)), tl.Y0 + (i >> 4), aligned);
    }
    // halo rows TH + {0,4,8,..}: full-width chunk loads
    uint4 hv = make_uint4(0, 0, 0, 0);
    const bool has_hrow = tid < nh * CH;
    if (has_hrow) hv = load16(fr, W, H, tl.X0 + 16 * (tid & (CH - 1)), tl.Y0 + TH + hoff(tid >> 4), aligned);
#pragma unroll
    for (int j = 0; j < TH * CH / NT; ++j) {
        int i = tid + j * NT;
        *reinterpret_cast<uint4 *>(buf + (i >> 4) * S + 16 * (i & (CH - 1))) = v[j];
    }
    if (has_hrow) *reinterpret_cast<uint4 *>(buf + (TH + (tid >> 4)) * S + 16 * (tid & (CH - 1))) = hv;
    // halo columns TW + {0,4,8,..} (and the halo x halo corner block): byte gathers.  Column
    // offset `off` is only ever touched on rows = 0 (mod max(off, 2)).
    for (int i = tid; i < HR * (TH + nh); i += NT) {
        int hc = i & (HR - 1), rr = i >> 3;
        if (hc >= nh) continue;
        int off = hoff(hc);
        int y = rr < TH ? rr : TH + hoff(rr - TH);
        if (rr < TH && (y & ((off ? off : 2) - 1))) continue;
        u32 gx = tl.X0 + TW + off, gy = tl.Y0 + y;
        buf[rr * S + TW + hc] = (gx < W && gy < H) ? fr[(size_t)gy * W + gx] : (u8)0;
    }
}

// ---------------------------------------------------------------------------------------------
// fused decode
// ---------------------------------------------------------------------------------------------
// One level (sub >= 2) of the tile in LDS, in place: buf holds residuals where a pixel is not yet
// decoded and reconstructed values where it is.
template <int INTERP>
__device__ __forceinline__ void dec_level_lds(u8 *buf, int s, Tile tl, u32 W, u32 H)
{
    const int tid = threadIdx.x;
    const int step = 2 * s;
    const int lstep = 31 - __clz(step);
    const int ncx = TW >> lstep, ncy = TH >> lstep, lncx = 31 - __clz(ncx);
    // cells inside the tile: natural LDS coordinates (x0 + step <= TW and y0 + step <= TH map to themselves)
    for (int i = tid; i < ncx * ncy; i += NT) {
        int x0 = (i & (ncx - 1)) << lstep, y0 = (i >> lncx) << lstep;
        if (tl.X0 + x0 >= W || tl.Y0 + y0 >= H) continue;
        u8 *c = buf + y0 * S + x0;
        u32 p = pred1<INTERP>(c[0], c[step * S], c[step], c[step * S + step]);
        bool xin = tl.X0 + x0 + s < W, yin = tl.Y0 + y0 + s < H;
        if (xin) c[s] = (u8)(c[s] + p);
        if (yin) c[s * S] = (u8)(c[s * S] + p);
        if (xin && yin) c[s * S + s] = (u8)(c[s * S + s] + p);
    }
    // halo cells: column x0 == TW (cy = 0..ncy) and row y0 == TH (cx = 0..ncx-1).  They recompute,
    // bit-identically, what the right / lower neighbour tiles compute for themselves.  Level `sub`
    // halo pixels are needed up to offset sub (sub >= 4); at sub == 2 only those on the tile edge.
    const int lim = s >= 4 ? s : 0;
    for (int j = tid; j < ncx + ncy + 1; j += NT) {
        int x0 = j <= ncy ? TW : (j - ncy - 1) << lstep;
        int y0 = j <= ncy ? j << lstep : TH;
        if (tl.X0 + x0 >= W || tl.Y0 + y0 >= H) continue;
        int r0 = lrow(y0) * S, r1 = lrow(y0 + step) * S, c0 = lcol(x0), c1 = lcol(x0 + step);
        u32 p = pred1<INTERP>(buf[r0 + c0], buf[r1 + c0], buf[r0 + c1], buf[r1 + c1]);
        bool xin = x0 + s <= TW + lim && tl.X0 + x0 + s < W;
        bool yin = y0 + s <= TH + lim && tl.Y0 + y0 + s < H;
        int rs = lrow(y0 + s) * S, cs = lcol(x0 + s);
        if (xin) buf[r0 + cs] = (u8)(buf[r0 + cs] + p);
        if (yin) buf[rs + c0] = (u8)(buf[rs + c0] + p);
        if (xin && yin) buf[rs + cs] = (u8)(buf[rs + cs] + p);
    }
}

template <int INTERP, bool SEEDED>
__global__ __launch_bounds__(NT) void k_dec_fused(const u8 *__restrict__ src, u8 *__restrict__ dst,
                                                  Frames f, u32 k, Seeds sd, u32 tiles_x,
                                                  u32 tiles_y, u32 ntiles, u32 aligned)
{
    __shared__ __attribute__((aligned(16))) u8 buf[R * S];
    const int tid = threadIdx.x;
    const Tile tl = tile_of_block(ntiles, tiles_x, tiles_y);
    const u32 W = f.width, H = f.height;
    const u8 *fr = src + (size_t)tl.frame * f.frame_stride;
    u8 *out = dst + (size_t)tl.frame * f.frame_stride;
    const int nh = k >= 2 ? (int)k : 1;

    stage_tile(buf, fr, W, H, tl, nh, aligned != 0);
    __syncthreads();
    if (SEEDED) {
        // lattice points = 0 (mod 2^k) come from the already decoded coarser pyramid
        const int ext = k >= 2 ? 2 : 1;   // offset 2^k beyond the tile is only ever read for k >= 2
        const int nbx = (TW >> k) + ext, nby = (TH >> k) + ext;
        const u8 *sp = sd.rec + (size_t)tl.frame * sd.stride;
        for (int i = tid; i < nbx * nby; i += NT) {
            int by = i / nbx, bx = i - by * nbx;
            u32 sx = (tl.X0 >> k) + bx, sy = (tl.Y0 >> k) + by;
            u8 v = (sx < sd.sw && sy < sd.sh) ? sp[(size_t)sy * sd.sw + sx] : (u8)0;
            buf[lrow(by << k) * S + lcol(bx << k)] = v;
        }
        __syncthreads();
    }
    for (int s = 1 << (k - 1); s >= 2; s >>= 1) {
        dec_level_lds<INTERP>(buf, s, tl, W, H);
        __syncthreads();
    }
    // finest level: 16 px x 2 rows per lane, LDS -> packed VALU -> 16-B global stores
#pragma unroll 2
    for (int i = tid; i < (TH / 2) * CH; i += NT) {
        const int y = 2 * (i >> 4), x = 16 * (i & (CH - 1));
        const u32 gx = tl.X0 + x, gy = tl.Y0 + y;
        if (gx >= W || gy >= H) continue;
        const u8 *r0 = buf + y * S + x;
        uint4 E = *reinterpret_cast<const uint4 *>(r0);
        uint4 O = *reinterpret_cast<const uint4 *>(r0 + S);
        uint4 F = *reinterpret_cast<const uint4 *>(r0 + 2 * S);
        u32 e16 = r0[16], f16 = r0[2 * S + 16];
        uint2 c, fl;
        c.x = __builtin_amdgcn_perm(E.y, E.x, 0x06040200u);
        c.y = __builtin_amdgcn_perm(E.w, E.z, 0x06040200u);
        fl.x = __builtin_amdgcn_perm(F.y, F.x, 0x06040200u);
        fl.y = __builtin_amdgcn_perm(F.w, F.z, 0x06040200u);
        u32 P0, P1;
        pred8<INTERP>(c, e16, fl, f16, P0, P1);
        u32 pp0 = __builtin_amdgcn_perm(P0, P0, 0x01010000u), pp1 = __builtin_amdgcn_perm(P0, P0, 0x03030202u);
        u32 pp2 = __builtin_amdgcn_perm(P1, P1, 0x01010000u), pp3 = __builtin_amdgcn_perm(P1, P1, 0x03030202u);
        const u32 odd = 0xFF00FF00u;
        uint4 o0 = make_uint4(add4(E.x, pp0 & odd), add4(E.y, pp1 & odd), add4(E.z, pp2 & odd), add4(E.w, pp3 & odd));
        uint4 o1 = make_uint4(add4(O.x, pp0), add4(O.y, pp1), add4(O.z, pp2), add4(O.w, pp3));
        store16(out, W, gx, gy, o0, aligned != 0);
        if (gy + 1 < H) store16(out, W, gx, gy + 1, o1, aligned != 0);
    }
}

// ---------------------------------------------------------------------------------------------
// fused encode
// ---------------------------------------------------------------------------------------------
// LDS planes: buf  = originals where a pixel is not yet coded, residuals (final output) where it is;
//             rbuf = reconstruction of the even/even lattice at half resolution -- the only
//                    reconstructed values a finer level ever reads (corners are = 0 mod 2*sub).
template <int INTERP, bool IDENT>
__device__ __forceinline__ void enc_level_lds(u8 *buf, u8 *rbuf, const u8 *slut, int s, Tile tl, u32 W, u32 H)
{
    const int tid = threadIdx.x;
    const int step = 2 * s, hs = s >> 1;
    const int lstep = 31 - __clz(step);
    const int ncx = TW >> lstep, ncy = TH >> lstep, lncx = 31 - __clz(ncx);
    for (int i = tid; i < ncx * ncy; i += NT) {
        int x0 = (i & (ncx - 1)) << lstep, y0 = (i >> lncx) << lstep;
        if (tl.X0 + x0 >= W || tl.Y0 + y0 >= H) continue;
        u8 *c = buf + y0 * S + x0;
        u8 *rc = rbuf + (y0 >> 1) * S2 + (x0 >> 1);
        u32 p = pred1<INTERP>(rc[0], rc[s * S2], rc[s], rc[s * S2 + s]);
        bool xin = tl.X0 + x0 + s < W, yin = tl.Y0 + y0 + s < H;
        if (xin) {
            u32 q = quant1<IDENT>(c[s], p, slut);
            c[s] = (u8)q;
            rc[hs] = (u8)(p + q);
        }
        if (yin) {
            u32 q = quant1<IDENT>(c[s * S], p, slut);
            c[s * S] = (u8)q;
            rc[hs * S2] = (u8)(p + q);
        }
        if (xin && yin) {
            u32 q = quant1<IDENT>(c[s * S + s], p, slut);
            c[s * S + s] = (u8)q;
            rc[hs * S2 + hs] = (u8)(p + q);
        }
    }
    const int lim = s >= 4 ? s : 0;
    for (int j = tid; j < ncx + ncy + 1; j += NT) {
        int x0 = j <= ncy ? TW : (j - ncy - 1) << lstep;
        int y0 = j <= ncy ? j << lstep : TH;
        if (tl.X0 + x0 >= W || tl.Y0 + y0 >= H) continue;
        int q0 = lrow2(y0) * S2, q1 = lrow2(y0 + step) * S2, d0 = lcol2(x0), d1 = lcol2(x0 + step);
        u32 p = pred1<INTERP>(rbuf[q0 + d0], rbuf[q1 + d0], rbuf[q0 + d1], rbuf[q1 + d1]);
        bool xin = x0 + s <= TW + lim && tl.X0 + x0 + s < W;
        bool yin = y0 + s <= TH + lim && tl.Y0 + y0 + s < H;
        int r0 = lrow(y0) * S, rs = lrow(y0 + s) * S, c0 = lcol(x0), cs = lcol(x0 + s);
        int qs = lrow2(y0 + s) * S2, ds = lcol2(x0 + s);
        if (xin) {
            u32 q = quant1<IDENT>(buf[r0 + cs], p, slut);
            buf[r0 + cs] = (u8)q;
            rbuf[q0 + ds] = (u8)(p + q);
        }
        if (yin) {
            u32 q = quant1<IDENT>(buf[rs + c0], p, slut);
            buf[rs + c0] = (u8)q;
            rbuf[qs + d0] = (u8)(p + q);
        }
        if (xin && yin) {
            u32 q = quant1<IDENT>(buf[rs + cs], p, slut);
            buf[rs + cs] = (u8)q;
            rbuf[qs + ds] = (u8)(p + q);
        }
    }
}

template <int INTERP, bool IDENT, bool SEEDED>
__global__ __launch_bounds__(NT) void k_enc_fused(const u8 *__restrict__ src, u8 *__restrict__ dst,
                                                  Frames f, u32 k, Lut256 lut, Seeds sd, u32 tiles_x,
                                                  u32 tiles_y, u32 ntiles, u32 aligned)
{
    __shared__ __attribute__((aligned(16))) u8 buf[R * S];
    __shared__ __attribute__((aligned(16))) u8 rbuf[R2 * S2];
    __shared__ __attribute__((aligned(16))) u8 slut[256];
    const int tid = threadIdx.x;
    const Tile tl = tile_of_block(ntiles, tiles_x, tiles_y);
    const u32 W = f.width, H = f.height;
    const u8 *fr = src + (size_t)tl.frame * f.frame_stride;
    u8 *out = dst + (size_t)tl.frame * f.frame_stride;
    const int nh = k >= 2 ? (int)k : 1;

    if (!IDENT && tid < 64) reinterpret_cast<u32 *>(slut)[tid] = lut.w[tid];
    // lattice points outside the image must read as 0 (src/interpolator.rs:75-82) and are never written
    for (int i = tid; i < R2 * S2 / 16; i += NT) reinterpret_cast<uint4 *>(rbuf)[i] = make_uint4(0, 0, 0, 0);
    stage_tile(buf, fr, W, H, tl, nh, aligned != 0);
    __syncthreads();
    {
        // lattice points = 0 (mod 2^k): reconstruction == original (src/encoder.rs:26-37), or the
        // coarser pyramid's reconstruction + residuals when this launch is the lower part of a
        // deeper pyramid.
        const int ext = k >= 2 ? 2 : 1;   // offset 2^k beyond the tile is only ever read for k >= 2
        const int nbx = (TW >> k) + ext, nby = (TH >> k) + ext;
        const u8 *sr = SEEDED ? sd.rec + (size_t)tl.frame * sd.stride : nullptr;
        const u8 *sq = SEEDED ? sd.q + (size_t)tl.frame * sd.stride : nullptr;
        for (int i = tid; i < nbx * nby; i += NT) {
            int by = i / nbx, bx = i - by * nbx;
            int li = lrow(by << k) * S + lcol(bx << k);
            u8 rv = buf[li];
            if (SEEDED) {
                u32 sx = (tl.X0 >> k) + bx, sy = (tl.Y0 >> k) + by;
                bool in = sx < sd.sw && sy < sd.sh;
                rv = in ? sr[(size_t)sy * sd.sw + sx] : (u8)0;
                buf[li] = in ? sq[(size_t)sy * sd.sw + sx] : (u8)0;
            }
            rbuf[lrow2(by << k) * S2 + lcol2(bx << k)] = rv;
        }
    }
    __syncthreads();
    for (int s = 1 << (k - 1); s >= 2; s >>= 1) {
        enc_level_lds<INTERP, IDENT>(buf, rbuf, slut, s, tl, W, H);
        __syncthreads();
    }
    // finest level: corners come packed from rbuf, originals from buf; residuals go straight to HBM
#pragma unroll 2
    for (int i = tid; i < (TH / 2) * CH; i += NT) {
        const int y = 2 * (i >> 4), x = 16 * (i & (CH - 1));
        const u32 gx = tl.X0 + x, gy = tl.Y0 + y;
        if (gx >= W || gy >= H) continue;
        const u8 *r0 = buf + y * S + x;
        const u8 *c0 = rbuf + (y >> 1) * S2 + (x >> 1);
        uint4 E = *reinterpret_cast<const uint4 *>(r0);
        uint4 O = *reinterpret_cast<const uint4 *>(r0 + S);
        uint2 c = *reinterpret_cast<const uint2 *>(c0);
        uint2 fl = *reinterpret_cast<const uint2 *>(c0 + S2);
        u32 c8 = c0[8], f8 = c0[S2 + 8];
        u32 P0, P1;
        pred8<INTERP>(c, c8, fl, f8, P0, P1);
        u32 pp0 = __builtin_amdgcn_perm(P0, P0, 0x01010000u), pp1 = __builtin_amdgcn_perm(P0, P0, 0x03030202u);
        u32 pp2 = __builtin_amdgcn_perm(P1, P1, 0x01010000u), pp3 = __builtin_amdgcn_perm(P1, P1, 0x03030202u);
        // row y: only the odd columns are new; gather them (cell i <-> byte i of P0/P1)
        u32 a0 = __builtin_amdgcn_perm(E.y, E.x, 0x07050301u), a1 = __builtin_amdgcn_perm(E.w, E.z, 0x07050301u);
        u32 q0 = quant4<IDENT>(a0, P0, slut), q1 = quant4<IDENT>(a1, P1, slut);
        uint4 o0 = make_uint4(__builtin_amdgcn_perm(q0, E.x, 0x05020400u), __builtin_amdgcn_perm(q0, E.y, 0x07020600u),
                              __builtin_amdgcn_perm(q1, E.z, 0x05020400u), __builtin_amdgcn_perm(q1, E.w, 0x07020600u));
        store16(out, W, gx, gy, o0, aligned != 0);
        if (gy + 1 < H) {
            uint4 o1 = make_uint4(quant4<IDENT>(O.x, pp0, slut), quant4<IDENT>(O.y, pp1, slut),
                                  quant4<IDENT>(O.z, pp2, slut), quant4<IDENT>(O.w, pp3, slut));
            store16(out, W, gx, gy + 1, o1, aligned != 0);
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

__global__ void k_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}
__global__ void k_copy1(const u8 *__restrict__ src, u8 *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
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

struct FusedGeom {
    u32 tiles_x, tiles_y, ntiles, aligned;
    bool ok;
};

static FusedGeom fused_geom(const void *a, const void *b, const Frames &f)
{
    FusedGeom g;
    g.tiles_x = (f.width + TW - 1) / TW;
    g.tiles_y = (f.height + TH - 1) / TH;
    u64 nt = (u64)g.tiles_x * g.tiles_y * f.batch;
    g.ok = nt > 0 && nt < (1ull << 31);
    g.ntiles = (u32)nt;
    g.aligned = (f.width % 16 == 0 && f.frame_stride % 16 == 0 && ptr16(a) && ptr16(b)) ? 1u : 0u;
    return g;
}

hipError_t launch_decode_fused(const uint8_t *grid, uint8_t *img, const Frames &f, uint32_t k, int interp,
                               const Seeds *seeds, hipStream_t s)
{
    FusedGeom g = fused_geom(grid, img, f);
    if (!g.ok || k < 1 || k > (u32)kFusedMaxLevels) return hipErrorInvalidValue;
    Seeds sd = seeds ? *seeds : Seeds{nullptr, nullptr, 0, 0, 0};
    dim3 gr(g.ntiles), b(NT);
#define HGI_DEC(I, SE) \
    hipLaunchKernelGGL((k_dec_fused<I, SE>), gr, b, 0, s, grid, img, f, k, sd, g.tiles_x, g.tiles_y, g.ntiles, g.aligned)
    if (interp == kInterpCrossed) {
        if (seeds) HGI_DEC(kInterpCrossed, true); else HGI_DEC(kInterpCrossed, false);
    } else {
        if (seeds) HGI_DEC(kInterpLeftTop, true); else HGI_DEC(kInterpLeftTop, false);
    }
#undef HGI_DEC
    return hipGetLastError();
}

hipError_t launch_encode_fused(const uint8_t *img, uint8_t *grid, const Frames &f, uint32_t k, int interp,
                               const Lut256 &lut, bool ident, const Seeds *seeds, hipStream_t s)
{
    FusedGeom g = fused_geom(img, grid, f);
    if (!g.ok || k < 1 || k > (u32)kFusedMaxLevels) return hipErrorInvalidValue;
    Seeds sd = seeds ? *seeds : Seeds{nullptr, nullptr, 0, 0, 0};
    dim3 gr(g.ntiles), b(NT);
#define HGI_ENC(I, ID, SE) \
    hipLaunchKernelGGL((k_enc_fused<I, ID, SE>), gr, b, 0, s, img, grid, f, k, lut, sd, g.tiles_x, g.tiles_y, g.ntiles, g.aligned)
#define HGI_ENC_I(I)                                                          \
    do {                                                                      \
        if (ident) { if (seeds) HGI_ENC(I, true, true); else HGI_ENC(I, true, false); } \
        else       { if (seeds) HGI_ENC(I, false, true); else HGI_ENC(I, false, false); } \
    } while (0)
    if (interp == kInterpCrossed) HGI_ENC_I(kInterpCrossed); else HGI_ENC_I(kInterpLeftTop);
#undef HGI_ENC_I
#undef HGI_ENC
    return hipGetLastError();
}

hipError_t launch_gather_lattice(const uint8_t *src, const Frames &f, uint32_t k, uint8_t *dst, uint32_t sw,
                                 uint32_t sh, uint64_t dst_stride, hipStream_t s)
{
    u64 n = (u64)sw * sh;
    hipLaunchKernelGGL(k_gather_lattice, dim3((u32)((n + 255) / 256), batch_grid_y(f)), dim3(256), 0, s, src, f,
                       k, dst, sw, sh, dst_stride);
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
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(k_copy16, dim3((u32)blocks), dim3(256), 0, s, reinterpret_cast<const uint4 *>(src),
                           reinterpret_cast<uint4 *>(dst), n16);
    } else {
        size_t blocks = (n + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(k_copy1, dim3((u32)blocks), dim3(256), 0, s, src, dst, n);
    }
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
