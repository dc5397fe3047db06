// gfx950 (MI355X / CDNA4) fused HGI kernels: all levels of a tile in one launch, LDS-resident.
//
// Reference algorithm (paths relative to pl0q1n/RustyHGI):
//   src/encoder.rs:39-71, src/decoder.rs:18-46, src/interpolator.rs:15-28 / :41-90, src/utils.rs:12-41
//
// Work decomposition
//   * ONE WAVE (64 lanes) owns one 128x64 tile.  A tile row is one 128-B line = 8 lanes x 16 B, so a
//     wave-wide load/store instruction moves 8 full lines.  No workgroup barrier exists anywhere: all
//     cross-lane traffic goes through the wave's own LDS slice, ordered by the wave's in-order LDS
//     queue; the other waves of the CU (other tiles, other phases) hide the latency.
//   * The tile plus a sparse one-sided halo (offsets {0,4,8,...,2^k} to the right and below,
//     SURVEY.md A.6) is staged in LDS once; halo pixels are recomputed bit-identically instead of
//     exchanged.  Every image byte is fetched from HBM once and every output byte written once.
//   * Levels sub >= 4 (6 % of the pixels): one lane per step-cell, byte LDS accesses.
//     Level sub == 2 (19 %): four cells per lane, packed v_lerp_u8 predictor, 16-B LDS accesses.
//     Level sub == 1 (75 %): 16 px x 2 rows per lane from LDS through packed-u8 VALU arithmetic
//     straight to 16-B buffer stores.
//   * Interior tiles (tile body inside the image) take a check-free path built on buffer loads whose
//     hardware range check returns 0 beyond the frame -- exactly the reference's out-of-image rule
//     (src/interpolator.rs:75-82).  Ragged tiles and unaligned widths take the generic path below.
//
// All arithmetic is u8/integer; there is no MFMA-shaped work on this path.
#include "hgi_dev.h"

namespace hgi {
namespace {

using namespace dev;

constexpr int TW = kTileW;
constexpr int TH = kTileH;
constexpr int NL = kThreads;       // lanes
constexpr int CH = TW / 16;        // 16-B chunks per tile row
constexpr int LCH = 3;             // log2(CH)
constexpr int HR = 8;              // reserved halo rows / columns (>= kFusedMaxLevels)
constexpr int S = TW + 16;         // LDS pitch, full-resolution plane (halo columns live at TW + idx)
constexpr int R = TH + HR;         // LDS rows (halo rows live at TH + idx)
constexpr int S2 = TW / 2 + 16;    // LDS pitch, half-resolution plane (encode: reconstruction lattice)
constexpr int R2 = TH / 2 + HR;

static_assert(NL == 64 && CH == (1 << LCH) && (TH & (TH - 1)) == 0 && (1 << kFusedMaxLevels) <= TH, "tile geometry");
static_assert(S % 16 == 0 && S2 % 8 == 0 && (R2 * S2) % 16 == 0, "LDS pitches keep vector alignment");

typedef u32 v4u __attribute__((ext_vector_type(4)));
typedef u32 v2u __attribute__((ext_vector_type(2)));

// dst.byte[K] = (dst.byte[K] + src.byte[J]) mod 256, other bytes of dst preserved (one SDWA VALU op)
#define HGI_ADDB(dst, K, src, J)                                                                    \
    asm("v_add_u16_sdwa %0, %0, %1 dst_sel:BYTE_" #K " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #K \
        " src1_sel:BYTE_" #J                                                                        \
        : "+v"(dst)                                                                                 \
        : "v"(src))

// Quantize + overflow fallback of ONE pixel in four single-issue VALU ops (src/encoder.rs:53-60):
//   d   = a - p (mod 256)                        Q_SUB   (clean byte, ready as LDS address)
//   q   = table[d]                               ds_read_u8
//   expected overflow  p + d > 255  <=>  a < p   Q_LT    (borrow of a - p)
//   actual overflow    p + q > 255  <=>  q > ~p  Q_GT    (np = 255 - p per byte)
//   residual = (expected != actual) ? d : q      Q_SEL   (s_xor + v_cndmask straight into byte K)
typedef unsigned long long lanemask;
#define Q_SUB(d, a, K, p, J)                                                                                   \
    asm("v_sub_u16_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:BYTE_" #K " src1_sel:BYTE_" #J \
        : "=v"(d) : "v"(a), "v"(p))
#define Q_LT(m, a, K, p, J) \
    asm("v_cmp_lt_u16_sdwa %0, %1, %2 src0_sel:BYTE_" #K " src1_sel:BYTE_" #J : "=s"(m) : "v"(a), "v"(p))
#define Q_GT(m, q, np, J) \
    asm("v_cmp_gt_u16_sdwa %0, %1, %2 src0_sel:BYTE_0 src1_sel:BYTE_" #J : "=s"(m) : "v"(q), "v"(np))
#define Q_SEL(out, K, mb, mc, q, d)                                                                    \
    asm("s_xor_b64 vcc, %1, %2\n\tv_cndmask_b32_sdwa %0, %3, %4, vcc dst_sel:BYTE_" #K                 \
        " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0"                                  \
        : "+v"(out) : "s"(mb), "s"(mc), "v"(q), "v"(d) : "vcc", "scc")
// four pixels: byte Kn of register On (original in, residual out) predicted by byte Jn of P
#define Q_PIX4(slut, P, NP, O0, K0, J0, O1, K1, J1, O2, K2, J2, O3, K3, J3)                              \
    do {                                                                                               \
        u32 d0_, d1_, d2_, d3_;                                                                        \
        lanemask b0_, b1_, b2_, b3_, c0_, c1_, c2_, c3_;                                               \
        Q_SUB(d0_, O0, K0, P, J0); Q_SUB(d1_, O1, K1, P, J1); Q_SUB(d2_, O2, K2, P, J2); Q_SUB(d3_, O3, K3, P, J3); \
        u32 q0_ = slut[d0_], q1_ = slut[d1_], q2_ = slut[d2_], q3_ = slut[d3_];                        \
        Q_LT(b0_, O0, K0, P, J0); Q_LT(b1_, O1, K1, P, J1); Q_LT(b2_, O2, K2, P, J2); Q_LT(b3_, O3, K3, P, J3); \
        Q_GT(c0_, q0_, NP, J0); Q_GT(c1_, q1_, NP, J1); Q_GT(c2_, q2_, NP, J2); Q_GT(c3_, q3_, NP, J3); \
        Q_SEL(O0, K0, b0_, c0_, q0_, d0_); Q_SEL(O1, K1, b1_, c1_, q1_, d1_);                          \
        Q_SEL(O2, K2, b2_, c2_, q2_, d2_); Q_SEL(O3, K3, b3_, c3_, q3_, d3_);                          \
    } while (0)
// dst.byte[KD] = (q.byte[KQ] + p.byte[J]) mod 256: the reconstruction of a freshly coded pixel
#define Q_REC(dst, KD, q, KQ, p, J)                                                                    \
    asm("v_add_u16_sdwa %0, %1, %2 dst_sel:BYTE_" #KD " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #KQ   \
        " src1_sel:BYTE_" #J                                                                           \
        : "+v"(dst) : "v"(q), "v"(p))

// ---------------------------------------------------------------------------------------------
// tile bookkeeping
// ---------------------------------------------------------------------------------------------
// Halo coordinates.  Beyond the tile only offsets {0, 4, 8, ..., 2^k} are ever touched (level `sub`
// reads corners at offset 2*sub and writes at offset sub, sub >= 4), so halo
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

// XCD-aware block -> tile map.  Workgroups are dealt round-robin over the 8 XCDs (b % 8), so giving
// XCD x the x-th contiguous eighth of the row-major tile list makes x-neighbours (which share halo
// lines) land in the same XCD's L2.  Speed only, never correctness.
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

// =============================================================================================
// GENERIC PATH (ragged tiles, unaligned widths): every access checked against the image
// =============================================================================================
// 16 image bytes at (gx, gy); zero beyond the image (src/interpolator.rs:75-82).
__device__ __forceinline__ uint4 load16(const u8 *__restrict__ fr, u32 W, u32 H, u32 gx, u32 gy, bool aligned)
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

__device__ __forceinline__ void store16(u8 *__restrict__ fr, u32 W, u32 gx, u32 gy, uint4 v, bool aligned)
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
__device__ __noinline__ void stage_tile_generic(u8 *buf, const u8 *__restrict__ fr, u32 W, u32 H, Tile tl, int nh,
                                                bool aligned)
{
    const int lane = threadIdx.x;
    for (int i = lane; i < (TH + nh) * CH; i += NL) {
        int rr = i >> LCH, c = i & (CH - 1);
        int y = rr < TH ? rr : TH + hoff(rr - TH);
        uint4 v = load16(fr, W, H, tl.X0 + 16 * c, tl.Y0 + y, aligned);
        *reinterpret_cast<uint4 *>(buf + rr * S + 16 * c) = v;
    }
    // halo columns TW + {0,4,8,..} (and the halo x halo corner block): byte gathers.  Column offset
    // `off` is only ever touched on rows = 0 (mod max(off, 2)).
    for (int i = lane; i < HR * (TH + nh); i += NL) {
        int hc = i & (HR - 1), rr = i >> 3;
        if (hc >= nh) continue;
        int off = hoff(hc);
        int y = rr < TH ? rr : TH + hoff(rr - TH);
        if (rr < TH && (y & ((off ? off : 2) - 1))) continue;
        u32 gx = tl.X0 + TW + off, gy = tl.Y0 + y;
        buf[rr * S + TW + hc] = (gx < W && gy < H) ? fr[(size_t)gy * W + gx] : (u8)0;
    }
}

// Halo cells of level `s`: column x0 == TW (cy = 0..ncy) and row y0 == TH (cx = 0..ncx-1).  They
// recompute, bit-identically, what the right / lower neighbour tiles compute for themselves.  Level
// `sub` halo pixels are needed up to offset sub (sub >= 4); at sub == 2 only those on the tile edge.
// Used by BOTH paths (the halo may leave the image even when the tile body does not).
template <int INTERP>
__device__ __forceinline__ void dec_halo_cells(u8 *buf, int s, Tile tl, u32 W, u32 H)
{
    const int step = 2 * s, lstep = 31 - __clz(step);
    const int ncx = TW >> lstep, ncy = TH >> lstep;
    const int lim = s >= 4 ? s : 0;
    for (int j = threadIdx.x; j < ncx + ncy + 1; j += NL) {
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

template <int INTERP, bool IDENT>
__device__ __forceinline__ void enc_halo_cells(u8 *buf, u8 *rbuf, const u8 *slut, int s, Tile tl, u32 W, u32 H)
{
    const int step = 2 * s, lstep = 31 - __clz(step);
    const int ncx = TW >> lstep, ncy = TH >> lstep;
    const int lim = s >= 4 ? s : 0;
    for (int j = threadIdx.x; j < ncx + ncy + 1; j += NL) {
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

// One level (sub >= 2) of the tile body in LDS, in place.  CHECK = test every pixel against the image.
template <int INTERP, bool CHECK>
__device__ __forceinline__ void dec_cells(u8 *buf, int s, Tile tl, u32 W, u32 H)
{
    const int step = 2 * s, lstep = 31 - __clz(step);
    const int ncx = TW >> lstep, ncy = TH >> lstep, lncx = 31 - __clz(ncx);
    // natural LDS coordinates: x0 + step <= TW and y0 + step <= TH map to themselves
    for (int i = threadIdx.x; i < ncx * ncy; i += NL) {
        int x0 = (i & (ncx - 1)) << lstep, y0 = (i >> lncx) << lstep;
        if (CHECK && (tl.X0 + x0 >= W || tl.Y0 + y0 >= H)) continue;
        u8 *c = buf + y0 * S + x0;
        u32 p = pred1<INTERP>(c[0], c[step * S], c[step], c[step * S + step]);
        bool xin = !CHECK || tl.X0 + x0 + s < W, yin = !CHECK || tl.Y0 + y0 + s < H;
        if (xin) c[s] = (u8)(c[s] + p);
        if (yin) c[s * S] = (u8)(c[s * S] + p);
        if (xin && yin) c[s * S + s] = (u8)(c[s * S + s] + p);
    }
}

// Encode planes: buf  = originals where a pixel is not yet coded, residuals (final output) where it is;
//                rbuf = reconstruction of the even/even lattice at half resolution -- the only
//                       reconstructed values a finer level ever reads (corners are = 0 mod 2*sub).
template <int INTERP, bool IDENT, bool CHECK>
__device__ __forceinline__ void enc_cells(u8 *buf, u8 *rbuf, const u8 *slut, int s, Tile tl, u32 W, u32 H)
{
    const int step = 2 * s, hs = s >> 1, lstep = 31 - __clz(step);
    const int ncx = TW >> lstep, ncy = TH >> lstep, lncx = 31 - __clz(ncx);
    for (int i = threadIdx.x; i < ncx * ncy; i += NL) {
        int x0 = (i & (ncx - 1)) << lstep, y0 = (i >> lncx) << lstep;
        if (CHECK && (tl.X0 + x0 >= W || tl.Y0 + y0 >= H)) continue;
        u8 *c = buf + y0 * S + x0;
        u8 *rc = rbuf + (y0 >> 1) * S2 + (x0 >> 1);
        u32 p = pred1<INTERP>(rc[0], rc[s * S2], rc[s], rc[s * S2 + s]);
        bool xin = !CHECK || tl.X0 + x0 + s < W, yin = !CHECK || tl.Y0 + y0 + s < H;
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
}

// finest level, generic: 16 px x 2 rows per lane, checked stores
template <int INTERP>
__device__ __noinline__ void dec_fine_generic(const u8 *buf, u8 *__restrict__ out, Tile tl, u32 W, u32 H, bool aligned)
{
    for (int i = threadIdx.x; i < (TH / 2) * CH; i += NL) {
        const int y = 2 * (i >> LCH), x = 16 * (i & (CH - 1));
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
        store16(out, W, gx, gy, o0, aligned);
        if (gy + 1 < H) store16(out, W, gx, gy + 1, o1, aligned);
    }
}

template <int INTERP, bool IDENT>
__device__ __noinline__ void enc_fine_generic(const u8 *buf, const u8 *rbuf, const u8 *slut, u8 *__restrict__ out,
                                              Tile tl, u32 W, u32 H, bool aligned)
{
    for (int i = threadIdx.x; i < (TH / 2) * CH; i += NL) {
        const int y = 2 * (i >> LCH), x = 16 * (i & (CH - 1));
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
        store16(out, W, gx, gy, o0, aligned);
        if (gy + 1 < H) {
            uint4 o1 = make_uint4(quant4<IDENT>(O.x, pp0, slut), quant4<IDENT>(O.y, pp1, slut),
                                  quant4<IDENT>(O.z, pp2, slut), quant4<IDENT>(O.w, pp3, slut));
            store16(out, W, gx, gy + 1, o1, aligned);
        }
    }
}

// =============================================================================================
// FAST PATH (tile body inside the image, 16-B aligned rows): no per-pixel checks
// =============================================================================================
struct Buf {
    __amdgpu_buffer_rsrc_t rs;   // frame being read  (range check -> 0 beyond width*height)
    __amdgpu_buffer_rsrc_t rd;   // frame being written
    u32 W, base;                 // base = Y0 * W + X0
};

// Tile body: 8 x 16-B loads per lane, all in flight before the first LDS write; halo rows as full
// lines; halo columns by one lane per (even) row: one 16-B load holds offsets 0/4/8, single dwords
// supply offsets 16/32/64.  Rows below the image return 0 from the buffer range check; columns right
// of the image are masked with wave-uniform tests.
__device__ __forceinline__ void stage_tile_fast(u8 *buf, const Buf &b, Tile tl, int k, int nh)
{
    const int lane = threadIdx.x, c = lane & (CH - 1), r = lane >> LCH;
    const u32 W = b.W;
    const u32 voff = b.base + r * W + 16 * c;
    v4u v[TH / 8];
#pragma unroll
    for (int j = 0; j < TH / 8; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(b.rs, voff, j * 8 * W, 0);
    const bool hrow = lane < nh * CH;
    v4u hv = {0, 0, 0, 0};
    if (hrow) hv = __builtin_amdgcn_raw_buffer_load_b128(b.rs, b.base + (TH + hoff(r)) * W + 16 * c, 0, 0);
    const bool xl = lane < TH / 2 + nh;
    const int hy = lane < TH / 2 ? 2 * lane : TH + hoff(lane - TH / 2);
    const u32 xo = b.base + hy * W + TW;
    const u32 xr = tl.X0 + TW;              // first column right of the tile
    v4u x0 = {0, 0, 0, 0};
    u32 d16 = 0, d32 = 0, d64 = 0;
    if (xl) {
        if (xr < W) x0 = __builtin_amdgcn_raw_buffer_load_b128(b.rs, xo, 0, 0);
        if (k >= 4 && xr + 16 < W) d16 = __builtin_amdgcn_raw_buffer_load_b32(b.rs, xo + 16, 0, 0);
        if (k >= 5 && xr + 32 < W) d32 = __builtin_amdgcn_raw_buffer_load_b32(b.rs, xo + 32, 0, 0);
        if (k >= 6 && xr + 64 < W) d64 = __builtin_amdgcn_raw_buffer_load_b32(b.rs, xo + 64, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < TH / 8; ++j) *reinterpret_cast<v4u *>(buf + (r + 8 * j) * S + 16 * c) = v[j];
    if (hrow) *reinterpret_cast<v4u *>(buf + (TH + r) * S + 16 * c) = hv;
    if (xl) {
        // halo column slots TW + {0..5} <- offsets {0, 4, 8, 16, 32, 64}
        v2u w;
        w.x = __builtin_amdgcn_perm(x0.y, x0.x, 0x0c0c0400u) | __builtin_amdgcn_perm(d16, x0.z, 0x04000c0cu);
        w.y = __builtin_amdgcn_perm(d64, d32, 0x0c0c0400u);
        const int rr = lane < TH / 2 ? 2 * lane : TH + (lane - TH / 2);
        *reinterpret_cast<v2u *>(buf + rr * S + TW) = w;
    }
}

// [x.b0, y.b0, z.b0, w.b0] of a 16-B row chunk: the four stride-4 lattice bytes
__device__ __forceinline__ u32 gather_b0(v4u a)
{
    return __builtin_amdgcn_perm(a.y, a.x, 0x0c0c0400u) | __builtin_amdgcn_perm(a.w, a.z, 0x04000c0cu);
}

// level sub == 2 of the tile body: four 4x4 cells (16 px x rows y0, y0+2; corners also from y0+4) per lane
template <int INTERP>
__device__ __forceinline__ void dec_level2_fast(u8 *buf)
{
#pragma unroll
    for (int it = 0; it < (TH / 4) * CH / NL; ++it) {
        const int i = threadIdx.x + it * NL;
        u8 *r0 = buf + 4 * (i >> LCH) * S + 16 * (i & (CH - 1));
        v4u A = *reinterpret_cast<const v4u *>(r0);
        v4u B = *reinterpret_cast<const v4u *>(r0 + 2 * S);
        v4u C = *reinterpret_cast<const v4u *>(r0 + 4 * S);
        u32 a16 = r0[16], c16 = r0[4 * S + 16];
        u32 ct = gather_b0(A), cb = gather_b0(C);
        u32 P = ct;
        if (INTERP == kInterpCrossed)
            P = pred4_crossed(ct, cb, __builtin_amdgcn_alignbyte(a16, ct, 1), __builtin_amdgcn_alignbyte(c16, cb, 1));
        u32 a0 = A.x, a1 = A.y, a2 = A.z, a3 = A.w, b0 = B.x, b1 = B.y, b2 = B.z, b3 = B.w;
        HGI_ADDB(a0, 2, P, 0); HGI_ADDB(a1, 2, P, 1); HGI_ADDB(a2, 2, P, 2); HGI_ADDB(a3, 2, P, 3);
        HGI_ADDB(b0, 0, P, 0); HGI_ADDB(b1, 0, P, 1); HGI_ADDB(b2, 0, P, 2); HGI_ADDB(b3, 0, P, 3);
        HGI_ADDB(b0, 2, P, 0); HGI_ADDB(b1, 2, P, 1); HGI_ADDB(b2, 2, P, 2); HGI_ADDB(b3, 2, P, 3);
        v4u An = {a0, a1, a2, a3}, Bn = {b0, b1, b2, b3};
        *reinterpret_cast<v4u *>(r0) = An;
        *reinterpret_cast<v4u *>(r0 + 2 * S) = Bn;
    }
}

// finest level: 16 px x 2 rows per lane, LDS -> packed VALU -> 16-B buffer stores
template <int INTERP>
__device__ __forceinline__ void dec_fine_fast(const u8 *buf, const Buf &b)
{
    const int lane = threadIdx.x;
    const u8 *r0 = buf + 2 * (lane >> LCH) * S + 16 * (lane & (CH - 1));
    u32 voff = b.base + 2 * (lane >> LCH) * b.W + 16 * (lane & (CH - 1));
#pragma unroll 2
    for (int it = 0; it < (TH / 2) * CH / NL; ++it, r0 += 2 * (NL / CH) * S, voff += 2 * (NL / CH) * b.W) {
        v4u E = *reinterpret_cast<const v4u *>(r0);
        v4u O = *reinterpret_cast<const v4u *>(r0 + S);
        v4u F = *reinterpret_cast<const v4u *>(r0 + 2 * S);
        u32 e16 = r0[16], f16 = r0[2 * S + 16];
        uint2 c, fl;
        c.x = __builtin_amdgcn_perm(E.y, E.x, 0x06040200u);
        c.y = __builtin_amdgcn_perm(E.w, E.z, 0x06040200u);
        fl.x = __builtin_amdgcn_perm(F.y, F.x, 0x06040200u);
        fl.y = __builtin_amdgcn_perm(F.w, F.z, 0x06040200u);
        u32 P0, P1;
        pred8<INTERP>(c, e16, fl, f16, P0, P1);
        u32 e0 = E.x, e1 = E.y, e2 = E.z, e3 = E.w, o0 = O.x, o1 = O.y, o2 = O.z, o3 = O.w;
        // row y: odd columns;  cell j of the lane = byte j of P0 (j < 4) or byte j-4 of P1
        HGI_ADDB(e0, 1, P0, 0); HGI_ADDB(e0, 3, P0, 1); HGI_ADDB(e1, 1, P0, 2); HGI_ADDB(e1, 3, P0, 3);
        HGI_ADDB(e2, 1, P1, 0); HGI_ADDB(e2, 3, P1, 1); HGI_ADDB(e3, 1, P1, 2); HGI_ADDB(e3, 3, P1, 3);
        // row y+1: every column
        HGI_ADDB(o0, 0, P0, 0); HGI_ADDB(o0, 1, P0, 0); HGI_ADDB(o0, 2, P0, 1); HGI_ADDB(o0, 3, P0, 1);
        HGI_ADDB(o1, 0, P0, 2); HGI_ADDB(o1, 1, P0, 2); HGI_ADDB(o1, 2, P0, 3); HGI_ADDB(o1, 3, P0, 3);
        HGI_ADDB(o2, 0, P1, 0); HGI_ADDB(o2, 1, P1, 0); HGI_ADDB(o2, 2, P1, 1); HGI_ADDB(o2, 3, P1, 1);
        HGI_ADDB(o3, 0, P1, 2); HGI_ADDB(o3, 1, P1, 2); HGI_ADDB(o3, 2, P1, 3); HGI_ADDB(o3, 3, P1, 3);
        v4u r0v = {e0, e1, e2, e3}, r1v = {o0, o1, o2, o3};
        __builtin_amdgcn_raw_buffer_store_b128(r0v, b.rd, voff, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(r1v, b.rd, voff, b.W, 0);
    }
}

// encode, level sub == 2 of the tile body: four cells per lane; corners from the half-resolution
// reconstruction lattice (rbuf), originals in / residuals out in buf, new reconstructions into rbuf
template <int INTERP, bool IDENT>
__device__ __forceinline__ void enc_level2_fast(u8 *buf, u8 *rbuf, const u8 *slut)
{
#pragma unroll
    for (int it = 0; it < (TH / 4) * CH / NL; ++it) {
        const int i = threadIdx.x + it * NL;
        const int g = i >> LCH, c = i & (CH - 1);
        u8 *r0 = buf + 4 * g * S + 16 * c;
        u8 *q0 = rbuf + 2 * g * S2 + 8 * c;
        v4u A = *reinterpret_cast<const v4u *>(r0);
        v4u B = *reinterpret_cast<const v4u *>(r0 + 2 * S);
        v2u cu = *reinterpret_cast<const v2u *>(q0);
        v2u cl = *reinterpret_cast<const v2u *>(q0 + 2 * S2);
        u32 c8 = q0[8], l8 = q0[2 * S2 + 8];
        u32 ct = __builtin_amdgcn_perm(cu.y, cu.x, 0x06040200u), cb = __builtin_amdgcn_perm(cl.y, cl.x, 0x06040200u);
        u32 P = ct;
        if (INTERP == kInterpCrossed)
            P = pred4_crossed(ct, cb, __builtin_amdgcn_alignbyte(c8, ct, 1), __builtin_amdgcn_alignbyte(l8, cb, 1));
        u32 a0 = A.x, a1 = A.y, a2 = A.z, a3 = A.w, b0 = B.x, b1 = B.y, b2 = B.z, b3 = B.w;
        v2u n0, n1;
        if (IDENT) {
            // identity table: residual = a - p, reconstruction = the original pixel
            const u32 pA = __builtin_amdgcn_perm(P, P, 0x0c010c00u), pB = __builtin_amdgcn_perm(P, P, 0x0c030c02u);
            n0.x = __builtin_amdgcn_perm(a0, ct, 0x0c010c00u) | __builtin_amdgcn_perm(a1, a0, 0x060c020cu);
            n0.y = __builtin_amdgcn_perm(a0, ct, 0x0c030c02u) | __builtin_amdgcn_perm(a3, a2, 0x060c020cu);
            n1.x = __builtin_amdgcn_perm(b1, b0, 0x06040200u);
            n1.y = __builtin_amdgcn_perm(b3, b2, 0x06040200u);
            const u32 m2 = 0x00FF0000u, m02 = 0x00FF00FFu;
            a0 = (a0 & ~m2) | (sub4(a0, pA << 16) & m2);
            a1 = (a1 & ~m2) | (sub4(a1, pA & m2) & m2);
            a2 = (a2 & ~m2) | (sub4(a2, pB << 16) & m2);
            a3 = (a3 & ~m2) | (sub4(a3, pB & m2) & m2);
            const u32 p0 = (pA & 0xFFu) * 0x00010001u, p1 = ((pA >> 16) & 0xFFu) * 0x00010001u;
            const u32 p2 = (pB & 0xFFu) * 0x00010001u, p3 = ((pB >> 16) & 0xFFu) * 0x00010001u;
            b0 = (b0 & ~m02) | (sub4(b0, p0) & m02);
            b1 = (b1 & ~m02) | (sub4(b1, p1) & m02);
            b2 = (b2 & ~m02) | (sub4(b2, p2) & m02);
            b3 = (b3 & ~m02) | (sub4(b3, p3) & m02);
        } else {
            const u32 NP = ~P;
            Q_PIX4(slut, P, NP, a0, 2, 0, a1, 2, 1, a2, 2, 2, a3, 2, 3);     // (x0+2, y0)
            Q_PIX4(slut, P, NP, b0, 0, 0, b1, 0, 1, b2, 0, 2, b3, 0, 3);     // (x0,   y0+2)
            Q_PIX4(slut, P, NP, b0, 2, 0, b1, 2, 1, b2, 2, 2, b3, 2, 3);     // (x0+2, y0+2)
            // lattice row 2g: corners stay, odd slots get the (x0+2, y0) reconstructions; row 2g+1: all new
            n0.x = __builtin_amdgcn_perm(ct, ct, 0x0c010c00u);
            n0.y = __builtin_amdgcn_perm(ct, ct, 0x0c030c02u);
            n1.x = 0;
            n1.y = 0;
            Q_REC(n0.x, 1, a0, 2, P, 0); Q_REC(n0.x, 3, a1, 2, P, 1); Q_REC(n0.y, 1, a2, 2, P, 2); Q_REC(n0.y, 3, a3, 2, P, 3);
            Q_REC(n1.x, 0, b0, 0, P, 0); Q_REC(n1.x, 1, b0, 2, P, 0); Q_REC(n1.x, 2, b1, 0, P, 1); Q_REC(n1.x, 3, b1, 2, P, 1);
            Q_REC(n1.y, 0, b2, 0, P, 2); Q_REC(n1.y, 1, b2, 2, P, 2); Q_REC(n1.y, 2, b3, 0, P, 3); Q_REC(n1.y, 3, b3, 2, P, 3);
        }
        v4u An = {a0, a1, a2, a3}, Bn = {b0, b1, b2, b3};
        *reinterpret_cast<v4u *>(r0) = An;
        *reinterpret_cast<v4u *>(r0 + 2 * S) = Bn;
        *reinterpret_cast<v2u *>(q0) = n0;
        *reinterpret_cast<v2u *>(q0 + S2) = n1;
    }
}

template <int INTERP, bool IDENT>
__device__ __forceinline__ void enc_fine_fast(const u8 *buf, const u8 *rbuf, const u8 *slut, const Buf &b)
{
    const int lane = threadIdx.x;
    const u8 *r0 = buf + 2 * (lane >> LCH) * S + 16 * (lane & (CH - 1));
    const u8 *c0 = rbuf + (lane >> LCH) * S2 + 8 * (lane & (CH - 1));
    u32 voff = b.base + 2 * (lane >> LCH) * b.W + 16 * (lane & (CH - 1));
    for (int it = 0; it < (TH / 2) * CH / NL; ++it, r0 += 2 * (NL / CH) * S, c0 += (NL / CH) * S2, voff += 2 * (NL / CH) * b.W) {
        v4u E = *reinterpret_cast<const v4u *>(r0);
        v4u O = *reinterpret_cast<const v4u *>(r0 + S);
        uint2 c = *reinterpret_cast<const uint2 *>(c0);
        uint2 fl = *reinterpret_cast<const uint2 *>(c0 + S2);
        u32 c8 = c0[8], f8 = c0[S2 + 8];
        u32 P0, P1;
        pred8<INTERP>(c, c8, fl, f8, P0, P1);
        u32 e0 = E.x, e1 = E.y, e2 = E.z, e3 = E.w, g0 = O.x, g1 = O.y, g2 = O.z, g3 = O.w;
        if (IDENT) {
            const u32 odd = 0xFF00FF00u;
            const u32 pp0 = __builtin_amdgcn_perm(P0, P0, 0x01010000u), pp1 = __builtin_amdgcn_perm(P0, P0, 0x03030202u);
            const u32 pp2 = __builtin_amdgcn_perm(P1, P1, 0x01010000u), pp3 = __builtin_amdgcn_perm(P1, P1, 0x03030202u);
            e0 = sub4(e0, pp0 & odd); e1 = sub4(e1, pp1 & odd); e2 = sub4(e2, pp2 & odd); e3 = sub4(e3, pp3 & odd);
            g0 = sub4(g0, pp0); g1 = sub4(g1, pp1); g2 = sub4(g2, pp2); g3 = sub4(g3, pp3);
        } else {
            const u32 N0 = ~P0, N1 = ~P1;
            // row y: only the odd columns are new (cell j of the lane = byte j of P0, or byte j-4 of P1)
            Q_PIX4(slut, P0, N0, e0, 1, 0, e0, 3, 1, e1, 1, 2, e1, 3, 3);
            Q_PIX4(slut, P1, N1, e2, 1, 0, e2, 3, 1, e3, 1, 2, e3, 3, 3);
            // row y+1: every column
            Q_PIX4(slut, P0, N0, g0, 0, 0, g0, 1, 0, g0, 2, 1, g0, 3, 1);
            Q_PIX4(slut, P0, N0, g1, 0, 2, g1, 1, 2, g1, 2, 3, g1, 3, 3);
            Q_PIX4(slut, P1, N1, g2, 0, 0, g2, 1, 0, g2, 2, 1, g2, 3, 1);
            Q_PIX4(slut, P1, N1, g3, 0, 2, g3, 1, 2, g3, 2, 3, g3, 3, 3);
        }
        v4u o0 = {e0, e1, e2, e3}, o1 = {g0, g1, g2, g3};
        __builtin_amdgcn_raw_buffer_store_b128(o0, b.rd, voff, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(o1, b.rd, voff, b.W, 0);
    }
}

__device__ __forceinline__ Buf make_buf(const u8 *fr, u8 *out, u32 W, u32 H, Tile tl)
{
    Buf b;
    const u32 bytes = W * H;   // the host only selects the fast path when this (plus the halo) fits 32 bits
    b.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u8 *>(fr), 0, bytes, 0x00020000);
    b.rd = __builtin_amdgcn_make_buffer_rsrc(out, 0, bytes, 0x00020000);
    b.W = W;
    b.base = tl.Y0 * W + tl.X0;
    return b;
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// flags bit 0: rows are 16-B aligned (width % 16 == 0, aligned pointers and frame stride)
//       bit 1: 32-bit buffer offsets are safe for this frame size (fast path allowed)
template <int INTERP, bool SEEDED>
__global__ __launch_bounds__(NL) void k_dec_fused(const u8 *__restrict__ src, u8 *__restrict__ dst, Frames f, u32 k,
                                                  Seeds sd, u32 tiles_x, u32 tiles_y, u32 ntiles, u32 flags)
{
    __shared__ __attribute__((aligned(16))) u8 buf[R * S];
    const int lane = threadIdx.x;
    const Tile tl = tile_of_block(ntiles, tiles_x, tiles_y);
    const u32 W = f.width, H = f.height;
    const u8 *fr = src + (size_t)tl.frame * f.frame_stride;
    u8 *out = dst + (size_t)tl.frame * f.frame_stride;
    const int nh = k >= 2 ? (int)k : 1;
    const bool aligned = (flags & 1u) != 0;
    const bool fast = flags == 3u && tl.X0 + TW <= W && tl.Y0 + TH <= H;
    const Buf b = make_buf(fr, out, W, H, tl);

    if (fast)
        stage_tile_fast(buf, b, tl, (int)k, nh);
    else
        stage_tile_generic(buf, fr, W, H, tl, nh, aligned);
    __syncthreads();   // one wave per workgroup: an LDS-ordering point, not a hardware barrier
    if (SEEDED) {
        // lattice points = 0 (mod 2^k) come from the already decoded coarser pyramid
        const int ext = k >= 2 ? 2 : 1;   // offset 2^k beyond the tile is only ever read for k >= 2
        const int nbx = (TW >> k) + ext, nby = (TH >> k) + ext;
        const u8 *sp = sd.rec + (size_t)tl.frame * sd.stride;
        for (int i = lane; i < nbx * nby; i += NL) {
            int by = i / nbx, bx = i - by * nbx;
            u32 sx = (tl.X0 >> k) + bx, sy = (tl.Y0 >> k) + by;
            u8 v = (sx < sd.sw && sy < sd.sh) ? sp[(size_t)sy * sd.sw + sx] : (u8)0;
            buf[lrow(by << k) * S + lcol(bx << k)] = v;
        }
        __syncthreads();
    }
    for (int s = 1 << (k - 1); s >= 2; s >>= 1) {
        if (fast) {
            if (s == 2)
                dec_level2_fast<INTERP>(buf);
            else
                dec_cells<INTERP, false>(buf, s, tl, W, H);
        } else {
            dec_cells<INTERP, true>(buf, s, tl, W, H);
        }
        dec_halo_cells<INTERP>(buf, s, tl, W, H);
        __syncthreads();
    }
    if (fast)
        dec_fine_fast<INTERP>(buf, b);
    else
        dec_fine_generic<INTERP>(buf, out, tl, W, H, aligned);
}

template <int INTERP, bool IDENT, bool SEEDED>
__global__ __launch_bounds__(NL) void k_enc_fused(const u8 *__restrict__ src, u8 *__restrict__ dst, Frames f, u32 k,
                                                  Lut256 lut, Seeds sd, u32 tiles_x, u32 tiles_y, u32 ntiles,
                                                  u32 flags)
{
    __shared__ __attribute__((aligned(16))) u8 buf[R * S];
    __shared__ __attribute__((aligned(16))) u8 rbuf[R2 * S2];
    __shared__ __attribute__((aligned(16))) u8 slut[256];
    const int lane = threadIdx.x;
    const Tile tl = tile_of_block(ntiles, tiles_x, tiles_y);
    const u32 W = f.width, H = f.height;
    const u8 *fr = src + (size_t)tl.frame * f.frame_stride;
    u8 *out = dst + (size_t)tl.frame * f.frame_stride;
    const int nh = k >= 2 ? (int)k : 1;
    const bool aligned = (flags & 1u) != 0;
    const bool fast = flags == 3u && tl.X0 + TW <= W && tl.Y0 + TH <= H;
    const Buf b = make_buf(fr, out, W, H, tl);

    if (!IDENT) reinterpret_cast<u32 *>(slut)[lane] = lut.w[lane];
    // lattice points outside the image must read as 0 (src/interpolator.rs:75-82) and are never written
    for (int i = lane; i < R2 * S2 / 16; i += NL) reinterpret_cast<uint4 *>(rbuf)[i] = make_uint4(0, 0, 0, 0);
    if (fast)
        stage_tile_fast(buf, b, tl, (int)k, nh);
    else
        stage_tile_generic(buf, fr, W, H, tl, nh, aligned);
    __syncthreads();
    {
        // lattice points = 0 (mod 2^k): reconstruction == original (src/encoder.rs:26-37), or the
        // coarser pyramid's reconstruction + residuals when this launch is the lower part of a
        // deeper pyramid.
        const int ext = k >= 2 ? 2 : 1;
        const int nbx = (TW >> k) + ext, nby = (TH >> k) + ext;
        const u8 *sr = SEEDED ? sd.rec + (size_t)tl.frame * sd.stride : nullptr;
        const u8 *sq = SEEDED ? sd.q + (size_t)tl.frame * sd.stride : nullptr;
        for (int i = lane; i < nbx * nby; i += NL) {
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
        if (fast) {
            if (s == 2)
                enc_level2_fast<INTERP, IDENT>(buf, rbuf, slut);
            else
                enc_cells<INTERP, IDENT, false>(buf, rbuf, slut, s, tl, W, H);
        } else {
            enc_cells<INTERP, IDENT, true>(buf, rbuf, slut, s, tl, W, H);
        }
        enc_halo_cells<INTERP, IDENT>(buf, rbuf, slut, s, tl, W, H);
        __syncthreads();
    }
    if (fast)
        enc_fine_fast<INTERP, IDENT>(buf, rbuf, slut, b);
    else
        enc_fine_generic<INTERP, IDENT>(buf, rbuf, slut, out, tl, W, H, aligned);
}

struct FusedGeom {
    u32 tiles_x, tiles_y, ntiles, flags;
    bool ok;
};

inline bool ptr16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

FusedGeom fused_geom(const void *a, const void *b, const Frames &f)
{
    FusedGeom g;
    g.tiles_x = (f.width + TW - 1) / TW;
    g.tiles_y = (f.height + TH - 1) / TH;
    u64 nt = (u64)g.tiles_x * g.tiles_y * f.batch;
    g.ok = nt > 0 && nt < (1ull << 31);
    g.ntiles = (u32)nt;
    const bool aligned = f.width % 16 == 0 && f.frame_stride % 16 == 0 && ptr16(a) && ptr16(b);
    // every 32-bit buffer offset the fast path forms: (Y0 + TH + 64) * W + X0 + TW + 64 + 16
    const bool fits32 = ((u64)f.height + 2 * TH + 64) * f.width + 1024 < (1ull << 32);
    g.flags = (aligned ? 1u : 0u) | (aligned && fits32 ? 2u : 0u);
    return g;
}

}  // namespace

hipError_t launch_decode_fused(const uint8_t *grid, uint8_t *img, const Frames &f, uint32_t k, int interp,
                               const Seeds *seeds, hipStream_t s)
{
    FusedGeom g = fused_geom(grid, img, f);
    if (!g.ok || k < 1 || k > (u32)kFusedMaxLevels) return hipErrorInvalidValue;
    Seeds sd = seeds ? *seeds : Seeds{nullptr, nullptr, 0, 0, 0};
    dim3 gr(g.ntiles), b(NL);
#define HGI_DEC(I, SE) \
    hipLaunchKernelGGL((k_dec_fused<I, SE>), gr, b, 0, s, grid, img, f, k, sd, g.tiles_x, g.tiles_y, g.ntiles, g.flags)
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
    dim3 gr(g.ntiles), b(NL);
#define HGI_ENC(I, ID, SE) \
    hipLaunchKernelGGL((k_enc_fused<I, ID, SE>), gr, b, 0, s, img, grid, f, k, lut, sd, g.tiles_x, g.tiles_y, g.ntiles, g.flags)
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

}  // namespace hgi
