// Device-side helpers shared by the level-wise and fused kernels (gfx950).
#pragma once
#include "hgi_kernels.h"

namespace hgi {
namespace dev {

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

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

}  // namespace dev
}  // namespace hgi
