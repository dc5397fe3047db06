// gfx950 (MI355X / CDNA4) fused HGI kernels: all levels of a tile in one launch, LDS-resident.
//
// This is the implementation.  It is compiled six times -- per direction (hgi_fused_dec.hip / hgi_fused_enc.hip:
// the directions want different LDS pitches and register budgets) and per tile height (128 x 64 tiles for
// throughput, 128 x 32 tiles for calls too small to fill the GPU, 128 x 16 tiles for the encode of a lone small frame;
// hgi_fused_*32.hip, hgi_fused_*16.hip) -- and hgi_capi.hip picks the build per launch.
//
// Reference algorithm (paths relative to pl0q1n/RustyHGI):
//   src/encoder.rs:39-71, src/decoder.rs:18-46, src/interpolator.rs:15-28 / :41-90, src/utils.rs:12-41
//
// Work decomposition
//   * ONE WAVE (64 lanes) owns one tile.  A tile row is one 128-B line = 8 lanes x 16 B, so a wave-wide
//     load/store instruction moves 8 full lines.  No workgroup barrier exists anywhere: all cross-lane traffic
//     goes through the wave's own LDS slice, ordered by the wave's in-order LDS queue; the other waves of the
//     CU (other tiles, other phases) hide the latency.
//   * The tile plus a sparse one-sided halo (offsets {0,4,8,...,2^k} to the right and below) is
//     loaded once; halo pixels are recomputed bit-identically instead of exchanged.  Every image byte is
//     fetched from HBM once and every output byte written once.
//   * Only the EVEN rows go to LDS.  Every level but the finest touches even coordinates only; the finest
//     level reads each odd row once, in the lane that loaded it, so odd rows stay in registers.
//   * Levels sub >= 4 (6 % of the pixels): one lane per step-cell, byte LDS accesses.
//     Level sub == 2 (19 %): four cells per lane, packed v_lerp_u8 predictor, 16-B LDS accesses.
//     Level sub == 1 (75 %): 16 px x 2 rows per lane through packed-u8 / SDWA arithmetic straight to 16-B
//     buffer stores.
//   * Interior tiles (tile body inside the image) take a check-free path built on buffer loads whose hardware
//     range check returns 0 beyond the frame -- exactly the reference's out-of-image rule
//     (src/interpolator.rs:75-82).  Ragged tiles and unaligned widths take a fully checked path in the SAME
//     launch (their blocks come first).
//
// All arithmetic is u8/integer; there is no MFMA-shaped work on this path.
#include "hgi_dev.h"
#include "hgi_fastdiv.h"
#include "hgi_knobs.h"

namespace hgi {
namespace {

using namespace dev;

#ifndef HGI_TILE_H
#define HGI_TILE_H 64
#endif
#define HGI_LOG2(v) ((v) == 32 ? 5 : (v) == 16 ? 4 : (v) == 8 ? 3 : (v) == 4 ? 2 : 1)
#define HGI_CAT2(a, b) a##_##b
#define HGI_CAT(a, b) HGI_CAT2(a, b)
#define HGI_TILED(name) HGI_CAT(name, HGI_TILE_H)   // launch_decode_fused -> launch_decode_fused_64
#define HGI_LANE ((int)threadIdx.x)   // the workgroup is one wave
constexpr int TW = kTileW;
constexpr int TH = HGI_TILE_H;
constexpr int MAXK = TH == 64 ? kFusedMaxLevels : TH == 32 ? kFusedMaxLevelsSmall : kFusedMaxLevelsTiny;   // deepest pyramid one tile holds
constexpr int NL = kThreads;       // lanes
constexpr int CH = TW / 16;        // 16-B chunks per tile row
constexpr int LCH = 3;             // log2(CH)
constexpr int HR = 6;   // halo rows / columns a tile can need (offsets 0,4,8,16,32,64); >= MAXK
// LDS layout (bank-conflict model and measurements: DESIGN.md "LDS layout").  LDS is dynamic: a tile
// that needs nh halo rows allocates TH / 2 + nh rows, which is what sets the waves per CU.
//   ODD image rows never enter LDS: every level but the finest touches even coordinates only, and the
//   finest level reads an odd row exactly once -- in the lane that loaded it, so it stays in that
//   lane's registers from the staging load to the store (the staging lane map IS the fine-level map).
//   even-row plane `buf`: [halo columns, transposed][LDS row r = image row 2r, r < TH / 2; then the
//     halo rows TH + {0,4,8,..} at r = TH / 2 + hmap], pitch S, full x resolution.
//     `buf` points at row 0; the halo COLUMNS live in front of it at buf[HCOL + idx * HP + r], so
//     that the lanes of a halo-cell pass (one lane per row) touch consecutive bytes instead of one
//     bank, and so that every offset stays a compile-time constant whatever nh is.
//   half-resolution plane `rbuf` (encode: reconstruction of the even/even lattice): same rows, x / 2.
#ifndef HGI_S_PAD
#define HGI_S_PAD 16    // row pad of the full-resolution plane (bank skew vs. LDS per wave)
#endif
#ifndef HGI_S2_PAD
#define HGI_S2_PAD 8
#endif
constexpr int S = TW + HGI_S_PAD;
constexpr int HP = 40;             // >= TH / 2 + HR
constexpr int HCOL = -(HR * HP);   // the transposed halo columns sit in front of row 0
constexpr int S2 = TW / 2 + HGI_S2_PAD;
constexpr int HP2 = 40;            // >= TH / 2 + HR
constexpr int RCOL = -(HR * HP2);

__host__ __device__ constexpr int buf_bytes(int nh) { return HR * HP + (TH / 2 + nh) * S; }
__host__ __device__ constexpr int rbuf_bytes(int nh) { return HR * HP2 + (TH / 2 + nh) * S2; }

static_assert(NL == 64 && CH == (1 << LCH) && (TH & (TH - 1)) == 0 && (TH == 64 || TH == 32 || TH == 16) && (1 << MAXK) <= TH, "tile geometry");
// 16-B accesses on the full-resolution rows, 8-B accesses on the half-resolution rows
static_assert(S % 16 == 0 && (HR * HP) % 16 == 0 && S2 % 8 == 0 && (HR * HP2) % 16 == 0 && HP >= TH / 2 + HR &&
                  HP2 >= TH / 2 + HR && buf_bytes(1) % 16 == 0 && rbuf_bytes(1) % 8 == 0,
              "LDS pitches keep vector alignment");
// k = 4 (the flagship configuration): LDS leaves room for 24 decode / 18 encode waves per CU
#ifdef HGI_FUSED_DECODE
static_assert(TH != 64 || 24 * buf_bytes(4) <= 160 * 1024, "LDS budget: 24 decode waves per CU at k = 4");
#endif
#ifdef HGI_FUSED_ENCODE
static_assert(TH != 64 || 18 * (buf_bytes(4) + rbuf_bytes(4) + 256) <= 160 * 1024, "LDS budget: 18 encode waves per CU at k = 4");
#endif

#ifndef HGI_LOAD_AUX
#define HGI_LOAD_AUX 0    // cache policy of the streaming tile-body loads (2 = nt)
#endif
#ifndef HGI_ODD_LOAD_AUX
#define HGI_ODD_LOAD_AUX 2   // odd-row loads: nt.  No tile ever takes a halo line from an odd row, so these are pure streaming
                             // reads (decode -2 %, encode -1 %); nt on the even rows costs 4 % (halo lines rely on L2)
#endif
#ifndef HGI_STORE_AUX
#define HGI_STORE_AUX 2   // cache policy of the output stores: nt (streamed once; measured 2-3 % faster than default)
#endif
#ifndef HGI_ODD_LATE
#define HGI_ODD_LATE 1      // interior tiles request their odd rows after the even rows are committed (stage_issue_odd); 0: with them
#endif
#ifndef HGI_DEC_CONE_FIRST
#define HGI_DEC_CONE_FIRST 0     // the decoder's cone load in front of the tile's staging loads (1) or behind them (0)
#endif
#ifndef HGI_DEC_SHALLOW_WAVES
#define HGI_DEC_SHALLOW_WAVES 20  // resident tiles per CU asked for on decodes 1-8 rounds deep (launch_decode_fused)
#endif
#ifndef HGI_DEC_DEEP_WAVES
#define HGI_DEC_DEEP_WAVES 16     // ... and on deeper ones (launches that rebuild levels in the kernel or start from seed planes)
#endif
#ifndef HGI_DEC_STREAM_WAVES
#define HGI_DEC_STREAM_WAVES 10   // resident tiles per CU of a plain decode (the tile holds the pyramid) of 8 192 tiles and more, rows up
#endif                            // to 4 096 pixels and up to four levels ...
#ifndef HGI_DEC_STREAM_WAVES_WIDE
#define HGI_DEC_STREAM_WAVES_WIDE 12   // ... and on wider rows or five levels
#endif
#ifndef HGI_DEC_STREAM_WAVES_L1
#define HGI_DEC_STREAM_WAVES_L1 8      // ... and at ONE level (the finest pass alone: a tile's lifetime is shortest there)
#endif
#ifndef HGI_XCD_MODE
#define HGI_XCD_MODE -1     // the XCD dealing policy (block_role): 0 contiguous eighths, 1 whole bands round-robin, -1 by size (xcd_mode())
#endif
#ifndef HGI_XCD_EIGHTHS_FROM_GIB
#define HGI_XCD_EIGHTHS_FROM_GIB kEncodeEighthsFromGiB   // encodes whose interior tiles span at least this many GiB deal contiguous eighths
#endif
#ifndef HGI_DEC_REVERSE_DEFAULT
#define HGI_DEC_REVERSE_DEFAULT 0
#endif
#ifndef HGI_TILE_ORDER
#define HGI_TILE_ORDER 0      // order of the interior tiles inside a frame: 0 row-major (experiment), 3 column-major bands (shipped; set by the direction's unit)
#endif
#ifndef HGI_TILE_BAND
#define HGI_TILE_BAND 8
#endif
#ifndef HGI_HALO_FIRST
#define HGI_HALO_FIRST 0
#endif
#ifndef HGI_HALO_ALL_ROWS
#define HGI_HALO_ALL_ROWS 0   // 1: fetch the halo-column offsets >= 16 on every even row (the round-1 behaviour)
#endif
// Analysis builds (tools/isa_phases.py): -DHGI_ANALYZE_K=4 compiles the interior-tile path alone with a constant
// depth, so that it is straight-line code, and leaves phase markers in the ISA.  Never linked into the library.
#ifdef HGI_ANALYZE_K
#define HGI_MARK(name) asm volatile("; HGI_MARK " name)
#else
#define HGI_MARK(name)
#endif
typedef u32 v4u __attribute__((ext_vector_type(4)));
typedef u32 v2u __attribute__((ext_vector_type(2)));
typedef u32 v3u __attribute__((ext_vector_type(3)));

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
// The encoder keeps the quantizer table at LDS offset 0 (k_enc_tiles checks it), so the clean residual byte d IS the
// LDS address of table[d]: an address-space-3 pointer made from the integer gives a bare `ds_read_u8 q, d` -- no
// per-pixel address add -- and stays visible to the compiler's s_waitcnt bookkeeping.
typedef __attribute__((address_space(3))) const u8 lds_cu8;
__device__ __forceinline__ u32 lut_at(u32 d) { return *(lds_cu8 *)(size_t)d; }
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
        u32 q0_ = lut_at(d0_), q1_ = lut_at(d1_), q2_ = lut_at(d2_), q3_ = lut_at(d3_);                \
        Q_LT(b0_, O0, K0, P, J0); Q_LT(b1_, O1, K1, P, J1); Q_LT(b2_, O2, K2, P, J2); Q_LT(b3_, O3, K3, P, J3); \
        Q_GT(c0_, q0_, NP, J0); Q_GT(c1_, q1_, NP, J1); Q_GT(c2_, q2_, NP, J2); Q_GT(c3_, q3_, NP, J3); \
        Q_SEL(O0, K0, b0_, c0_, q0_, d0_); Q_SEL(O1, K1, b1_, c1_, q1_, d1_);                          \
        Q_SEL(O2, K2, b2_, c2_, q2_, d2_); Q_SEL(O3, K3, b3_, c3_, q3_, d3_);                          \
    } while (0)
// one pixel held in its own registers (a = original, p = prediction, both clean bytes) -> residual
#define Q_PIX1(out, a, p, slut)                                                                        \
    do {                                                                                               \
        u32 d_, np_ = ~(p);                                                                            \
        lanemask b_, c_;                                                                               \
        Q_SUB(d_, a, 0, p, 0);                                                                         \
        u32 q_ = lut_at(d_);                                                                           \
        Q_LT(b_, a, 0, p, 0);                                                                          \
        Q_GT(c_, q_, np_, 0);                                                                          \
        asm("s_xor_b64 vcc, %1, %2\n\tv_cndmask_b32_e32 %0, %3, %4, vcc"                               \
            : "=v"(out) : "s"(b_), "s"(c_), "v"(q_), "v"(d_) : "vcc", "scc");                          \
    } while (0)
// dst.byte[KD] = (q.byte[KQ] + p.byte[J]) mod 256: the reconstruction of a freshly coded pixel
#define Q_REC(dst, KD, q, KQ, p, J)                                                                    \
    asm("v_add_u16_sdwa %0, %1, %2 dst_sel:BYTE_" #KD " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #KQ   \
        " src1_sel:BYTE_" #J                                                                           \
        : "+v"(dst) : "v"(q), "v"(p))

// src/encoder.rs:53-60 for one pixel in its own registers (coarse levels, halo cells)
template <bool IDENT>
__device__ __forceinline__ u32 quant1s(u32 a, u32 p, const u8 *slut)
{
    if (IDENT) return (a - p) & 255u;
    u32 out;
    Q_PIX1(out, a, p, slut);
    return out;
}

// N pixels that share one prediction (the new pixels of one cell; N = 6: of two cells), split-phase: all residuals,
// then all table look-ups back to back, then the fallback tests -- ONE dependent LDS round trip instead of N.
template <bool IDENT, int N>
__device__ __forceinline__ void quant_batch(const u32 (&a)[N], const u32 (&p)[N], u32 (&out)[N])
{
    if (IDENT) {
#pragma unroll
        for (int i = 0; i < N; ++i) out[i] = (a[i] - p[i]) & 255u;
        return;
    }
    u32 d[N], q[N];
#pragma unroll
    for (int i = 0; i < N; ++i) Q_SUB(d[i], a[i], 0, p[i], 0);
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = lut_at(d[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const u32 np = ~p[i];
        lanemask b, c;
        Q_LT(b, a[i], 0, p[i], 0);
        Q_GT(c, q[i], np, 0);
        asm("s_xor_b64 vcc, %1, %2\n\tv_cndmask_b32_e32 %0, %3, %4, vcc"
            : "=v"(out[i]) : "s"(b), "s"(c), "v"(q[i]), "v"(d[i]) : "vcc", "scc");
    }
}

// ---------------------------------------------------------------------------------------------
// tile bookkeeping
// ---------------------------------------------------------------------------------------------
// Halo coordinates.  Beyond the tile only offsets {0, 4, 8, ..., 2^k} are ever touched (level `sub`
// reads corners at offset 2*sub and writes at offset sub, sub >= 4), so halo
// rows/columns are stored compactly at index hmap(offset).
__device__ __forceinline__ int hmap(int off) { return off ? 30 - __clz(off) : 0; }   // 4->1, 8->2 ...
__device__ __forceinline__ int hoff(int idx) { return idx ? 2 << idx : 0; }           // 1->4, 2->8 ...
// LDS row of the EVEN image row y of the tile (halo included); both planes index rows alike
__device__ __forceinline__ int lrow(int y) { return y < TH ? y >> 1 : TH / 2 + hmap(y - TH); }
__device__ __forceinline__ int lrow2(int y) { return lrow(y); }
// byte offset of pixel (x, y), y even, of the tile (halo included) in the even-row / half-resolution plane
__device__ __forceinline__ int laddr(int x, int y) { return x < TW ? lrow(y) * S + x : HCOL + hmap(x - TW) * HP + lrow(y); }
__device__ __forceinline__ int laddr2(int x, int y)
{
    return x < TW ? lrow2(y) * S2 + (x >> 1) : RCOL + hmap(x - TW) * HP2 + lrow2(y);
}
// Value of lane + 1: the first dword of the next 16-B chunk of the same row.  Through the LDS crossbar
// (ds_bpermute_b32; no memory access, no bank conflicts) rather than DPP: a VALU write -> DPP read needs
// two wait states, and the compiler cannot count them through the inline-asm SDWA statements.  Same
// speed; tools/check_isa.py rejects any DPP in these kernels.
__device__ __forceinline__ u32 from_next_lane(u32 v)
{
    return (u32)__builtin_amdgcn_ds_bpermute(((HGI_LANE + 1) << 2), (int)v);
}

struct Tile {
    u32 frame, X0, Y0;
};

// XCD-aware block -> tile map.  Workgroups are dealt round-robin over the 8 XCDs (b % 8 labels the XCD a
// block runs on), so XCD x gets the x-th contiguous eighth of the row-major tile list: x-neighbours
// (which share halo lines) and consecutive tile rows land in the same XCD's L2.  Speed only, never
// correctness.
__device__ __forceinline__ u32 range_first(u32 ntiles, u32 x) { return x * (ntiles >> 3) + (x < (ntiles & 7u) ? x : (ntiles & 7u)); }

// Compile-time ordering point for the wave's LDS traffic.  A wave's LDS instructions execute in
// order, so a ds_read issued after a ds_write sees it without any wait; the compiler only has to be
// kept from moving accesses across (it cannot see that lanes exchange data).  No s_barrier, no
// vmcnt drain: the workgroup is one wave.
#define LDS_ORDER() asm volatile("" ::: "memory")

// Timeline builds (-DHGI_TIMELINE, never shipped): every interior block records when it started, when its staging
// loads had landed, when its last store was acknowledged (100 MHz s_memrealtime) and where it ran (XCC, HW_ID).
#ifdef HGI_TIMELINE
__device__ __forceinline__ u64 tl_now(bool drain)
{
    u64 t;
    if (drain)
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    else
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ void tl_write(u64 *tl, u64 te, u64 t0, u64 t1)
{
    const u64 t2 = tl_now(true);
    const u32 hw = __builtin_amdgcn_s_getreg(4 | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));
    if (tl && threadIdx.x == 0) {
        u64 *r = tl + 8 * (size_t)blockIdx.x;
        r[0] = t0; r[1] = t1; r[2] = t2; r[3] = ((u64)xcc << 32) | hw; r[4] = te;
    }
}
#define HGI_TL_ENTRY() const u64 tl_te = tl_now(false)      /* first statement of the kernel: the wave is on its CU */
#define HGI_TL_START() const u64 tl_t0 = tl_now(false)      /* prologue done (arguments, table, tile index): loads go out */
#define HGI_TL_STAGED() const u64 tl_t1 = tl_now(true)
#define HGI_TL_END() tl_write(g.timeline, tl_te, tl_t0, tl_t1)
#else
#define HGI_TL_ENTRY()
#define HGI_TL_START()
#define HGI_TL_STAGED()
#define HGI_TL_END()
#endif

// The two 16-B row stores of a lane's fine-level task, then two wait states during which the eight
// data registers stay allocated.  A VALU write to a data VGPR of a > 64-bit store in the issue slot
// right after it can reach memory instead of the stored value.  LLVM pads for that only when soffset
// is not an SGPR; on gfx950 it was observed with soffset in an SGPR as well (first data dword, last
// quad of every 16 lanes, only with the store path backed up -- DESIGN.md 4.5), so the spacing is
// made explicit here and checked in the ISA (tools/check_isa.py rule 4).
__device__ __forceinline__ void store_row_pair(v4u r0, v4u r1, __amdgpu_buffer_rsrc_t rd, u32 voff, u32 pitch)
{
    __builtin_amdgcn_raw_buffer_store_b128(r0, rd, voff, 0, HGI_STORE_AUX);
    __builtin_amdgcn_raw_buffer_store_b128(r1, rd, voff, pitch, HGI_STORE_AUX);
    asm volatile("s_nop 1" ::"v"(r0), "v"(r1));
}

// Per-dword byte masks of a 16-B chunk of which only the first `nvalid` bytes lie inside the image
// (nvalid <= 0: none, >= 16: all).
__device__ __forceinline__ v4u chunk_mask(int nvalid)
{
    v4u m;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int nb = nvalid - 4 * d;
        m[d] = nb >= 4 ? ~0u : (nb <= 0 ? 0u : (1u << (8 * nb)) - 1u);
    }
    return m;
}
// ... and of the 8-B half-resolution chunk that goes with it (byte j <-> image column 2 j of the chunk)
__device__ __forceinline__ v2u chunk_mask2(int nvalid)
{
    const v4u m = chunk_mask((nvalid + 1) >> 1);
    v2u r = {m.x, m.y};
    return r;
}

// The row pair of a lane of an EDGE tile: whole chunks as 16-B stores, of the chunk that straddles the right image
// edge only the dwords inside, rows below the image not at all.  The data registers are held like in store_row_pair.
__device__ __forceinline__ void store_rows_edge(v4u r0, v4u r1, __amdgpu_buffer_rsrc_t rd, u32 voff, u32 pitch, int nvalid,
                                                bool row0, bool row1)
{
    // the straddling chunk: its whole dwords, then (widths that are not multiples of 4) the last 1..3 bytes
    const u32 v1 = voff + pitch;
    if (nvalid >= 16 && row1) {            // whole chunk, both rows inside (row1 implies row0)
        __builtin_amdgcn_raw_buffer_store_b128(r0, rd, voff, 0, HGI_STORE_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(r1, rd, voff, pitch, HGI_STORE_AUX);
    } else if (nvalid >= 16) {
        if (row0) __builtin_amdgcn_raw_buffer_store_b128(r0, rd, voff, 0, HGI_STORE_AUX);
    } else if (nvalid >= 12) {
        if (row0) __builtin_amdgcn_raw_buffer_store_b96(v3u{r0.x, r0.y, r0.z}, rd, voff, 0, 0);
        if (row1) __builtin_amdgcn_raw_buffer_store_b96(v3u{r1.x, r1.y, r1.z}, rd, v1, 0, 0);
    } else if (nvalid >= 8) {
        if (row0) __builtin_amdgcn_raw_buffer_store_b64(v2u{r0.x, r0.y}, rd, voff, 0, 0);
        if (row1) __builtin_amdgcn_raw_buffer_store_b64(v2u{r1.x, r1.y}, rd, v1, 0, 0);
    } else if (nvalid >= 4) {
        if (row0) __builtin_amdgcn_raw_buffer_store_b32(r0.x, rd, voff, 0, 0);
        if (row1) __builtin_amdgcn_raw_buffer_store_b32(r1.x, rd, v1, 0, 0);
    }
    if (nvalid > 0 && nvalid < 16 && (nvalid & 3)) {
        const int d = nvalid >> 2;
        const u32 w0 = d == 0 ? r0.x : d == 1 ? r0.y : d == 2 ? r0.z : r0.w;
        const u32 w1 = d == 0 ? r1.x : d == 1 ? r1.y : d == 2 ? r1.z : r1.w;
        for (int j = 0; j < (nvalid & 3); ++j) {
            if (row0) __builtin_amdgcn_raw_buffer_store_b8((u8)(w0 >> (8 * j)), rd, voff + 4 * d + j, 0, 0);
            if (row1) __builtin_amdgcn_raw_buffer_store_b8((u8)(w1 >> (8 * j)), rd, v1 + 4 * d + j, 0, 0);
        }
    }
    asm volatile("s_nop 1" ::"v"(r0), "v"(r1));
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
        } else if (gx + 16 <= W) {
            // chunk inside the row at any byte alignment: still one 16-B access (gfx9+ under ROCm runs with
            // unaligned access enabled; the compiler emits global_load_dwordx4 for this)
            __builtin_memcpy(&v, p, 16);
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
    } else if (gx + 16 <= W) {
        __builtin_memcpy(p, &v, 16);
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
    const int lane = HGI_LANE;
    for (int i = lane; i < (TH / 2 + nh) * CH; i += NL) {      // even rows and halo rows only
        int rr = i >> LCH, c = i & (CH - 1);
        int y = rr < TH / 2 ? 2 * rr : TH + hoff(rr - TH / 2);
        uint4 v = load16(fr, W, H, tl.X0 + 16 * c, tl.Y0 + y, aligned);
        *reinterpret_cast<uint4 *>(buf + rr * S + 16 * c) = v;
    }
    // halo columns TW + {0,4,8,..} (and the halo x halo corner block): byte gathers.  Column offset
    // `off` is only ever touched on rows = 0 (mod max(off, 2)).
    for (int i = lane; i < 8 * (TH / 2 + nh); i += NL) {
        int hc = i & 7, rr = i >> 3;
        if (hc >= nh) continue;
        int off = hoff(hc);
        int y = rr < TH / 2 ? 2 * rr : TH + hoff(rr - TH / 2);
        if (rr < TH / 2 && (y & ((off ? off : 2) - 1))) continue;
        u32 gx = tl.X0 + TW + off, gy = tl.Y0 + y;
        buf[HCOL + hc * HP + rr] = (gx < W && gy < H) ? fr[(size_t)gy * W + gx] : (u8)0;
    }
}

// Halo cells of level `s`: the cell column x0 == TW (rows y0 = 0, step, .., TH) and the cell row
// y0 == TH (columns x0 = 0, step, .., TW - step).  They recompute, bit-identically, what the right /
// lower neighbour tiles compute for themselves.  Level-`sub` halo pixels off the tile edge are needed
// only for sub >= 4 (`deep`); at sub == 2 only those on the edge itself.  Column cells work on the
// transposed halo columns (one lane per row -> consecutive bytes), row cells on the natural halo
// rows; the slot indices hmap(sub), hmap(2*sub) are wave-uniform.  Used by BOTH paths (the halo may
// leave the image even when the tile body does not), so every pixel is checked against the image.
template <int INTERP>
__device__ __forceinline__ void dec_halo_cells(u8 *buf, int s, Tile tl, u32 W, u32 H)
{
    const int step = 2 * s, lstep = 31 - __clz(step);
    const int ncx = TW >> lstep, ncy = TH >> lstep;
    const bool deep = s >= 4;
    const int hs = hmap(s), h2 = hmap(step);
    u8 *hc = buf + HCOL;
    const int lane = HGI_LANE;
    if (tl.X0 + TW < W && lane <= ncy) {           // column cells; lane == ncy is the corner cell (TW, TH)
        const int y0 = lane << lstep;
        if (tl.Y0 + y0 < H) {
            const bool corner = y0 == TH;
            // LDS rows of image rows y0, y0 + step, y0 + s
            const int z0 = y0 >> 1, za = corner ? TH / 2 + h2 : (y0 + step) >> 1, zb = corner ? TH / 2 + hs : (y0 + s) >> 1;
            u32 p = pred1<INTERP>(hc[z0], hc[za], hc[h2 * HP + z0], hc[h2 * HP + za]);
            const bool xin = deep && tl.X0 + TW + s < W;
            const bool yin = (deep || !corner) && tl.Y0 + y0 + s < H;
            if (xin) hc[hs * HP + z0] = (u8)(hc[hs * HP + z0] + p);
            if (yin) hc[zb] = (u8)(hc[zb] + p);
            if (xin && yin) hc[hs * HP + zb] = (u8)(hc[hs * HP + zb] + p);
        }
    }
    if (tl.Y0 + TH < H && lane < ncx) {            // row cells
        const int x0 = lane << lstep;
        if (tl.X0 + x0 < W) {
            u8 *r0 = buf + (TH / 2) * S + x0, *r1 = buf + (TH / 2 + h2) * S + x0, *rs = buf + (TH / 2 + hs) * S + x0;
            const bool lastc = x0 + step == TW;    // right-hand corners are halo column 0
            u32 lb = lastc ? hc[TH / 2] : r0[step], rb = lastc ? hc[TH / 2 + h2] : r1[step];
            u32 p = pred1<INTERP>(r0[0], r1[0], lb, rb);
            const bool xin = tl.X0 + x0 + s < W;
            const bool yin = deep && tl.Y0 + TH + s < H;
            if (xin) r0[s] = (u8)(r0[s] + p);
            if (yin) rs[0] = (u8)(rs[0] + p);
            if (xin && yin) rs[s] = (u8)(rs[s] + p);
        }
    }
}

// One level (sub >= 2) of the tile body in LDS, in place.  CHECK = test every pixel against the image.
template <int INTERP, bool CHECK>
__device__ __forceinline__ void dec_cells(u8 *buf, int s, Tile tl, u32 W, u32 H)
{
    const int step = 2 * s, lstep = 31 - __clz(step);
    const int ncx = TW >> lstep, ncy = TH >> lstep, lncx = 31 - __clz(ncx);
    // natural LDS coordinates: x0 + step <= TW maps to itself, image row y <= TH to LDS row y / 2
    const int hs = s >> 1;
    for (int i = HGI_LANE; i < ncx * ncy; i += NL) {
        int x0 = (i & (ncx - 1)) << lstep, y0 = (i >> lncx) << lstep;
        if (CHECK && (tl.X0 + x0 >= W || tl.Y0 + y0 >= H)) continue;
        u8 *c = buf + (y0 >> 1) * S + x0;
        // right-hand corners of the last cell column are halo column 0 (transposed); y0 + step <= TH is a natural row
        const u8 *cr = x0 + step < TW ? c + step : buf + HCOL + (y0 >> 1);
        const int dn = x0 + step < TW ? s * S : s;
        u32 p = pred1<INTERP>(c[0], c[s * S], cr[0], cr[dn]);
        bool xin = !CHECK || tl.X0 + x0 + s < W, yin = !CHECK || tl.Y0 + y0 + s < H;
        if (xin) c[s] = (u8)(c[s] + p);
        if (yin) c[hs * S] = (u8)(c[hs * S] + p);
        if (xin && yin) c[hs * S + s] = (u8)(c[hs * S + s] + p);
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
    for (int i = HGI_LANE; i < ncx * ncy; i += NL) {
        int x0 = (i & (ncx - 1)) << lstep, y0 = (i >> lncx) << lstep;
        if (CHECK && (tl.X0 + x0 >= W || tl.Y0 + y0 >= H)) continue;
        u8 *c = buf + (y0 >> 1) * S + x0;
        u8 *rc = rbuf + (y0 >> 1) * S2 + (x0 >> 1);
        const u8 *rr = x0 + step < TW ? rc + s : rbuf + RCOL + (y0 >> 1);
        const int dn = x0 + step < TW ? s * S2 : s;
        u32 p = pred1<INTERP>(rc[0], rc[s * S2], rr[0], rr[dn]);
        bool xin = !CHECK || tl.X0 + x0 + s < W, yin = !CHECK || tl.Y0 + y0 + s < H;
        if (xin) {
            u32 q = quant1s<IDENT>(c[s], p, slut);
            c[s] = (u8)q;
            rc[hs] = (u8)(p + q);
        }
        if (yin) {
            u32 q = quant1s<IDENT>(c[hs * S], p, slut);
            c[hs * S] = (u8)q;
            rc[hs * S2] = (u8)(p + q);
        }
        if (xin && yin) {
            u32 q = quant1s<IDENT>(c[hs * S + s], p, slut);
            c[hs * S + s] = (u8)q;
            rc[hs * S2 + hs] = (u8)(p + q);
        }
    }
}

// finest level, generic: 16 px x 2 rows per lane, checked stores
template <int INTERP>
__device__ __noinline__ void dec_fine_generic(const u8 *buf, const u8 *__restrict__ fr, u8 *__restrict__ out, Tile tl,
                                              u32 W, u32 H, bool aligned)
{
    for (int i = HGI_LANE; i < (TH / 2) * CH; i += NL) {
        const int z = i >> LCH, y = 2 * z, x = 16 * (i & (CH - 1));
        const u32 gx = tl.X0 + x, gy = tl.Y0 + y;
        if (gx >= W || gy >= H) continue;
        const u8 *r0 = buf + z * S + x;
        uint4 E = *reinterpret_cast<const uint4 *>(r0);
        uint4 O = load16(fr, W, H, gx, gy + 1, aligned);     // odd rows never enter LDS
        uint4 F = *reinterpret_cast<const uint4 *>(r0 + S);
        u32 e16 = x + 16 < TW ? r0[16] : buf[HCOL + z], f16 = x + 16 < TW ? r0[S + 16] : buf[HCOL + z + 1];
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
__device__ __forceinline__ void enc_fine_generic(const u8 *buf, const u8 *rbuf, const u8 *slut, const u8 *__restrict__ fr,
                                              u8 *__restrict__ out, Tile tl, u32 W, u32 H, bool aligned)
{
    for (int i = HGI_LANE; i < (TH / 2) * CH; i += NL) {
        const int y = 2 * (i >> LCH), x = 16 * (i & (CH - 1));
        const u32 gx = tl.X0 + x, gy = tl.Y0 + y;
        if (gx >= W || gy >= H) continue;
        const u8 *r0 = buf + (y >> 1) * S + x;
        const u8 *c0 = rbuf + (y >> 1) * S2 + (x >> 1);
        uint4 E = *reinterpret_cast<const uint4 *>(r0);
        uint4 O = load16(fr, W, H, gx, gy + 1, aligned);     // odd rows never enter LDS
        uint2 c = *reinterpret_cast<const uint2 *>(c0);
        uint2 fl = *reinterpret_cast<const uint2 *>(c0 + S2);
        u32 c8 = x + 16 < TW ? c0[8] : rbuf[RCOL + (y >> 1)], f8 = x + 16 < TW ? c0[S2 + 8] : rbuf[RCOL + (y >> 1) + 1];
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
        __builtin_amdgcn_sched_barrier(0);   // row y + 1 after row y: this cold path must not set the kernel's register count
        if (gy + 1 < H) {
            uint4 o1 = make_uint4(quant4<IDENT>(O.x, pp0, slut), quant4<IDENT>(O.y, pp1, slut),
                                  quant4<IDENT>(O.z, pp2, slut), quant4<IDENT>(O.w, pp3, slut));
            store16(out, W, gx, gy + 1, o1, aligned);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// encode, coarse levels, split-phase: one LDS dependency chain per level
// ---------------------------------------------------------------------------------------------
// A dependent LDS round trip costs a few hundred cycles when the CU is busy.  Within a level nothing
// depends on anything of that level, so a pass issues all its reads first (unconditionally: an idle lane
// reads offset 0), then the predictions, then all table look-ups, and only the writes are conditional.
// How many cells a pass keeps in flight per lane is a register-budget decision (the kernel is held to
// 96 VGPRs = 5 waves per SIMD): two body cells at sub == 4, the halo cells in passes of their own.
struct CellAddr {
    int lt, rt, lb, rb;      // corners, half-resolution reconstruction plane
    int nx, ny, nxy;         // new pixels (x0+s, y0), (x0, y0+s), (x0+s, y0+s), full-resolution plane
    int rx, ry, rxy;         // the same three in the reconstruction plane
    bool xw, yw;             // write (x0+s, y0) / (x0, y0+s); (x0+s, y0+s) needs both
};

__device__ __forceinline__ CellAddr idle_cell()
{
    CellAddr c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, false, false};
    return c;
}

// i-th cell of level `s` inside the tile body (fast path: the body is inside the image)
__device__ __forceinline__ CellAddr enc_body_cell(int i, int s, bool on)
{
    const int step = 2 * s, hsub = s >> 1, lstep = 31 - __clz(step);
    const int ncx = TW >> lstep, lncx = 31 - __clz(ncx);
    const int x0 = (i & (ncx - 1)) << lstep, y0 = (i >> lncx) << lstep;
    const int c = (y0 >> 1) * S + x0, rc = (y0 >> 1) * S2 + (x0 >> 1);
    const bool lastc = x0 + step == TW;      // right-hand corners are halo column 0 (transposed)
    CellAddr a;
    a.lt = rc;
    a.rt = rc + s * S2;
    a.lb = lastc ? RCOL + (y0 >> 1) : rc + s;
    a.rb = lastc ? RCOL + (y0 >> 1) + s : rc + s * S2 + s;
    a.nx = c + s;
    a.ny = c + hsub * S;
    a.nxy = c + hsub * S + s;
    a.rx = rc + hsub;
    a.ry = rc + hsub * S2;
    a.rxy = rc + hsub * S2 + hsub;
    a.xw = a.yw = true;
    return on ? a : idle_cell();
}

// j-th halo cell of level `s`: j = 0..ncy are the column cells x0 == TW (y0 = j*step; y0 == TH is the
// corner cell), j = ncy+1 .. ncy+ncx the row cells y0 == TH.  They recompute, bit-identically, what the
// right / lower neighbour tiles compute for themselves.  Level-`sub` halo pixels off the tile edge are
// needed only for sub >= 4 (`deep`); at sub == 2 only those on the edge itself.
__device__ __forceinline__ CellAddr enc_halo_cell(int j, int s, Tile tl, u32 W, u32 H)
{
    const int step = 2 * s, hsub = s >> 1, lstep = 31 - __clz(step);
    const int ncx = TW >> lstep, ncy = TH >> lstep;
    const bool deep = s >= 4;
    const int hs = hmap(s), h2 = hmap(step);
    const bool col = j <= ncy;
    const int y0 = j << lstep;
    const bool corner = y0 == TH;
    const int z0 = y0 >> 1, za = corner ? TH / 2 + h2 : (y0 + step) >> 1, zb = corner ? TH / 2 + hs : (y0 + s) >> 1;
    const int x0 = (j - ncy - 1) << lstep, w0 = x0 >> 1;
    const bool lastc = x0 + step == TW;
    const bool active = col ? (tl.X0 + TW < W && tl.Y0 + y0 < H) : (j <= ncy + ncx && tl.Y0 + TH < H && tl.X0 + x0 < W);
    CellAddr a;
    a.lt = col ? RCOL + z0 : (TH / 2) * S2 + w0;
    a.rt = col ? RCOL + za : (TH / 2 + h2) * S2 + w0;
    a.lb = col ? RCOL + h2 * HP2 + z0 : (lastc ? RCOL + TH / 2 : (TH / 2) * S2 + w0 + s);
    a.rb = col ? RCOL + h2 * HP2 + za : (lastc ? RCOL + TH / 2 + h2 : (TH / 2 + h2) * S2 + w0 + s);
    a.nx = col ? HCOL + hs * HP + z0 : (TH / 2) * S + x0 + s;
    a.ny = col ? HCOL + zb : (TH / 2 + hs) * S + x0;
    a.nxy = col ? HCOL + hs * HP + zb : (TH / 2 + hs) * S + x0 + s;
    a.rx = col ? RCOL + hs * HP2 + z0 : (TH / 2) * S2 + w0 + hsub;
    a.ry = col ? RCOL + zb : (TH / 2 + hs) * S2 + w0;
    a.rxy = col ? RCOL + hs * HP2 + zb : (TH / 2 + hs) * S2 + w0 + hsub;
    a.xw = col ? (deep && tl.X0 + TW + s < W) : (tl.X0 + x0 + s < W);
    a.yw = col ? ((deep || !corner) && tl.Y0 + y0 + s < H) : (deep && tl.Y0 + TH + s < H);
    return active ? a : idle_cell();
}

struct CellVal {
    u32 lt, rt, lb, rb, ax, ay, axy;
};

__device__ __forceinline__ CellVal cell_load(const u8 *buf, const u8 *rbuf, const CellAddr &a)
{
    CellVal v = {rbuf[a.lt], rbuf[a.rt], rbuf[a.lb], rbuf[a.rb], buf[a.nx], buf[a.ny], buf[a.nxy]};
    return v;
}

template <int INTERP, bool IDENT>
__device__ __forceinline__ void cell_finish(u8 *buf, u8 *rbuf, const u8 *slut, const CellAddr &a, const CellVal &v)
{
    const u32 p = pred1<INTERP>(v.lt, v.rt, v.lb, v.rb);
    const u32 orig[3] = {v.ax, v.ay, v.axy}, pp[3] = {p, p, p};
    u32 res[3];
    quant_batch<IDENT, 3>(orig, pp, res);
    const u32 qx = res[0], qy = res[1], qxy = res[2];
    if (a.xw) {
        buf[a.nx] = (u8)qx;
        rbuf[a.rx] = (u8)(p + qx);
    }
    if (a.yw) {
        buf[a.ny] = (u8)qy;
        rbuf[a.ry] = (u8)(p + qy);
    }
    if (a.xw && a.yw) {
        buf[a.nxy] = (u8)qxy;
        rbuf[a.rxy] = (u8)(p + qxy);
    }
}

// All halo cells of level `s` in one pass (<= 49 lanes); the ragged-tile path uses it after its checked
// body-cell loop, the interior path folds the same cells into its own chains below.
template <int INTERP, bool IDENT>
__device__ __forceinline__ void enc_halo_pass(u8 *buf, u8 *rbuf, const u8 *slut, int s, Tile tl, u32 W, u32 H)
{
    const int lstep = 31 - __clz(2 * s), nhalo = (TH >> lstep) + 1 + (TW >> lstep);
    const CellAddr a = HGI_LANE < nhalo ? enc_halo_cell(HGI_LANE, s, tl, W, H) : idle_cell();
    const CellVal v = cell_load(buf, rbuf, a);
    cell_finish<INTERP, IDENT>(buf, rbuf, slut, a, v);
}

// One coarse level (sub >= 4) of an interior tile, body cells and halo cells in one chain.
// sub == 4: 128 body cells (two per lane), then the 25 halo cells; sub >= 8: <= 32 body cells on lanes 0..31,
// the halo cells on lanes 32..63.
template <int INTERP, bool IDENT>
__device__ __forceinline__ void enc_level_coarse_fast(u8 *buf, u8 *rbuf, const u8 *slut, int s, Tile tl, u32 W, u32 H)
{
    const int lane = HGI_LANE;
    if (s == 4) {
        constexpr int ncells4 = (TW / 8) * (TH / 8);
        const CellAddr a0 = enc_body_cell(lane, 4, lane < ncells4), a1 = enc_body_cell(lane + NL, 4, lane + NL < ncells4);
        const CellVal v0 = cell_load(buf, rbuf, a0), v1 = cell_load(buf, rbuf, a1);
        cell_finish<INTERP, IDENT>(buf, rbuf, slut, a0, v0);
        cell_finish<INTERP, IDENT>(buf, rbuf, slut, a1, v1);
        // the 25 halo cells in a pass of their own: three cells in flight per lane was the register peak
        LDS_ORDER();
        HGI_MARK("halo");
        enc_halo_pass<INTERP, IDENT>(buf, rbuf, slut, 4, tl, W, H);
    } else {
        const int step = 2 * s, lstep = 31 - __clz(step);
        const int ncells = (TW >> lstep) * (TH >> lstep), nhalo = (TH >> lstep) + 1 + (TW >> lstep);
        const CellAddr a = lane < 32 ? enc_body_cell(lane, s, lane < ncells)
                                     : (lane - 32 < nhalo ? enc_halo_cell(lane - 32, s, tl, W, H) : idle_cell());
        const CellVal v = cell_load(buf, rbuf, a);
        cell_finish<INTERP, IDENT>(buf, rbuf, slut, a, v);
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

// Row pair (image rows 2p, 2p + 1) a lane owns in iteration `it` of the finest level, p = fine_pair0 + 8 * it:
// the octets of a half-wave take pairs 0,2,4,6 / 1,3,5,7, i.e. LDS rows two apart (bank skew).
// The staging loads of the odd rows use the same map, which is what lets those rows stay in registers.
__device__ __forceinline__ int fine_pair0() { return 2 * ((HGI_LANE >> LCH) & 3) + (HGI_LANE >> 5); }
constexpr int NFINE = (TH / 2) * CH / NL;   // fine-level iterations = odd rows a lane holds

// Everything a tile stages: all loads are issued before the first LDS write.
struct Stage {
    v4u e[TH / 16];     // even rows: 16 B per lane and row group (8 full 128-B lines per wave instruction) -> LDS
    v4u o[NFINE];       // odd rows, fine-level lane map: stay in registers until the finest level
    v4u hv;             // halo rows TH + {0,4,8,..}: full lines
    v3u x0;             // halo columns: one lane per (even) row; this 12-B load holds offsets 0/4/8
    u32 d16, d32, d64;  // ... and single dwords supply offsets 16/32/64
    bool zero4, zero8;  // wave-uniform: column offsets 4 / 8 lie beyond the image (their bytes of x0 are cleared at commit)
};

// Rows below the image return 0 from the buffer range check; columns right of the image are masked
// with wave-uniform tests.
// RAGGED tiles that finish in the generic fine level fetch their odd rows there; bottom-ragged tiles
// (ODD_CHECKED) keep the fast fine level and load them here, row offsets in voffset like the even rows.
template <bool RAGGED, bool ODD_CHECKED = false>
__device__ __forceinline__ void stage_issue(Stage &st, const Buf &b, Tile tl, int k, int nh)
{
    const int lane = HGI_LANE, c = lane & (CH - 1), r = lane >> LCH;
    const u32 W = __builtin_amdgcn_readfirstlane(b.W);   // soffset operands must be provably uniform
    const u32 voff = b.base + 2 * r * W + 16 * c;                       // even rows 2 * (r + 8 j)
    const u32 vodd = b.base + (2 * fine_pair0() + 1) * W + 16 * c;      // odd rows 2 * (pair0 + 8 it) + 1
    // Interior tiles: every chunk is inside the image.  Ragged tiles (checked path): rows below the
    // image come back as 0 from the range check (the row offsets go through voffset there, which is
    // what the check sees), chunks right of it are masked.
    const bool cin = tl.X0 + 16 * c < W;
    // the chunk that straddles the right edge (widths that are not multiples of 16) holds the next row's
    // first pixels behind the image's last ones: cleared after the load, like everything else outside
    const bool narrow = RAGGED && tl.X0 + TW > W;   // wave-uniform: this tile straddles the right edge
    const v4u cm = narrow ? chunk_mask((int)W - (int)(tl.X0 + 16 * c)) : v4u{~0u, ~0u, ~0u, ~0u};
    // Issue order.  HGI_HALO_FIRST: the few halo loads go out before the tile's own rows, so that they reach L2 ahead of
    // the body loads of the tiles that own those lines (which the band order dispatches later).
    auto issue_body = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < TH / 16; ++j) {
        st.e[j] = v4u{0, 0, 0, 0};
        if (RAGGED) {
            if (cin) st.e[j] = __builtin_amdgcn_raw_buffer_load_b128(b.rs, voff + j * 16 * W, 0, 0);
            if (narrow) st.e[j] &= cm;
        } else {
            st.e[j] = __builtin_amdgcn_raw_buffer_load_b128(b.rs, voff, j * 16 * W, HGI_LOAD_AUX);
        }
    }
#pragma unroll
    for (int j = 0; j < NFINE; ++j) {
        st.o[j] = v4u{0, 0, 0, 0};
        if (!RAGGED && !HGI_ODD_LATE) st.o[j] = __builtin_amdgcn_raw_buffer_load_b128(b.rs, vodd, j * 16 * W, HGI_ODD_LOAD_AUX);
        if (RAGGED && ODD_CHECKED && cin) st.o[j] = __builtin_amdgcn_raw_buffer_load_b128(b.rs, vodd + j * 16 * W, 0, HGI_ODD_LOAD_AUX);
        if (narrow) st.o[j] &= cm;
    }
    };
    auto issue_halo = [&]() __attribute__((always_inline)) {
    st.hv = v4u{0, 0, 0, 0};
    if (lane < nh * CH && cin) st.hv = __builtin_amdgcn_raw_buffer_load_b128(b.rs, b.base + (TH + hoff(r)) * W + 16 * c, 0, 0);
    if (narrow) st.hv &= cm;
    const int hy = lane < TH / 2 ? 2 * lane : TH + hoff(lane - TH / 2);
    const u32 xo = b.base + hy * W + TW;
    const u32 xr = tl.X0 + TW;              // first column right of the tile
    st.x0 = v3u{0, 0, 0};
    st.d16 = st.d32 = st.d64 = 0;
    st.zero4 = st.zero8 = false;
    if (lane < TH / 2 + nh) {
        if (xr < W) st.x0 = __builtin_amdgcn_raw_buffer_load_b96(b.rs, xo, 0, 0);
        // offsets 4 / 8 can lie beyond the image when the width is not a multiple of 16: cleared when the column is
        // committed (stage_commit), not here -- touching the loaded registers now would wait for every load issued so far
        st.zero4 = xr + 4 >= W;
        st.zero8 = xr + 8 >= W;
        // Column offset `off` is only ever touched on rows = 0 (mod off): level off / 2 reads it as a corner of the
        // halo cells (rows = 0 mod step = off), level off codes it (rows = 0 mod s = off).  The other lanes do not
        // fetch that line at all.
        const bool all = HGI_HALO_ALL_ROWS;
        if (k >= 4 && (all || !(hy & 15)) && xr + 16 < W) st.d16 = __builtin_amdgcn_raw_buffer_load_b32(b.rs, xo + 16, 0, 0);
        if (k >= 5 && (all || !(hy & 31)) && xr + 32 < W) st.d32 = __builtin_amdgcn_raw_buffer_load_b32(b.rs, xo + 32, 0, 0);
        if (k >= 6 && (all || !(hy & 63)) && xr + 64 < W) st.d64 = __builtin_amdgcn_raw_buffer_load_b32(b.rs, xo + 64, 0, 0);
    }
    };
    if (HGI_HALO_FIRST) {
        issue_halo();
        issue_body();
    } else {
        issue_body();
        issue_halo();
    }
}

// The odd rows of an interior tile -- half its bytes, first needed by the finest level -- are requested only after the even
// rows have been committed: the tile waits for half as much before its first level, and the rest flies while the coarse
// levels compute.  What that buys depends on how deep the launch is (tools/ab.py, profiles/r03_ab_odd_late.txt): a lone
// 8192^2 frame -11 % / -6 %, 4096^2 -8 % / -9 %, 16384^2 level 8 -5 % / -3 %, 16 x 4096^2 -4 % / -2 %, 64 x 4096^2 -0.6 % /
// +0.4 % -- fill and drain of a launch are made of tile lifetimes, the steady state is not.  Issuing everything up front and
// waiting for the even rows only (halo loads first, HGI_HALO_FIRST) gains a third of that: what helps is fewer requests
// queued at once, not the shorter wait alone.
__device__ __forceinline__ void stage_issue_odd(Stage &st, const Buf &b)
{
    const int c = HGI_LANE & (CH - 1);
    const u32 W = __builtin_amdgcn_readfirstlane(b.W);
    const u32 vodd = b.base + (2 * fine_pair0() + 1) * W + 16 * c;
#pragma unroll
    for (int j = 0; j < NFINE; ++j) st.o[j] = __builtin_amdgcn_raw_buffer_load_b128(b.rs, vodd, j * 16 * W, HGI_ODD_LOAD_AUX);
}

// the even bytes of a 16-B row chunk: its eight even/even lattice points
__device__ __forceinline__ v2u even_bytes(v4u a)
{
    v2u r = {__builtin_amdgcn_perm(a.y, a.x, 0x06040200u), __builtin_amdgcn_perm(a.w, a.z, 0x06040200u)};
    return r;
}

// LATTICE (encode): the half-resolution plane starts as a copy of the even/even pixels -- originals, i.e.
// the reconstruction of the base lattice (src/encoder.rs:26-37), zeros beyond the image -- written from the
// same registers.  Points of finer lattices hold originals until their level codes them; nothing reads
// them earlier.
template <bool LATTICE>
__device__ __forceinline__ void stage_commit(u8 *buf, u8 *rbuf, const Stage &st, int nh)
{
    const int lane = HGI_LANE, c = lane & (CH - 1), r = lane >> LCH;
#pragma unroll
    for (int j = 0; j < TH / 16; ++j) {
        *reinterpret_cast<v4u *>(buf + (r + 8 * j) * S + 16 * c) = st.e[j];
        if (LATTICE) *reinterpret_cast<v2u *>(rbuf + (r + 8 * j) * S2 + 8 * c) = even_bytes(st.e[j]);
    }
    if (lane < nh * CH) {
        *reinterpret_cast<v4u *>(buf + (TH / 2 + r) * S + 16 * c) = st.hv;
        if (LATTICE) *reinterpret_cast<v2u *>(rbuf + (TH / 2 + r) * S2 + 8 * c) = even_bytes(st.hv);
    }
    if (lane < TH / 2 + nh) {
        // transposed halo columns: slot {0..5} <- offsets {0, 4, 8, 16, 32, 64}, one byte per LDS row
        const u32 v[HR] = {st.x0.x, st.zero4 ? 0u : st.x0.y, st.zero8 ? 0u : st.x0.z, st.d16, st.d32, st.d64};
        u8 *h = buf + HCOL + lane, *h2 = rbuf + RCOL + lane;
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            h[i * HP] = (u8)v[i];
            if (LATTICE) h2[i * HP2] = (u8)v[i];
        }
    }
}

// the same initialisation out of LDS, for tiles staged by stage_tile_generic
__device__ __forceinline__ void lattice_from_buf(const u8 *buf, u8 *rbuf, int nh)
{
    for (int i = HGI_LANE; i < (TH / 2 + nh) * CH; i += NL) {
        const int rr = i >> LCH, c = i & (CH - 1);
        *reinterpret_cast<v2u *>(rbuf + rr * S2 + 8 * c) = even_bytes(*reinterpret_cast<const v4u *>(buf + rr * S + 16 * c));
    }
    for (int i = HGI_LANE; i < HR * (TH / 2 + nh); i += NL) {
        const int hc = i / (TH / 2 + nh), rr = i - hc * (TH / 2 + nh);
        rbuf[RCOL + hc * HP2 + rr] = buf[HCOL + hc * HP + rr];
    }
}

// [x.b0, y.b0, z.b0, w.b0] of a 16-B row chunk: the four stride-4 lattice bytes
__device__ __forceinline__ u32 gather_b0(v4u a)
{
    return __builtin_amdgcn_perm(a.y, a.x, 0x0c0c0400u) | __builtin_amdgcn_perm(a.w, a.z, 0x04000c0cu);
}

// level sub == 2 of the tile body: four 4x4 cells (16 px x rows y0, y0+2; corners also from y0+4) per lane.
// EDGE: the tile crosses the image edge after `rows` rows and / or `cols` columns (a multiple of 16: a lane's
// chunk is entirely inside or outside).  Pixels beyond were staged as zeros and must stay zeros: they are the
// out-of-image corners of the cells inside (src/interpolator.rs:75-82).
// EDGE == 1: the tile crosses only the lower edge (full width inside, even height): row masks suffice.
// EDGE == 2: any ragged tile.
template <int INTERP, int EDGE = 0>
__device__ __forceinline__ void dec_level2_fast(u8 *buf, int rows = TH, int cols = TW)
{
    // (TH / 4) * CH lane tasks: whole iterations of the wave for 32- and 64-row tiles, half a wave for 16-row tiles (the
    // idle lanes compute on task 0 and do not store)
    constexpr int N2 = (TH / 4) * CH;
#pragma unroll
    for (int it = 0; it < (N2 + NL - 1) / NL; ++it) {
        const bool on = N2 % NL == 0 || HGI_LANE + it * NL < N2;
        const int i = on ? HGI_LANE + it * NL : 0;
        u8 *r0 = buf + 2 * (i >> LCH) * S + 16 * (i & (CH - 1));      // image rows y0, y0 + 2, y0 + 4 = LDS rows z0 ..
        v4u A = *reinterpret_cast<const v4u *>(r0);
        v4u B = *reinterpret_cast<const v4u *>(r0 + S);
        v4u C = *reinterpret_cast<const v4u *>(r0 + 2 * S);
        // ninth corner of each row: first byte of the next chunk = lane + 1, or the halo column
        const int z0 = 2 * (i >> LCH);
        const bool last = (i & (CH - 1)) == CH - 1;
        u32 a16 = from_next_lane(A.x), c16 = from_next_lane(C.x);
        const u32 ha = buf[HCOL + z0], hc = buf[HCOL + z0 + 2];
        a16 = last ? ha : a16;
        c16 = last ? hc : c16;
        u32 ct = gather_b0(A), cb = gather_b0(C);
        u32 P = ct;
        if (INTERP == kInterpCrossed)
            P = pred4_crossed(ct, cb, __builtin_amdgcn_alignbyte(a16, ct, 1), __builtin_amdgcn_alignbyte(c16, cb, 1));
        u32 a0 = A.x, a1 = A.y, a2 = A.z, a3 = A.w, b0 = B.x, b1 = B.y, b2 = B.z, b3 = B.w;
        HGI_ADDB(a0, 2, P, 0); HGI_ADDB(a1, 2, P, 1); HGI_ADDB(a2, 2, P, 2); HGI_ADDB(a3, 2, P, 3);
        HGI_ADDB(b0, 0, P, 0); HGI_ADDB(b1, 0, P, 1); HGI_ADDB(b2, 0, P, 2); HGI_ADDB(b3, 0, P, 3);
        HGI_ADDB(b0, 2, P, 0); HGI_ADDB(b1, 2, P, 1); HGI_ADDB(b2, 2, P, 2); HGI_ADDB(b3, 2, P, 3);
        v4u An = {a0, a1, a2, a3}, Bn = {b0, b1, b2, b3};
        if (EDGE == 2 && cols < TW) {   // columns beyond the image stay zero
            const v4u m = chunk_mask(cols - 16 * (i & (CH - 1)));
            An &= m;
            Bn &= m;
        }
        if (on && (!EDGE || 2 * z0 < rows)) *reinterpret_cast<v4u *>(r0) = An;
        if (on && (!EDGE || 2 * z0 + 2 < rows)) *reinterpret_cast<v4u *>(r0 + S) = Bn;
    }
}

// finest level: 16 px x 2 rows per lane; even rows from LDS, odd rows from the registers they were loaded
// into, packed VALU, 16-B buffer stores
// EDGE: only the `rows` x `cols` part of the tile that lies inside the image is stored.
template <int INTERP, int EDGE = 0>
__device__ __forceinline__ void dec_fine_fast(const u8 *buf, const Buf &b, const v4u (&odd)[NFINE], int rows = TH, int cols = TW)
{
    const int lane = HGI_LANE;
    const int rp0 = fine_pair0();
    const bool last = (lane & (CH - 1)) == CH - 1;
    const u8 *r0 = buf + rp0 * S + 16 * (lane & (CH - 1));
    const u8 *h0 = buf + HCOL + rp0;
    u32 voff = b.base + 2 * rp0 * b.W + 16 * (lane & (CH - 1));
#pragma unroll
    for (int it = 0; it < NFINE; ++it, r0 += (NL / CH) * S, h0 += NL / CH, voff += 2 * (NL / CH) * b.W) {
        v4u E = *reinterpret_cast<const v4u *>(r0);
        v4u O = odd[it];
        v4u F = *reinterpret_cast<const v4u *>(r0 + S);
        // ninth corner of each lattice row: first byte of the next chunk = lane + 1, or the halo column
        u32 e16 = from_next_lane(E.x), f16 = from_next_lane(F.x);
        const u32 he = h0[0], hf = h0[1];
        e16 = last ? he : e16;
        f16 = last ? hf : f16;
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
        if (EDGE == 2) {
            const int y = 2 * (rp0 + it * (NL / CH));
            store_rows_edge(r0v, r1v, b.rd, voff, __builtin_amdgcn_readfirstlane(b.W), cols - 16 * (lane & (CH - 1)), y < rows, y + 1 < rows);
        } else {   // EDGE == 1: row pairs below the image are dropped by the buffer range check (even height)
            store_row_pair(r0v, r1v, b.rd, voff, __builtin_amdgcn_readfirstlane(b.W));
        }
    }
}

// encode, level sub == 2 of the tile body: four cells per lane; corners from the half-resolution
// reconstruction lattice (rbuf), originals in / residuals out in buf, new reconstructions into rbuf
template <int INTERP, bool IDENT, int EDGE = 0>
__device__ __forceinline__ void enc_level2_fast(u8 *buf, u8 *rbuf, const u8 *slut, Tile tl, u32 W, u32 H)
{
    const int rows = EDGE ? (int)(H - tl.Y0) : TH, cols = EDGE ? (int)(W - tl.X0) : TW;   // see dec_level2_fast
    // One iteration (64 lanes x 4 cells) at a time, all its reads before its first write, and the level's
    // halo cells in a pass of their own afterwards: with the odd rows parked in registers the level is
    // the kernel's register high-water mark, and 96 VGPRs (5 waves per SIMD) beat the longer chains that
    // batching both iterations and the halo cells into one bought at 12 waves per CU.
    constexpr int N2 = (TH / 4) * CH, NIT = (N2 + NL - 1) / NL;      // see dec_level2_fast
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const bool on = N2 % NL == 0 || HGI_LANE + it * NL < N2;
        const int i = on ? HGI_LANE + it * NL : 0;
        const int g = i >> LCH, c = i & (CH - 1);
        u8 *r0 = buf + 2 * g * S + 16 * c;
        u8 *q0 = rbuf + 2 * g * S2 + 8 * c;
        const v4u A = *reinterpret_cast<const v4u *>(r0);
        const v4u B = *reinterpret_cast<const v4u *>(r0 + S);
        const v2u cu = *reinterpret_cast<const v2u *>(q0);
        const v2u cl = *reinterpret_cast<const v2u *>(q0 + 2 * S2);
        const u32 hu = rbuf[RCOL + 2 * g], hl = rbuf[RCOL + 2 * g + 2];
        LDS_ORDER();
        const bool last = c == CH - 1;
        u32 c8 = from_next_lane(cu.x), l8 = from_next_lane(cl.x);
        c8 = last ? hu : c8;
        l8 = last ? hl : l8;
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
        if (EDGE == 2 && cols < TW) {   // columns beyond the image stay zero in both planes
            const v4u m = chunk_mask(cols - 16 * c);
            const v2u m2 = chunk_mask2(cols - 16 * c);
            An &= m;
            Bn &= m;
            n0 &= m2;
            n1 &= m2;
        }
        if (on && (!EDGE || 4 * g < rows)) {
            *reinterpret_cast<v4u *>(r0) = An;
            *reinterpret_cast<v2u *>(q0) = n0;
        }
        if (on && (!EDGE || 4 * g + 2 < rows)) {
            *reinterpret_cast<v4u *>(r0 + S) = Bn;
            *reinterpret_cast<v2u *>(q0 + S2) = n1;
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the next iteration's reads behind this one's arithmetic
    }
    LDS_ORDER();
    HGI_MARK("halo");
    enc_halo_pass<INTERP, IDENT>(buf, rbuf, slut, 2, tl, W, H);
}

template <int INTERP, bool IDENT, int EDGE = 0>
__device__ __forceinline__ void enc_fine_fast(const u8 *buf, const u8 *rbuf, const u8 *slut, const Buf &b,
                                              const v4u (&odd)[NFINE], int rows = TH, int cols = TW)
{
    const int lane = HGI_LANE;
    const int rp0 = fine_pair0();
    const bool last = (lane & (CH - 1)) == CH - 1;
    const u8 *r0 = buf + rp0 * S + 16 * (lane & (CH - 1));
    const u8 *c0 = rbuf + rp0 * S2 + 8 * (lane & (CH - 1));
    const u8 *h0 = rbuf + RCOL + rp0;
    u32 voff = b.base + 2 * rp0 * b.W + 16 * (lane & (CH - 1));
    const u32 Ws = __builtin_amdgcn_readfirstlane(b.W);
#ifndef HGI_FINE_BATCH
#define HGI_FINE_BATCH 2
#endif
    constexpr int NIT = NFINE, PAIR = NFINE >= HGI_FINE_BATCH ? HGI_FINE_BATCH : 1;   // row-pair groups per LDS dependency chain
    static_assert(NIT % PAIR == 0, "fine level iterations come in pairs");
#pragma unroll
    for (int it = 0; it < NIT; it += PAIR) {
        v4u E_[PAIR], O_[PAIR];
        uint2 c_[PAIR], f_[PAIR];
        u32 hc_[PAIR], hf_[PAIR];
#pragma unroll
        for (int j = 0; j < PAIR; ++j) {
            const u8 *r = r0 + j * (NL / CH) * S, *c = c0 + j * (NL / CH) * S2, *h = h0 + j * (NL / CH);
            E_[j] = *reinterpret_cast<const v4u *>(r);
            O_[j] = odd[it + j];
            c_[j] = *reinterpret_cast<const uint2 *>(c);
            f_[j] = *reinterpret_cast<const uint2 *>(c + S2);
            hc_[j] = h[0];
            hf_[j] = h[1];
        }
#pragma unroll
        for (int j = 0; j < PAIR; ++j) {
            const v4u E = E_[j], O = O_[j];
            const uint2 c = c_[j], fl = f_[j];
            // ninth corner of each lattice row: lane + 1, or the transposed halo column
            u32 c8 = from_next_lane(c.x), f8 = from_next_lane(fl.x);
            c8 = last ? hc_[j] : c8;
            f8 = last ? hf_[j] : f8;
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
            const u32 vo = voff + j * 2 * (NL / CH) * Ws;
            if (EDGE == 2) {
                const int y = 2 * (rp0 + (it + j) * (NL / CH));
                store_rows_edge(o0, o1, b.rd, vo, Ws, cols - 16 * (lane & (CH - 1)), y < rows, y + 1 < rows);
            } else {
                store_row_pair(o0, o1, b.rd, vo, Ws);
            }
        }
        r0 += PAIR * (NL / CH) * S;
        c0 += PAIR * (NL / CH) * S2;
        h0 += PAIR * (NL / CH);
        voff += PAIR * 2 * (NL / CH) * Ws;
    }
}

// Descriptors are built from values forced wave-uniform (readfirstlane): once they are loop-carried
// the compiler can no longer prove it, and would wrap every buffer access in a waterfall loop.
__device__ __forceinline__ u8 *uniform_ptr(const u8 *p)
{
    const u64 a = reinterpret_cast<u64>(p);
    const u32 lo = __builtin_amdgcn_readfirstlane((u32)a), hi = __builtin_amdgcn_readfirstlane((u32)(a >> 32));
    return reinterpret_cast<u8 *>(((u64)hi << 32) | lo);
}

// `tail` (0 or 3) extra records on the READ side: when rows are not a multiple of 4 bytes a dword of a load can
// straddle the end of the frame, and the range check would drop it whole -- valid bytes included.  The host grants
// the 3 bytes only when reading them is safe (fused_geom); what they hold lies right of the image and is masked.
// Dwords that START at or beyond W * H are still out of range, so rows below the image keep reading as zero.
__device__ __forceinline__ Buf make_buf(const u8 *fr, u8 *out, u32 W, u32 H, Tile tl, u32 tail)
{
    Buf b;
    const u32 bytes = W * H;   // the host only selects the fast path when this (plus the halo) fits 32 bits
    b.rs = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(fr), 0, bytes + tail, 0x00020000);
    b.rd = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(out), 0, bytes, 0x00020000);
    b.W = W;
    b.base = __builtin_amdgcn_readfirstlane(tl.Y0 * W + tl.X0);
    return b;
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// Tile lists.  `full_x` x `full_y` tiles per frame lie entirely inside the image: the fast kernels walk
// those; the checked path takes the rest (right column first, then the bottom rows).
// (FastDiv / make_fastdiv / fdiv: hgi_fastdiv.h -- the block -> tile index math divides by launch-wide constants only)
struct TileGrid {
    u32 tiles_x, tiles_y;   // all tiles of a frame
    u32 full_x, full_y;     // tiles whose body is inside the image (0 x 0 when the fast path is off)
    u32 nfast, nedge;       // totals over the batch
    u32 reverse;            // walk the interior tile list backwards (speed only: see launch_decode_fused)
    u32 band;               // tile rows per band of the column-major walk (fast_tile)
    u32 xmode;              // how the band-ordered tile list is dealt to the XCDs (block_role)
    // derived by finish_grid() on the host, so that the kernels neither divide nor re-derive launch constants:
    u32 ex, nf;             // interior tile columns / interior tiles the walk runs on
    u32 tpf, P, nfull, rem_rows;   // tiles per frame; per band; in a frame's whole bands; rows of its last, shorter band
    u32 rr_own, rr_tail0;   // round-robin dealing: blocks per XCD that belong to whole rounds of eight bands; first tile behind them
    FastDiv fd_tpf, fd_P, fd_band, fd_rem, fd_ex;
#ifdef HGI_TIMELINE
    u64 *timeline;          // experiment builds (tools/timeline.py): eight u64 per block -- start, staged, end, hardware id, entry
#endif
};

__device__ __forceinline__ Tile fast_tile(u32 t, const TileGrid &g)
{
    t = __builtin_amdgcn_readfirstlane(g.reverse ? g.nf - 1u - t : t);
    Tile tl;
    tl.frame = fdiv(t, g.fd_tpf);
    const u32 tt = t - tl.frame * g.tpf;
    u32 ty, tx;
#if HGI_TILE_ORDER == 0          // row-major (experiment)
    ty = fdiv(tt, g.fd_ex);
    tx = tt - ty * g.ex;
#else
    // Bands of g.band tile rows, column-major inside a band: x-neighbours are dispatched `rows` tiles apart,
    // y-neighbours next to each other.  The last band of a frame takes the rows that are left.
    {
        u32 rows, row0, r;
        if (tt < g.nfull) {
            const u32 band = fdiv(tt, g.fd_P);
            r = tt - band * g.P;
            row0 = band * g.band;
            rows = g.band;
            tx = fdiv(r, g.fd_band);
        } else {
            rows = g.rem_rows;
            r = tt - g.nfull;
            row0 = g.full_y - rows;
            tx = fdiv(r, g.fd_rem);
        }
        ty = row0 + (r - tx * rows);
    }
#endif
    tl.X0 = tx * TW;
    tl.Y0 = ty * TH;
    return tl;
}

__device__ __forceinline__ Tile edge_tile(u32 e, const TileGrid &g)
{
    const u32 right = (g.tiles_x - g.full_x) * g.tiles_y;          // tiles with tx >= full_x
    const u32 epf = right + g.full_x * (g.tiles_y - g.full_y);     // + tiles with ty >= full_y, tx < full_x
    Tile tl;
    tl.frame = e / epf;
    u32 i = e - tl.frame * epf, tx, ty;
    if (i < right) {
        const u32 w = g.tiles_x - g.full_x;
        ty = i / w;
        tx = g.full_x + (i - ty * w);
    } else {
        i -= right;
        ty = g.full_y + i / g.full_x;
        tx = i % g.full_x;
    }
    tl.X0 = tx * TW;
    tl.Y0 = ty * TH;
    return tl;
}

struct TileCtx {
    Tile tl;
    Buf b;
};

// Every launch constant the prologue of an interior tile reads is asked for at the kernel's first instruction, in one go:
// left to itself the compiler places the scalar loads next to their uses, in three dependent rounds (block role -> tile
// index -> addresses), each a trip to the scalar cache.  Worth 0.1-0.2 us of a tile's lifetime, which shows where launches
// are shallow (tools/ab.py, profiles/r03_ab_args_early.txt: 16384^2 level 8 encode -1.7 %, lone 4096^2 -2.7 % / -1.6 %,
// 1024^2 -2.3 % / -2.8 %, 64 x 4096^2 0).  -DHGI_ARGS_EARLY=0: the compiler's placement.
#ifndef HGI_ARGS_EARLY
#define HGI_ARGS_EARLY 1
#endif
__device__ __forceinline__ void args_early(const u8 *src, const u8 *dst, const Frames &f, const TileGrid &g, u32 aligned, u32 k)
{
#if HGI_ARGS_EARLY
    asm volatile("" ::"s"(k), "s"(g.full_x), "s"(g.full_y), "s"(g.tiles_x), "s"(g.tiles_y));
    asm volatile("" ::"s"(src), "s"(dst), "s"(f.width), "s"(f.height), "s"(f.frame_stride), "s"(aligned), "s"(g.nedge), "s"(g.nf), "s"(g.xmode),
                 "s"(g.rr_own), "s"(g.rr_tail0), "s"(g.P), "s"(g.fd_P.m), "s"(g.fd_P.s1), "s"(g.fd_P.s2), "s"(g.reverse), "s"(g.tpf), "s"(g.fd_tpf.m),
                 "s"(g.fd_tpf.s1), "s"(g.fd_tpf.s2), "s"(g.nfull), "s"(g.rem_rows), "s"(g.band), "s"(g.ex), "s"(g.fd_band.m), "s"(g.fd_band.s1),
                 "s"(g.fd_band.s2), "s"(g.fd_rem.m), "s"(g.fd_rem.s1), "s"(g.fd_rem.s2));
#endif
}

__device__ __forceinline__ TileCtx fast_ctx(u32 t, const u8 *src, u8 *dst, const Frames &f, const TileGrid &g, u32 tail)
{
    TileCtx c;
    c.tl = fast_tile(t, g);
    c.b = make_buf(src + (size_t)c.tl.frame * f.frame_stride, dst + (size_t)c.tl.frame * f.frame_stride, f.width,
                   f.height, c.tl, tail);
    return c;
}

// ---- decode ---------------------------------------------------------------------------------------
// Seeds: lattice points = 0 (mod 2^k) of a deeper pyramid come from the already coded coarser pyramid
// (compact planes, one byte per lattice point).  A tile touches (TW >> k) + 2 by (TH >> k) + 2 of them
// (<= 64 for every k the seeded kernels run with), one per lane; the loads are issued together with the
// tile's own loads so their latency is not a chain of its own.
struct SeedRegs {
    u32 rec, q;      // reconstruction / residual of this lane's lattice point (0 outside the image)
    int bx, by;      // lattice coordinates inside the tile's halo frame
    bool on;
};

template <bool WANT_Q>
__device__ __forceinline__ SeedRegs seed_issue(const Seeds &sd, Tile tl, u32 k)
{
    const int ext = k >= 2 ? 2 : 1;   // offset 2^k beyond the tile is only ever read for k >= 2
    const int nbx = (TW >> k) + ext, nby = (TH >> k) + ext;
    const int i = HGI_LANE;
    SeedRegs r;
    r.by = i / nbx;
    r.bx = i - r.by * nbx;
    r.on = i < nbx * nby;
    r.rec = r.q = 0;
    const u32 sx = (tl.X0 >> k) + r.bx, sy = (tl.Y0 >> k) + r.by;
    if (r.on && sx < sd.sw && sy < sd.sh) {
        const size_t at = (size_t)tl.frame * sd.stride + (size_t)sy * sd.sw + sx;
        r.rec = sd.rec[at];
        if (WANT_Q) r.q = sd.q[at];
    }
    return r;
}

__device__ __forceinline__ void dec_seed_commit(u8 *buf, const SeedRegs &r, u32 k)
{
    if (r.on) buf[laddr(r.bx << k, r.by << k)] = (u8)r.rec;
    LDS_ORDER();
}

// ---- seeds rebuilt in the tile kernel, general form: the cone above a k = 4 tile (encode AND decode) ---------------
// A pyramid run with four fused levels has `up` = 1 ... 4 levels above the tile that the tile rebuilds (and, beyond eight
// levels, seed planes of the stride-256 lattice under those: Seeds in hgi_kernels.h); their pixels are the
// stride-16 lattice, a sw x sh plane with a pyramid of its own, and the tile needs reconstruction (encode: and residual)
// of (TW / 16 + 2) x (TH / 16 + 2) of its points -- seed_issue()'s layout.  What those depend on is a CONE: a point that
// is new at plane step 2s takes its corners from the multiples of 2s around it (src/interpolator.rs:57-91), so level by
// level the set of points halves in density and grows by at most one step at its far side (and down to the aligned
// origin at the near side).  In plane coordinates, with the tile's origin (A, B) = (X0 / 16, Y0 / 16) and s = 2^t:
//     box_t = [A & ~(s - 1), ... + (nx_t - 1) s] x [B & ~(s - 1), ...],    nx = 10, 6, 4, 3, 3;  ny = 6, 4, 3, 3, 3 (TH = 64)
// -- 60 points of the tile's own (level 0) and 24 + 12 + 9 + 9 = 54 above them.  Level 0 lies in the tile's halo frame: its
// input bytes are staged anyway.  The 54 others get ONE lane each (lane - cone_off(t) is the point's index in level t),
// which fetches the point's byte (source pixel when encoding, grid byte when decoding) with a single load instruction
// behind the tile's staging loads -- or takes it from the staged frame when the point lies in it -- and prepares its
// indices while the loads fly.  After staging the wave walks the levels from the base down: the lanes of a level read
// their four corners of the coarser level from a small byte array in LDS, predict, code (src/encoder.rs:46-65) or add
// the residual (src/decoder.rs:32-41), and write their point to the array; last the 60 seeds, which stay in registers.
// Neighbouring tiles recompute the same points -- pure functions of the input -- so nothing is exchanged and no launch
// runs in front of the tile kernel (a 16384^2 level-8 encode had a 9 us plane launch there).  Points outside the plane
// are 0, which is the out-of-image rule of the corners.
constexpr int kConeMaxUp = 4;
__host__ __device__ constexpr int cone_n(int tile_px, int t)      // points per dimension at stride 2^t (worst case over tile positions)
{
    const int c = tile_px >> 4;      // the tile origin is a multiple of c plane points
    int n = c + 2;
    for (int i = 0; i < t; ++i) n = (c % (2 << i) == 0 ? (n - 1) / 2 : n / 2) + 2;      // origin / 2^i even, or maybe odd
    return n;
}
__host__ __device__ constexpr int cone_off(int t)      // first entry (= first lane) of level t, t = 1 .. kConeMaxUp + 1 (level 0 stays in registers)
{
    int o = 0;
    for (int i = 1; i < t; ++i) o += cone_n(TW, i) * cone_n(TH, i);
    return o;
}
// The arrays -- one byte of reconstruction per point and, encoding, one byte of residual -- live in the halo-column slots
// of offsets 32 and 64, which a four-level tile never touches after staging (buf: reconstruction, rbuf: residuals): LDS is
// allocated in 1280-byte granules and the encoder's 7 648 bytes leave 32 to spare -- an array of its own costs three of 21 tiles per CU.
static_assert(cone_n(TW, 0) * cone_n(TH, 0) <= NL && cone_off(kConeMaxUp + 1) <= NL, "one lane per cone point");
static_assert(HR == 6 && cone_off(kConeMaxUp + 1) <= 2 * HP && cone_off(kConeMaxUp + 1) <= 2 * HP2, "the cone's arrays fit the two unused halo-column slots");
__device__ __forceinline__ u8 *cone_rec_array(u8 *buf) { return buf + HCOL + 4 * HP; }
__device__ __forceinline__ u8 *cone_q_array(u8 *rbuf) { return rbuf + RCOL + 4 * HP2; }
static_assert(cone_n(TW, 0) == 10 && cone_n(TW, 1) == 6 && cone_n(TW, 2) == 4 && cone_n(TW, 3) == 3 && cone_n(TW, 4) == 3, "cone widths of a 128-pixel tile");

struct ConeLane {
    u32 v, vq;        // levels >= 1: the lane's input byte (loaded; 0 outside the plane; read from the staged frame at the walk when
                      // framed1).  A base point taken from seed planes: v = reconstruction, vq = residual
    u32 t;            // its level (0: the lane has no point above level 0)
    u32 src1;         // LDS offset in the staged frame of a level >= 1 point that lies in it (framed1)
    bool framed1;
    u32 from1, nx1;   // index of its first corner in the arrays, and the pitch of that (coarser) level's box
    bool in1, down1;  // inside the plane / a point of the coarser lattice (handed down)
    u32 src0, from0;  // the same for the lane's seed (level 0, lanes < 10 x ny0)
    bool on0, in0, down0;
};

// the corner index of plane point (x, y) of level T in level T + 1's array, whose pitch is nx2
__device__ __forceinline__ u32 cone_corner(u32 x, u32 y, u32 A, u32 B, u32 T, u32 nx2)
{
    const u32 m2 = (2u << T) - 1u;
    return (((y & ~m2) - (B & ~m2)) >> (T + 1)) * nx2 + (((x & ~m2) - (A & ~m2)) >> (T + 1));
}

// Index work and the one load; nothing here waits.  fr: the frame (source or grid).  Pyramids deeper than 4 + sd.up levels: the
// base points come from the seed planes (stride-2^(4 + up) lattice, coded by earlier launches) instead of the frame.
template <bool WANT_Q>
__device__ __forceinline__ ConeLane cone_issue(const u8 *__restrict__ fr, u32 W, u32 H, const Seeds &sd, Tile tl)
{
    const u32 up = sd.up, sw = ((W - 1u) >> 4) + 1u, sh = ((H - 1u) >> 4) + 1u;      // the stride-16 plane
    constexpr u32 o2 = cone_off(2), o3 = cone_off(3), o4 = cone_off(4), o5 = cone_off(5);
    constexpr u32 nx0 = cone_n(TW, 0), ny0 = cone_n(TH, 0);
    const u32 lane = HGI_LANE, A = tl.X0 >> 4, B = tl.Y0 >> 4;
    ConeLane c;
    // levels 1 .. 4
    const u32 t = 1u + (lane >= o2) + (lane >= o3) + (lane >= o4);
    const u32 first = t == 1 ? 0u : t == 2 ? o2 : t == 3 ? o3 : o4;
    const u32 nx = t == 1 ? (u32)cone_n(TW, 1) : t == 2 ? (u32)cone_n(TW, 2) : t == 3 ? (u32)cone_n(TW, 3) : (u32)cone_n(TW, 4);
    const u32 rcp = t == 1 ? 43u : t == 2 ? 64u : 86u;      // (i * rcp) >> 8 == i / nx for nx = 6, 4, 3 and i < 64
    static_assert(cone_n(TW, 1) == 6 && cone_n(TW, 2) == 4 && cone_n(TW, 3) == 3 && cone_n(TW, 4) == 3, "reciprocals above");
    const u32 i = lane - first, iy = (i * rcp) >> 8, ix = i - iy * nx, m = (1u << t) - 1u;
    const u32 x = (A & ~m) + (ix << t), y = (B & ~m) + (iy << t);
    const bool on = lane < o5 && t <= up;
    c.t = on ? t : 0u;
    c.in1 = on && x < sw && y < sh;
    c.down1 = !((x | y) & (1u << t));
    c.nx1 = t == 1 ? (u32)cone_n(TW, 2) : (u32)cone_n(TW, 3);      // (level 4 has no coarser array)
    c.from1 = (t == 1 ? o2 : t == 2 ? o3 : o4) + cone_corner(x, y, A, B, t, c.nx1);
    const bool framed = x >= A && x < A + nx0 && y >= B && y < B + ny0;
    c.framed1 = c.in1 && framed;
    c.src1 = c.framed1 ? (u32)laddr((int)((x - A) << 4), (int)((y - B) << 4)) : 0u;
    c.v = c.vq = 0u;
    if (t == up && sd.rec) {      // (uniform in sd.rec) the base, coded earlier: plane point (x >> up, y >> up)
        c.framed1 = false;
        if (c.in1) {
            const size_t at = (size_t)tl.frame * sd.stride + (size_t)(y >> up) * sd.sw + (x >> up);
            c.v = sd.rec[at];
            if (WANT_Q) c.vq = sd.q[at];
        }
    } else if (c.in1 && !framed) {
        c.v = fr[((size_t)y << 4) * W + ((size_t)x << 4)];
    }
    if (!WANT_Q || !sd.rec) c.vq = c.v;      // base samples of the frame itself: residual == sample
    // level 0: the seeds
    const u32 by = (lane * 26u) >> 8, bx = lane - by * nx0;      // lane / 10
    static_assert(nx0 == 10, "reciprocal above");
    c.on0 = lane < nx0 * ny0;
    c.in0 = c.on0 && A + bx < sw && B + by < sh;
    c.down0 = !(((A + bx) | (B + by)) & 1u);      // (B is odd for every other row of 16-row tiles)
    c.from0 = cone_corner(A + bx, B + by, A, B, 0u, (u32)cone_n(TW, 1));
    c.src0 = c.on0 ? (u32)laddr((int)(bx << 4), (int)(by << 4)) : 0u;
    return c;
}

template <int INTERP, bool ENC, bool IDENT>
__device__ __forceinline__ void cone_code(u32 v, const u8 *r2, const u8 *q2, u32 nx2, bool down, bool inside, const u8 *slut, u32 &rec, u32 &q)
{
    const u32 c00 = r2[0], c01 = r2[nx2], c10 = r2[1], c11 = r2[nx2 + 1];      // (x0,y0) (x0,y1) (x1,y0) (x1,y1)
    const u32 q00 = ENC ? (u32)q2[0] : 0u;
    const u32 p = pred1<INTERP>(c00, c01, c10, c11);
    q = ENC ? quant1<IDENT>(v, p, slut) : v;
    rec = (p + q) & 255u;
    if (down) {      // a point of the coarser lattice: handed down
        rec = c00;
        q = q00;
    }
    if (!inside) rec = q = 0u;
}

// After staging (and, encoding, after the table is in LDS).  buf / rbuf: the staged planes (rbuf: encode only).
template <int INTERP, bool ENC, bool IDENT>
__device__ __forceinline__ SeedRegs cone_finish(const ConeLane &c, u8 *buf, u8 *rbuf, u32 up, const u8 *slut)
{
    const u32 lane = HGI_LANE;
    u8 *R = cone_rec_array(buf), *Q = ENC ? cone_q_array(rbuf) : nullptr;
    const u32 v1 = c.framed1 ? (u32)buf[(int)c.src1] : c.v;
    const u32 v0 = c.on0 ? (u32)buf[(int)c.src0] : 0u;
    if (c.t == up) {      // base samples travel as they are (src/encoder.rs:26-37, src/decoder.rs:22-28); 0 outside
        R[lane] = (u8)v1;
        if (ENC) Q[lane] = (u8)(c.framed1 ? v1 : c.vq);
    }
    LDS_ORDER();
    for (u32 T = up - 1u; T >= 1u; --T) {         // (uniform)
        if (c.t == T) {
            u32 rec, q;
            cone_code<INTERP, ENC, IDENT>(v1, R + c.from1, ENC ? Q + c.from1 : nullptr, c.nx1, c.down1, c.in1, slut, rec, q);
            R[lane] = (u8)rec;
            if (ENC) Q[lane] = (u8)q;
        }
        LDS_ORDER();
    }
    SeedRegs r;
    r.by = (int)((lane * 26u) >> 8);
    r.bx = (int)lane - r.by * cone_n(TW, 0);
    r.on = c.on0;
    r.rec = r.q = 0u;
    if (c.on0) cone_code<INTERP, ENC, IDENT>(v0, R + c.from0, ENC ? Q + c.from0 : nullptr, (u32)cone_n(TW, 1), c.down0, c.in0, slut, r.rec, r.q);
    LDS_ORDER();
    return r;
}

// One tile of the fast path, out of LDS: levels sub = 2^(k-1) .. 2 in place, then the finest level to HBM.
template <int INTERP>
__device__ __forceinline__ void dec_tile_fast(u8 *buf, const TileCtx &cur, const v4u (&odd)[NFINE], u32 k, u32 W, u32 H)
{
    // straight-line chain, the level a compile-time constant in each link (see enc_tile_fast)
#define HGI_DEC_COARSE(SUB)                                                                    \
    if (k > HGI_LOG2(SUB)) {                                                                   \
        HGI_MARK("coarse");                                                                    \
        dec_cells<INTERP, false>(buf, SUB, cur.tl, W, H);                                      \
        HGI_MARK("halo");                                                                      \
        dec_halo_cells<INTERP>(buf, SUB, cur.tl, W, H);                                        \
        LDS_ORDER();                                                                           \
    }
    if (MAXK >= 6) HGI_DEC_COARSE(32)
    if (MAXK >= 5) HGI_DEC_COARSE(16)
    HGI_DEC_COARSE(8)
    HGI_DEC_COARSE(4)
#undef HGI_DEC_COARSE
    if (k >= 2) {
        HGI_MARK("level2");
        dec_level2_fast<INTERP>(buf);
        HGI_MARK("halo");
        dec_halo_cells<INTERP>(buf, 2, cur.tl, W, H);
        LDS_ORDER();
    }
    HGI_MARK("fine");
    dec_fine_fast<INTERP>(buf, cur.b, odd);
}

// A ragged tile (any width, any height, any alignment).  Staging zeroes what lies outside the image (range check
// below it, lane and byte masks right of it), so the tile keeps the buffer staging and the fast finest level; the
// level passes in between only have to leave the out-of-image lattice points untouched, and the finest level
// stores just the part inside.
// EDGE == 1 is the lean form for the commonest class -- full width inside, even height (1080 / 720 / 2160 rows under
// widths that are multiples of 128): no column logic at all, and the range check alone disposes of the stores below
// the image.
template <int INTERP, int EDGE>
__device__ __forceinline__ void dec_tile_edge(u8 *buf, const TileCtx &cur, const v4u (&odd)[NFINE], u32 k, u32 W, u32 H)
{
    const int rows = (int)(H - cur.tl.Y0), cols = (int)(W - cur.tl.X0);
    // the levels as a straight-line chain with the level a compile-time constant in each link, like the interior path: a
    // lone frame's launch ends when its ragged tiles do, and their address arithmetic with a run-time step was the
    // longest chain in it (1920 x 1080: tools/size_sweep.py)
#define HGI_DEC_EDGE_COARSE(SUB)                                   \
    if (k > HGI_LOG2(SUB)) {                                       \
        dec_cells<INTERP, true>(buf, SUB, cur.tl, W, H);           \
        dec_halo_cells<INTERP>(buf, SUB, cur.tl, W, H);            \
        LDS_ORDER();                                               \
    }
    if (MAXK >= 6) HGI_DEC_EDGE_COARSE(32)
    if (MAXK >= 5) HGI_DEC_EDGE_COARSE(16)
    HGI_DEC_EDGE_COARSE(8)
    HGI_DEC_EDGE_COARSE(4)
#undef HGI_DEC_EDGE_COARSE
    if (k >= 2) {
        dec_level2_fast<INTERP, EDGE>(buf, rows, cols);
        dec_halo_cells<INTERP>(buf, 2, cur.tl, W, H);
        LDS_ORDER();
    }
    dec_fine_fast<INTERP, EDGE>(buf, cur.b, odd, rows, cols);
}

// One block (= one wave) per tile, ONE launch per batch.  The first blocks take the ragged tiles
// (their count padded to a multiple of 8 so that b % 8 keeps labelling the XCD), so the slow tiles
// start first and overlap the interior ones; the interior tiles follow in XCD-contiguous order.
// (A persistent variant -- resident waves pulling tiles from per-XCD atomic counters and prefetching
// the next tile into registers -- was built and measured: not faster on MI355X, see DESIGN.md
// "Scheduling".)
struct BlockRole {
    bool edge, idle;
    u32 index;       // edge tile index, or position of the interior tile in the XCD-contiguous order
};

__device__ __forceinline__ BlockRole block_role(const TileGrid &g)
{
    const u32 b = blockIdx.x, ne8 = (g.nedge + 7u) & ~7u;
    BlockRole r;
    r.edge = b < ne8;
    r.idle = r.edge && b >= g.nedge;
    const u32 fb = b - ne8;
    r.index = r.edge ? b : range_first(g.nf, fb & 7u) + (fb >> 3);
    // Which tiles the eight XCDs work on at one time (speed only; g.xmode, host policy xcd_mode()).  0: each XCD walks its
    // own contiguous eighth of the band-ordered list -- eight places an eighth of the batch apart, a power-of-two distance
    // on power-of-two frames.  1: whole bands dealt round-robin, so the XCDs work on eight CONSECUTIVE bands (16384^2:
    // -10 % encode, -9 % decode; 64 x 4096^2: -1.5 ... -1.8 %; profiles/r03_ab_xcd.txt); what is left after the last
    // multiple of eight bands is split contiguously as in mode 0.  (finish_grid() clears xmode when a frame's rows do not
    // divide into whole bands.)
    if (!r.edge && g.xmode == 1) {
        const u32 x = fb & 7u, sq = fb >> 3;
        if (sq < g.rr_own) {
            const u32 round = fdiv(sq, g.fd_P);
            r.index = (round * 8u + x) * g.P + (sq - round * g.P);
        } else {
            r.index = g.rr_tail0 + range_first(g.nf - g.rr_tail0, x) + (sq - g.rr_own);
        }
    }
    return r;
}

#ifndef HGI_DEC_WAVES_PER_EU
#define HGI_DEC_WAVES_PER_EU 8
#endif
// SEEDED: 0 = the pyramid fits the tile; 1 = seeds from planes coded by earlier launches (sd.up == 0); 2 = seeds rebuilt in
// the kernel (cone_*: k == 4, sd.up levels above the tile).  (That one is allowed 80 registers -- six waves per SIMD, 24 tiles
// per CU: the cone's lane state is live while the staging loads fly, and the decoder's rate does not depend on occupancy
// down to 16 tiles per CU, profiles/r03_waves_sweep.txt.)
template <int INTERP, int SEEDED, int TILE_ROWS>   // TILE_ROWS == TH: only there to name the build in profiles
__global__ __launch_bounds__(NL) __attribute__((amdgpu_waves_per_eu(SEEDED == 2 ? HGI_DEC_WAVES_PER_EU - 2 : HGI_DEC_WAVES_PER_EU))) void k_dec_tiles(const u8 *__restrict__ src, u8 *__restrict__ dst, Frames f, u32 k,
                                                  Seeds sd, TileGrid g, u32 aligned)
{
    HGI_TL_ENTRY();
    args_early(src, dst, f, g, aligned, k);
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
#ifdef HGI_ANALYZE_K
    k = HGI_ANALYZE_K;
#endif
    const int nh = k >= 2 ? (int)k : 1;
    u8 *buf = smem - HCOL;
    const u32 W = f.width, H = f.height;
#ifdef HGI_ANALYZE_K
    BlockRole role = block_role(g);
    role.edge = role.idle = false;
#else
    const BlockRole role = block_role(g);
#endif
    if (role.idle) return;
    if (!role.edge) {
        TileCtx cur = fast_ctx(role.index, src, dst, f, g, (aligned & 4u) ? 3u : 0u);
        Stage st;
        SeedRegs seeds;
        HGI_MARK("stage_issue");
        HGI_TL_START();
        ConeLane cone;
#if HGI_DEC_CONE_FIRST
        if (SEEDED == 2) cone = cone_issue<false>(src + (size_t)cur.tl.frame * f.frame_stride, W, H, sd, cur.tl);
#endif
        stage_issue<false>(st, cur.b, cur.tl, (int)k, nh);
        if (SEEDED == 1) seeds = seed_issue<false>(sd, cur.tl, k);
#if !HGI_DEC_CONE_FIRST
        if (SEEDED == 2) cone = cone_issue<false>(src + (size_t)cur.tl.frame * f.frame_stride, W, H, sd, cur.tl);
#endif
        HGI_MARK("stage_commit");
        stage_commit<false>(buf, nullptr, st, nh);
        LDS_ORDER();
        HGI_TL_STAGED();      // (timeline build: waits for everything requested so far -- in front of the odd rows' request)
        if (HGI_ODD_LATE) stage_issue_odd(st, cur.b);
        if (SEEDED == 2) seeds = cone_finish<INTERP, false, true>(cone, buf, nullptr, sd.up, nullptr);
        LDS_ORDER();
        if (SEEDED) dec_seed_commit(buf, seeds, k);
        dec_tile_fast<INTERP>(buf, cur, st.o, k, W, H);
        HGI_MARK("end");
        HGI_TL_END();
        return;
    }
    // ragged tile (body crosses the image edge), unaligned rows, or offsets beyond 32 bits: every access checked
    const Tile tl = edge_tile(role.index, g);
    const u8 *fr = src + (size_t)tl.frame * f.frame_stride;
    u8 *out = dst + (size_t)tl.frame * f.frame_stride;
    SeedRegs seeds;
    ConeLane cone;
    if (SEEDED == 1) seeds = seed_issue<false>(sd, tl, k);
    if (SEEDED == 2) cone = cone_issue<false>(fr, W, H, sd, tl);
    if (aligned & 2u) {   // 32-bit buffer offsets: buffer staging and the check-free levels with edge masks
        TileCtx cur = {tl, make_buf(fr, out, W, H, tl, (aligned & 4u) ? 3u : 0u)};
        Stage st;
        stage_issue<true, true>(st, cur.b, tl, (int)k, nh);
        stage_commit<false>(buf, nullptr, st, nh);
        LDS_ORDER();
        if (SEEDED == 2) seeds = cone_finish<INTERP, false, true>(cone, buf, nullptr, sd.up, nullptr);
        if (SEEDED) dec_seed_commit(buf, seeds, k);
        if (tl.X0 + TW <= W && !(H & 1u))
            dec_tile_edge<INTERP, 1>(buf, cur, st.o, k, W, H);
        else
            dec_tile_edge<INTERP, 2>(buf, cur, st.o, k, W, H);
        return;
    }
    // frames whose byte offsets do not fit 32 bits: every access checked, 64-bit addressing
    stage_tile_generic(buf, fr, W, H, tl, nh, (aligned & 1u) != 0);
    LDS_ORDER();
    if (SEEDED == 2) seeds = cone_finish<INTERP, false, true>(cone, buf, nullptr, sd.up, nullptr);
    if (SEEDED) dec_seed_commit(buf, seeds, k);
    for (int s = 1 << (k - 1); s >= 2; s >>= 1) {
        dec_cells<INTERP, true>(buf, s, tl, W, H);
        dec_halo_cells<INTERP>(buf, s, tl, W, H);
        LDS_ORDER();
    }
    dec_fine_generic<INTERP>(buf, fr, out, tl, W, H, (aligned & 1u) != 0);
}

// ---- encode ---------------------------------------------------------------------------------------
// lattice points = 0 (mod 2^k): reconstruction == original (src/encoder.rs:26-37) -- which is what staging
// left in rbuf -- or, when this launch is the lower part of a deeper pyramid, the coarser pyramid's
// reconstruction, with its residuals taking the place of the originals in the output.
template <int SEEDED>
__device__ __forceinline__ void enc_seed_commit(u8 *buf, u8 *rbuf, const SeedRegs &r, u32 k)
{
    if (SEEDED && r.on) {
        buf[laddr(r.bx << k, r.by << k)] = (u8)r.q;
        rbuf[laddr2(r.bx << k, r.by << k)] = (u8)r.rec;
    }
    LDS_ORDER();
}

template <int INTERP, bool IDENT, int EDGE>
__device__ __forceinline__ void enc_tile_edge(u8 *buf, u8 *rbuf, const u8 *slut, const TileCtx &cur,
                                              const v4u (&odd)[NFINE], u32 k, u32 W, u32 H)
{
#define HGI_ENC_EDGE_COARSE(SUB)                                                   \
    if (k > HGI_LOG2(SUB)) {                                                       \
        enc_cells<INTERP, IDENT, true>(buf, rbuf, slut, SUB, cur.tl, W, H);        \
        enc_halo_pass<INTERP, IDENT>(buf, rbuf, slut, SUB, cur.tl, W, H);          \
        LDS_ORDER();                                                               \
    }
    if (MAXK >= 6) HGI_ENC_EDGE_COARSE(32)
    if (MAXK >= 5) HGI_ENC_EDGE_COARSE(16)
    HGI_ENC_EDGE_COARSE(8)
    HGI_ENC_EDGE_COARSE(4)
#undef HGI_ENC_EDGE_COARSE
    if (k >= 2) {
        enc_level2_fast<INTERP, IDENT, EDGE>(buf, rbuf, slut, cur.tl, W, H);
        LDS_ORDER();
    }
    enc_fine_fast<INTERP, IDENT, EDGE>(buf, rbuf, slut, cur.b, odd, (int)(H - cur.tl.Y0), (int)(W - cur.tl.X0));
}

template <int INTERP, bool IDENT>
__device__ __forceinline__ void enc_tile_fast(u8 *buf, u8 *rbuf, const u8 *slut, const TileCtx &cur,
                                              const v4u (&odd)[NFINE], u32 k, u32 W, u32 H)
{
    // One straight-line chain with the level as a compile-time constant in each link (k only selects where the chain
    // is entered): every LDS address of a level is then `lane-dependent base + immediate`, instead of shifts by a
    // run-time log2(step) -- the address arithmetic was a third of the kernel's VALU instructions.
#define HGI_ENC_COARSE(SUB)                                                                    \
    if (k > HGI_LOG2(SUB)) {                                                                   \
        HGI_MARK("coarse");                                                                    \
        enc_level_coarse_fast<INTERP, IDENT>(buf, rbuf, slut, SUB, cur.tl, W, H);              \
        LDS_ORDER();                                                                           \
    }
    if (MAXK >= 6) HGI_ENC_COARSE(32)
    if (MAXK >= 5) HGI_ENC_COARSE(16)
    HGI_ENC_COARSE(8)
    HGI_ENC_COARSE(4)
#undef HGI_ENC_COARSE
    if (k >= 2) {
        HGI_MARK("level2");
        enc_level2_fast<INTERP, IDENT>(buf, rbuf, slut, cur.tl, W, H);
        LDS_ORDER();
    }
    HGI_MARK("fine");
    enc_fine_fast<INTERP, IDENT, 0>(buf, rbuf, slut, cur.b, odd);
}

// Occupancy targets handed to the register allocator: encode fits 96 VGPRs (5 waves per SIMD, 20 per CU;
// LDS allows 21 at k = 4) without spilling -- tests/test_isa.py checks that -- decode needs ~50.
#ifndef HGI_ENC_WAVES_PER_EU
#define HGI_ENC_WAVES_PER_EU 5
#endif
// SEEDED: as in k_dec_tiles
template <int INTERP, bool IDENT, int SEEDED, int TILE_ROWS>
__global__ __launch_bounds__(NL) __attribute__((amdgpu_waves_per_eu(IDENT ? HGI_ENC_WAVES_PER_EU - 1 : HGI_ENC_WAVES_PER_EU))) void k_enc_tiles(const u8 *__restrict__ src, u8 *__restrict__ dst, Frames f, u32 k,
                                                  Lut256 lut, Seeds sd, TileGrid g, u32 aligned)
{
    HGI_TL_ENTRY();
    args_early(src, dst, f, g, aligned, k);
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
#ifdef HGI_ANALYZE_K
    k = HGI_ANALYZE_K;
#endif
    const int nh = k >= 2 ? (int)k : 1;
    // the table first, at LDS offset 0: a residual byte then IS the address of its table entry (lut_at(); the
    // planes behind it have a size that depends on k).  Dynamic LDS starts at 0 because the kernel has no static LDS:
    // launch_encode_fused verifies that on the host (hipFuncGetAttributes) before the first launch of each
    // instantiation and refuses to launch otherwise -- nothing on the device can abort.
    u8 *slut = smem;
    u8 *buf = smem + 256 - HCOL;
    u8 *rbuf = smem + 256 + buf_bytes(nh) - RCOL;
    const u32 W = f.width, H = f.height;
#ifdef HGI_ANALYZE_K
    BlockRole role = block_role(g);
    role.edge = role.idle = false;
#else
    const BlockRole role = block_role(g);
#endif
    if (role.idle) return;
    // The table entry of this lane is a VECTOR load from the kernel-argument segment (the index is the lane): a trip to L2
    // or beyond.  Done first, the whole prologue stalled on it (the timeline build showed 1.4-1.8 us of prologue per
    // encode tile against 0.5-0.7 for decode: tools/timeline.py).  It is issued BEHIND the tile's staging loads instead
    // and lands with them -- the table is first needed after staging has been committed to LDS anyway.
    u32 lutv = 0;
    if (!role.edge) {
        TileCtx cur = fast_ctx(role.index, src, dst, f, g, (aligned & 4u) ? 3u : 0u);
        Stage st;
        SeedRegs seeds;
        ConeLane cone;
        HGI_MARK("stage_issue");
        HGI_TL_START();
        stage_issue<false>(st, cur.b, cur.tl, (int)k, nh);
        if (SEEDED == 1) seeds = seed_issue<true>(sd, cur.tl, k);
        if (SEEDED == 2) cone = cone_issue<true>(src + (size_t)cur.tl.frame * f.frame_stride, W, H, sd, cur.tl);
        if (!IDENT) lutv = lut.w[HGI_LANE];
        HGI_MARK("stage_commit");
        stage_commit<true>(buf, rbuf, st, nh);
        if (!IDENT) reinterpret_cast<u32 *>(slut)[HGI_LANE] = lutv;
        LDS_ORDER();
        HGI_TL_STAGED();
        if (HGI_ODD_LATE) stage_issue_odd(st, cur.b);
        if (SEEDED == 2) seeds = cone_finish<INTERP, true, IDENT>(cone, buf, rbuf, sd.up, slut);
        enc_seed_commit<SEEDED>(buf, rbuf, seeds, k);
        enc_tile_fast<INTERP, IDENT>(buf, rbuf, slut, cur, st.o, k, W, H);
        HGI_MARK("end");
        HGI_TL_END();
        return;
    }
    const Tile tl = edge_tile(role.index, g);
    const u8 *fr = src + (size_t)tl.frame * f.frame_stride;
    u8 *out = dst + (size_t)tl.frame * f.frame_stride;
    SeedRegs seeds;
    if (SEEDED == 1) seeds = seed_issue<true>(sd, tl, k);
    ConeLane cone;
    if (SEEDED == 2) cone = cone_issue<true>(fr, W, H, sd, tl);
    if (aligned & 2u) {   // 32-bit buffer offsets: buffer staging and the check-free levels with edge masks
        TileCtx cur = {tl, make_buf(fr, out, W, H, tl, (aligned & 4u) ? 3u : 0u)};
        Stage st;
        stage_issue<true, true>(st, cur.b, tl, (int)k, nh);
        if (!IDENT) lutv = lut.w[HGI_LANE];
        stage_commit<true>(buf, rbuf, st, nh);
        if (!IDENT) reinterpret_cast<u32 *>(slut)[HGI_LANE] = lutv;
        LDS_ORDER();
        if (SEEDED == 2) seeds = cone_finish<INTERP, true, IDENT>(cone, buf, rbuf, sd.up, slut);
        enc_seed_commit<SEEDED>(buf, rbuf, seeds, k);
        if (tl.X0 + TW <= W && !(H & 1u))
            enc_tile_edge<INTERP, IDENT, 1>(buf, rbuf, slut, cur, st.o, k, W, H);
        else
            enc_tile_edge<INTERP, IDENT, 2>(buf, rbuf, slut, cur, st.o, k, W, H);
        return;
    }
    if (!IDENT) reinterpret_cast<u32 *>(slut)[HGI_LANE] = lut.w[HGI_LANE];
    stage_tile_generic(buf, fr, W, H, tl, nh, (aligned & 1u) != 0);
    LDS_ORDER();
    lattice_from_buf(buf, rbuf, nh);
    LDS_ORDER();
    if (SEEDED == 2) seeds = cone_finish<INTERP, true, IDENT>(cone, buf, rbuf, sd.up, slut);
    enc_seed_commit<SEEDED>(buf, rbuf, seeds, k);
    for (int s = 1 << (k - 1); s >= 2; s >>= 1) {
        enc_cells<INTERP, IDENT, true>(buf, rbuf, slut, s, tl, W, H);
        enc_halo_pass<INTERP, IDENT>(buf, rbuf, slut, s, tl, W, H);
        LDS_ORDER();
    }
    enc_fine_generic<INTERP, IDENT>(buf, rbuf, slut, fr, out, tl, W, H, (aligned & 1u) != 0);
}

inline bool ptr16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

struct FusedGeom {
    TileGrid g;
    u32 aligned;
    bool ok;
};

// row_limit (pixel rows, a multiple of 64; 0 = none): only the tile rows above it are launched.  The frame keeps its
// true height, so the out-of-image rule is untouched -- this is how a band of a frame is coded while the rows further
// down are still being uploaded (hgi_capi.hip, host-pointer calls).
FusedGeom fused_geom(const void *a, const void *b, const Frames &f, u32 row_limit)
{
    FusedGeom r;
    TileGrid &g = r.g;
    g.tiles_x = (f.width + TW - 1) / TW;
    g.tiles_y = (f.height + TH - 1) / TH;
    if (row_limit && row_limit / TH < g.tiles_y) g.tiles_y = row_limit / TH;
    const bool aligned = f.width % 16 == 0 && f.frame_stride % 16 == 0 && ptr16(a) && ptr16(b);
    // every 32-bit buffer offset the fast path forms: (Y0 + TH + 64) * W + X0 + TW + 64 + 16
    const bool fits32 = ((u64)f.height + 2 * TH + 64) * f.width + 1024 < (1ull << 32);
    // Buffer loads / stores take any alignment (ROCm runs gfx9+ with unaligned access enabled; rows that start
    // mid-line only cost extra line touches), so the check-free paths do not need 16-B aligned rows.  With rows that
    // are a multiple of 4 bytes every dword of every load lines up with the end of the frame (num_records = W * H).
    // Otherwise the read descriptors get 3 extra records -- if the caller's memory allows it: the over-read of every
    // frame but the last lands in the next frame (or the stride padding), that of the last frame must stay in the
    // 4-KiB page that holds the frame's last byte.
    const bool force_checked = HGI_SWITCH(HGI_FORCE_CHECKED);   // knobs build (tests): every tile through the byte-checked path
    const uintptr_t end = reinterpret_cast<uintptr_t>(a) + (uintptr_t)(f.batch - 1) * f.frame_stride + (uintptr_t)f.width * f.height;
    const bool dword_rows = f.width % 4 == 0;
    const bool tail_ok = ((end - 1) >> 12) == ((end + 2) >> 12) && (f.batch == 1 || f.frame_stride >= (u64)f.width * f.height);
    const bool fast = fits32 && (dword_rows || tail_ok) && !force_checked;
    g.full_x = fast ? f.width / TW : 0;
    g.full_y = fast ? f.height / TH : 0;
    if (g.full_y > g.tiles_y) g.full_y = g.tiles_y;
    if (g.full_x == 0 || g.full_y == 0) g.full_x = g.full_y = 0;
    const u64 all = (u64)g.tiles_x * g.tiles_y * f.batch, nfast = (u64)g.full_x * g.full_y * f.batch;
    r.ok = all > 0 && all + 8 < (1ull << 31);
    g.nfast = (u32)nfast;
    g.nedge = (u32)(all - nfast);
    g.reverse = 0;
    g.band = HGI_TILE_BAND;
    g.xmode = 0;
    // bit 0: 16-B aligned rows and pointers (vector accesses of the byte-checked path); bit 1: check-free paths allowed
    // (32-bit buffer offsets); bit 2: read descriptors carry 3 extra records (rows not a multiple of 4 bytes)
    r.aligned = (aligned ? 1u : 0u) | (fast ? 2u : 0u) | (fast && !dword_rows ? 4u : 0u);
#ifdef HGI_TIMELINE
    g.timeline = g_timeline;
#endif
    return r;
}

// Tile rows per band of the interior walk.  A band is walked column-major, so the tiles in flight at one moment span
// `rows * TH` image rows of a few tile columns: rows * TH * W bytes of address space.  Measured with the bands dealt
// round-robin to the XCDs (tools/order_sweep.sh, profiles/r03_order_sweep.txt; tools/ab.py, profiles/r03_ab_c4_order.txt):
// four rows are best in both directions on 4096-wide frames (64 x 4096^2: 354.6 / 347.3 us against 357.2 / 351.4 at
// eight and 366.0 / 353.9 at two); on a 16384-wide frame the encoder still wants four (102 us; two: 109) and the decoder
// two (98 us; four: 104) -- and with round 2's eight-row decoder bands and contiguous eighths that decode took 135 us,
// which round 2 mistook for a read-after-write penalty.  Hence: four rows, capped at 4 MiB (encode) / 2 MiB (decode) of
// address space.  (Knobs build: HGI_ENC_BAND / HGI_DEC_BAND force a height -- tools/band_sweep.sh.)
inline u32 band_rows(const Frames &f, bool encode)
{
    const int forced = encode ? HGI_KNOB(HGI_ENC_BAND, 0) : HGI_KNOB(HGI_DEC_BAND, 0);
    if (forced > 0) return (u32)forced;
    const u64 cap = encode ? (4u << 20) : (2u << 20);
    u32 rows = HGI_TILE_BAND;
    while (rows > 1 && (u64)rows * TH * f.width > cap) rows >>= 1;
    return rows;
}

// Everything the kernels would otherwise derive per block from the launch's constants (and the divisions by them).
inline void finish_grid(TileGrid &g)
{
    g.ex = g.full_x;
    g.nf = g.nfast;
    if (g.band < 1) g.band = 1;
    g.tpf = g.ex * g.full_y;
    g.P = g.band * g.ex;
    g.nfull = (g.full_y / g.band) * g.P;
    g.rem_rows = g.full_y % g.band;
    g.fd_tpf = make_fastdiv(g.tpf);
    g.fd_P = make_fastdiv(g.P);
    g.fd_band = make_fastdiv(g.band);
    g.fd_rem = make_fastdiv(g.rem_rows);
    g.fd_ex = make_fastdiv(g.ex);
    g.rr_own = g.rr_tail0 = 0;
    if (g.xmode != 1 || g.rem_rows != 0 || g.P == 0 || g.nf == 0) {
        g.xmode = 0;
    } else {
        const u32 nb8 = (g.nf / g.P) & ~7u;          // bands in whole rounds of eight
        g.rr_own = (nb8 >> 3) * g.P;
        g.rr_tail0 = nb8 * g.P;
    }
}

// Dynamic LDS of a launch padded so that at most `waves` blocks fit a CU (160 KiB of LDS): a launch only a few rounds of
// resident tiles deep pays a tile LIFETIME for filling and draining the chip, and the lifetime is resident tiles / rate
// (Little) -- fewer resident tiles at the same rate shorten it.  (Knobs build: HGI_DEC_WAVES / HGI_ENC_WAVES, tools/waves_sweep.sh.)
inline size_t lds_for_waves(size_t lds, int waves)
{
    if (waves <= 0) return lds;
    const size_t cap = ((size_t)160 * 1024 / (size_t)waves) & ~(size_t)255;
    return cap > lds ? cap : lds;
}

// How the band-ordered tile list is dealt to the eight XCDs (block_role).  Whole bands round-robin -- the chip works on eight
// CONSECUTIVE bands -- is what a launch up to a few GiB wants: contiguous eighths put the XCDs' eight working points a
// power-of-two distance of 32 ... 256 MiB apart, and those streams beat against each other in the memory system (16384^2:
// +10 %; 64 and 128 x 4096^2: +2.5 ... +6 %; profiles/r03_ab_xcd.txt, r04_c3_xcd_sweep.txt).  Once the eighths are 768 MiB and
// more apart that is over, and for the ENCODER eight separate fronts are then faster than one: 384 ... 768 x 4096^2 -5 ... -6 %,
// on every box sampled at 512 frames -0.7 ... -6 % (2.65-2.79 against 2.81-2.82 ms; equal at 256 and at 1024 frames).  The
// DECODER is slower with eighths on every box -- at the ten resident tiles per CU it runs with +3 ... +5 % (2.77-2.80 against
// 2.66-2.69 ms; at 16, where it stood when the policy was made, +0.5 ... +2.4 %; profiles/r04_c3_xcd_boxes.txt) -- and keeps the
// round-robin dealing.
// Hence by direction and size: contiguous eighths for encodes from FOUR GiB of interior tiles per plane (kEncodeEighthsFromGiB; six
// until the planes' line-up spread the grid plane over two classes -- since then, against round-robin on 32-row tiles, the encoder
// of 256 / 320 x 4096^2: 1.374 -> 1.329 / 1.722 -> 1.661 ms, 80 x 8192^2 -1.1 %, 1280 x 2048^2 -1.9 %, 20 x 16384^2 -7 %; at
// 192 x 4096^2, three GiB, still level or behind; profiles/r04_mid_sizes.txt).  (Knobs build: HGI_XCD_MODE = 0 | 1 forces one
// for both directions.)
inline u32 xcd_mode(const TileGrid &g, bool encode)
{
    const int forced = HGI_KNOB(HGI_XCD_MODE, HGI_XCD_MODE);
    if (forced >= 0) return (u32)forced;
    return encode && (u64)g.nfast * TW * TH >= ((u64)HGI_XCD_EIGHTHS_FROM_GIB << 30) ? 0u : 1u;
}

}  // namespace

#ifdef HGI_FUSED_DECODE
hipError_t HGI_TILED(launch_decode_fused)(const uint8_t *grid, uint8_t *img, const Frames &f, uint32_t k, int interp,
                                          const Seeds *seeds, hipStream_t s, uint32_t row_limit, int resident_tiles)
{
    FusedGeom r = fused_geom(grid, img, f, row_limit);
    if (!r.ok || k < 1 || k > (u32)MAXK) return hipErrorInvalidValue;
    // The decoder walks the tile list FORWARDS, like the encoder.  Walking it backwards -- so that in an encode -> decode chain
    // the grid bytes the encoder wrote last, the ones still in the 256 MiB Infinity Cache, are read first -- was built and
    // measured on 16384^2 and on the 64 x 4096^2 shard: no difference (92.5-92.8 us either way, profiles/r03_ab_reverse.txt,
    // DESIGN.md 6.1), so it is not shipped; g.reverse stays for the knobs build (HGI_DEC_REVERSE=1).  Order never changes the bytes.
    r.g.reverse = HGI_KNOB(HGI_DEC_REVERSE, HGI_DEC_REVERSE_DEFAULT) ? 1u : 0u;
    r.g.band = band_rows(f, false);
    r.g.xmode = xcd_mode(r.g, false);
    finish_grid(r.g);
    const TileGrid &g = r.g;
    if (seeds && k < (uint32_t)kSeededMinLevels) return hipErrorInvalidValue;      // one lane per lattice point (seed_issue)
    Seeds sd = seeds ? *seeds : Seeds{nullptr, nullptr, 0, 0, 0, 0};
    // sd.up levels above the tile are rebuilt by the kernel (cone_*): four fused levels, one lane per cone point
    const bool cone = seeds && seeds->up != 0;
    if (cone && (k != 4 || sd.up > (u32)kConeMaxUp)) return hipErrorInvalidValue;
    if (seeds && !cone && !seeds->rec) return hipErrorInvalidValue;
    const dim3 b(NL);
    const int nh = k >= 2 ? (int)k : 1;
    // How many tiles a CU holds at one time (the dynamic LDS is padded to that: lds_for_waves).  Its LDS would allow 32.
    //  * Fewer than 8 192 tiles: all it can get -- the launch ends when its slowest wave does.
    //  * A PLAIN decode (the tile holds the pyramid: no seeds) of 8 192 tiles and more: TEN on rows up to 4 096 pixels and up to
    //    four levels, twelve otherwise.  Round 4, in-process A/B of builds on the same placed planes (tools/ab.py,
    //    profiles/r04_ab_dec_waves.txt): 64 x 4096^2 level 4 against the 16 of the round's first builds: 9 tiles -1.9 %, 10 -5.2 %,
    //    11 -3.1 %, 12 -0.8 %; 256 x -6.3 %, 512 x -5.5 % (2.789 -> 2.636 ms = 0.815 of 8 TB/s), 40 / 24 / 16 x -4.3 / -4.1 / -3.4 %,
    //    8 x -0.8 %; levels 1 / 2 -1.4 / -2.1 %; 300 x 1920 x 1080 -2.3 %.  On 8192- and 16384-wide rows and at five levels twelve
    //    is the better number (-2.8 / -2.3 ... -3.6 / -2.0 % against -2.4 / -0.6 ... -2.3 / -0.2 % at ten).  The likely reason:
    //    with ten tiles per CU the 320 tiles an XCD has in flight read 2.7 MB at a time, which -- halo lines included -- stays in
    //    its 4 MB L2; at 16 and more it does not (round 1 counted 1.06 x the algorithmic bytes fetched at 32); below ten the
    //    latency is no longer covered.  At ONE level (the finest pass alone, P_fine) eight: 64 x 4096^2 0.3360 -> 0.3277 ms, 512 x
    //    2.768 -> 2.692 (profiles/r04_ab_pfine.txt).
    //  * A decode that rebuilds levels above its tiles (the cone) or starts from seed planes: 20 on launches one to eight
    //    rounds of resident tiles deep (8 192 ... 65 536 tiles: a lone 16384^2 frame at level 8 has 32 768; the tile lifetime --
    //    what filling and draining the chip costs -- is shorter at the same rate: 101.3 us at 32, 99.1 at 20, 98.5 at 16, 109.8
    //    at 12, profiles/r03_waves_sweep.txt), 16 on deeper ones; these lose with ten or twelve (+3 ... +10 %: the cone's far
    //    loads and its walk want the occupancy).  (The encoder needs all 20 it can get: 98 us, 106 at 16.)
    const int dec_waves_forced = HGI_KNOB(HGI_DEC_WAVES, -1);
    const u64 tiles = (u64)g.nfast + g.nedge;
    int dec_waves = 0;
    if (resident_tiles >= 0)
        dec_waves = resident_tiles;      // the caller's choice (the placement probe, hgi_planes.hip: it has to keep measuring the
                                         // same thing whatever the policy or a knob says)
    else if (dec_waves_forced >= 0)
        dec_waves = dec_waves_forced;
    else if (tiles < 8192)
        dec_waves = 0;
    else if (!seeds)
        dec_waves = f.width > 4096 || k > 4 ? HGI_DEC_STREAM_WAVES_WIDE : k == 1 ? HGI_DEC_STREAM_WAVES_L1 : HGI_DEC_STREAM_WAVES;
    else
        dec_waves = tiles >= 65536 ? HGI_DEC_DEEP_WAVES : HGI_DEC_SHALLOW_WAVES;
    const size_t lds = lds_for_waves((size_t)buf_bytes(nh), dec_waves);
    const dim3 blocks(((g.nedge + 7u) & ~7u) + g.nfast);
#define HGI_DEC(I, SE) hipLaunchKernelGGL((k_dec_tiles<I, SE, TH>), blocks, b, lds, s, grid, img, f, k, sd, g, r.aligned)
    // (seed PLANES without a cone -- SEEDED == 1 -- exist for 64-row tiles only: six fused levels, hgi_capi.hip host_banded)
#define HGI_DEC_I(I)                                                     \
    do {                                                                 \
        if (!seeds) HGI_DEC(I, 0);                                       \
        else if (cone) HGI_DEC(I, 2);                                    \
        else if constexpr (TH == 64) HGI_DEC(I, 1);                      \
        else return hipErrorInvalidValue;                                \
    } while (0)
    if (interp == kInterpCrossed) HGI_DEC_I(kInterpCrossed); else HGI_DEC_I(kInterpLeftTop);
#undef HGI_DEC_I
#undef HGI_DEC
    return hipGetLastError();
}

#endif  // HGI_FUSED_DECODE

#ifdef HGI_FUSED_ENCODE
namespace {
hipError_t static_lds_is_empty(const void *kernel)
{
    hipFuncAttributes a;
    const hipError_t e = hipFuncGetAttributes(&a, kernel);
    if (e != hipSuccess) return e;
    return a.sharedSizeBytes == 0 ? hipSuccess : hipErrorInvalidDeviceFunction;
}
}  // namespace

hipError_t HGI_TILED(launch_encode_fused)(const uint8_t *img, uint8_t *grid, const Frames &f, uint32_t k, int interp,
                                          const Lut256 &lut, bool ident, const Seeds *seeds, hipStream_t s, uint32_t row_limit)
{
    FusedGeom r = fused_geom(img, grid, f, row_limit);
    if (!r.ok || k < 1 || k > (u32)MAXK) return hipErrorInvalidValue;
    r.g.band = band_rows(f, true);
    r.g.xmode = xcd_mode(r.g, true);
    finish_grid(r.g);
    const TileGrid &g = r.g;
    if (seeds && k < (uint32_t)kSeededMinLevels) return hipErrorInvalidValue;      // one lane per lattice point (seed_issue)
    Seeds sd = seeds ? *seeds : Seeds{nullptr, nullptr, 0, 0, 0, 0};
    // sd.up levels above the tile are rebuilt by the kernel (cone_*): four fused levels, one lane per cone point
    const bool cone = seeds && seeds->up != 0;
    if (cone && (k != 4 || sd.up > (u32)kConeMaxUp)) return hipErrorInvalidValue;
    if (seeds && (!seeds->rec != !seeds->q || (!cone && !seeds->rec))) return hipErrorInvalidValue;
    const dim3 b(NL);
    const int nh = k >= 2 ? (int)k : 1;
    const int enc_waves = HGI_KNOB(HGI_ENC_WAVES, 0);
    const size_t lds = lds_for_waves((size_t)buf_bytes(nh) + ((rbuf_bytes(nh) + 15) & ~15) + 256, enc_waves);
    const dim3 blocks(((g.nedge + 7u) & ~7u) + g.nfast);
    // lut_at() addresses the table from LDS offset 0: only valid while the kernel has no static LDS in front of its
    // dynamic segment.  Checked once per instantiation on the host; a build that breaks it fails here, not on the device.
#define HGI_ENC(I, ID, SE)                                                                                        \
    do {                                                                                                          \
        static const hipError_t lds0 = static_lds_is_empty(reinterpret_cast<const void *>(&k_enc_tiles<I, ID, SE, TH>)); \
        if (lds0 != hipSuccess) return lds0;                                                                      \
        hipLaunchKernelGGL((k_enc_tiles<I, ID, SE, TH>), blocks, b, lds, s, img, grid, f, k, lut, sd, g, r.aligned); \
    } while (0)
    // (seed PLANES without a cone -- SEEDED == 1 -- exist for 64-row tiles only: six fused levels, hgi_capi.hip host_banded)
#define HGI_ENC_ID(I, ID)                                                     \
    do {                                                                      \
        if (cone) HGI_ENC(I, ID, 2);                                          \
        else if (!seeds) HGI_ENC(I, ID, 0);                                   \
        else if constexpr (TH == 64) HGI_ENC(I, ID, 1);                       \
        else return hipErrorInvalidValue;                                     \
    } while (0)
#define HGI_ENC_I(I)                                                          \
    do {                                                                      \
        if (ident) HGI_ENC_ID(I, true); else HGI_ENC_ID(I, false);            \
    } while (0)
    if (interp == kInterpCrossed) HGI_ENC_I(kInterpCrossed); else HGI_ENC_I(kInterpLeftTop);
#undef HGI_ENC_I
#undef HGI_ENC_ID
#undef HGI_ENC
    return hipGetLastError();
}

#endif  // HGI_FUSED_ENCODE

}  // namespace hgi
