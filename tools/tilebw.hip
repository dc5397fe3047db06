// What can a TILE-structured streaming kernel reach on MI355X?  (tools/membw.hip asked it of linear kernels, with buffers
// wherever the allocator put them.)  Here: planes placed in different HBM regions (hgi_planes_alloc), one wave per
// 8 KiB tile of 64 frames of 4096 x 4096, XCD-contiguous tile ranges -- the codec's structure with the arithmetic taken
// out -- and the structural knobs one at a time:
//   shape   TW x TH = 128x64, 256x32, 512x16, 1024x8
//   order   row-major tiles inside a frame, or bands of R tile rows walked column-major (the codec's order)
//   phases  all loads then all stores (1), or P groups of loads each followed by its stores
//   lds     registers only, or through LDS (write, read back) like the codec
//   lds cap extra dynamic LDS to bound the waves per CU
// Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -Iinclude tools/tilebw.hip -Lrustyhgi_amd -lhgi_hip -Wl,-rpath,$PWD/rustyhgi_amd -o tilebw
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "hgi.h"

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));  \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

struct Geo {
    uint32_t W, H, tiles_x, tiles_y, ntiles, band;   // band = 0: row-major
};

__device__ __forceinline__ uint32_t range_first(uint32_t ntiles, uint32_t x) { return x * (ntiles >> 3) + (x < (ntiles & 7u) ? x : (ntiles & 7u)); }

// TWT x THT tile per wave; LPR lanes per row, RPL rows per load instruction, NLD load instructions per tile
template <int TWT, int PHASES, bool VIA_LDS, int AUX_LD, int AUX_ST>
__global__ __launch_bounds__(64) void k_tiles(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, Geo g)
{
    constexpr int LPR = TWT / 16, RPL = 64 / LPR, THT = 8192 / TWT, NLD = THT / RPL, PER = NLD / PHASES;
    static_assert(NLD % PHASES == 0, "phases divide the loads");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t b = blockIdx.x;
    uint32_t t = __builtin_amdgcn_readfirstlane(range_first(g.ntiles, b & 7u) + (b >> 3));
    const uint32_t tpf = g.tiles_x * g.tiles_y, frame = t / tpf, tt = t - frame * tpf;
    uint32_t ty = tt / g.tiles_x, tx = tt - ty * g.tiles_x;
    if (g.band) {
        const uint32_t per = g.band * g.tiles_x, nfull = (g.tiles_y / g.band) * per;
        uint32_t rows = g.band, row0, r;
        if (tt < nfull) {
            const uint32_t bd = tt / per;
            r = tt - bd * per;
            row0 = bd * g.band;
        } else {
            rows = g.tiles_y % g.band;
            r = tt - nfull;
            row0 = g.tiles_y - rows;
        }
        tx = r / rows;
        ty = row0 + (r - tx * rows);
    }
    const size_t fbytes = (size_t)g.W * g.H;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src) + frame * fbytes, 0, (uint32_t)fbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst + frame * fbytes, 0, (uint32_t)fbytes, 0x00020000);
    const uint32_t lane = threadIdx.x;
    const uint32_t base = (ty * THT + lane / LPR) * g.W + tx * TWT + (lane % LPR) * 16u;
    const uint32_t W = __builtin_amdgcn_readfirstlane(g.W);
#pragma unroll
    for (int p = 0; p < PHASES; ++p) {
        v4u v[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, base, (p * PER + j) * RPL * W, AUX_LD);
        if (VIA_LDS) {
#pragma unroll
            for (int j = 0; j < PER; ++j) *reinterpret_cast<v4u *>(smem + ((p * PER + j) * 64 + lane) * 16) = v[j];
            asm volatile("" ::: "memory");
            // read back another lane's chunk (lane ^ 9: other row, other column), so that the data really crosses LDS
#pragma unroll
            for (int j = 0; j < PER; ++j) v[j] = *reinterpret_cast<const v4u *>(smem + ((p * PER + j) * 64 + (lane ^ 9u)) * 16);
            const uint32_t l2 = lane ^ 9u;
            const uint32_t base2 = (ty * THT + l2 / LPR) * g.W + tx * TWT + (l2 % LPR) * 16u;
#pragma unroll
            for (int j = 0; j < PER; ++j) __builtin_amdgcn_raw_buffer_store_b128(v[j], rd, base2, (p * PER + j) * RPL * W, AUX_ST);
        } else {
#pragma unroll
            for (int j = 0; j < PER; ++j) __builtin_amdgcn_raw_buffer_store_b128(v[j], rd, base, (p * PER + j) * RPL * W, AUX_ST);
        }
    }
}

// 128 x 64 tile, rows loaded the way the codec loads them: even rows and odd rows in separate instructions, each with its
// own cache policy; SPLIT: the even rows as left halves (bytes 0..63 of each line, the part a neighbour tile's halo loads
// also touch) and right halves (64..127) in separate instructions with separate policies.
template <int AUX_EVEN_L, int AUX_EVEN_R, int AUX_ODD, bool SPLIT>
__global__ __launch_bounds__(64) void k_tiles_rows(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, Geo g)
{
    const uint32_t b = blockIdx.x;
    uint32_t t = __builtin_amdgcn_readfirstlane(range_first(g.ntiles, b & 7u) + (b >> 3));
    const uint32_t tpf = g.tiles_x * g.tiles_y, frame = t / tpf, tt = t - frame * tpf;
    const uint32_t per = g.band * g.tiles_x, bd = tt / per, r = tt - bd * per;
    const uint32_t tx = r / g.band, ty = bd * g.band + (r - tx * g.band);
    const size_t fbytes = (size_t)g.W * g.H;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src) + frame * fbytes, 0, (uint32_t)fbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst + frame * fbytes, 0, (uint32_t)fbytes, 0x00020000);
    const uint32_t lane = threadIdx.x, W = __builtin_amdgcn_readfirstlane(g.W);
    const uint32_t tile = ty * 64u * g.W + tx * 128u;
    v4u e[4], o[4];
    uint32_t be[4];
    if (SPLIT) {
        // instruction j < 2: left halves of even rows 2 * (16 j + lane / 4); j >= 2: right halves
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            be[j] = tile + 2u * (16u * (j & 1) + (lane >> 2)) * g.W + (j >> 1) * 64u + (lane & 3u) * 16u;
            e[j] = j < 2 ? __builtin_amdgcn_raw_buffer_load_b128(rs, be[j], 0, AUX_EVEN_L) : __builtin_amdgcn_raw_buffer_load_b128(rs, be[j], 0, AUX_EVEN_R);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            be[j] = tile + 2u * (8u * j + (lane >> 3)) * g.W + (lane & 7u) * 16u;
            e[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, be[j], 0, AUX_EVEN_L);
        }
    }
    const uint32_t bo = tile + (2u * (lane >> 3) + 1u) * g.W + (lane & 7u) * 16u;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, bo, j * 16 * W, AUX_ODD);
#pragma unroll
    for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b128(e[j], rd, be[j], 0, 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b128(o[j], rd, bo, j * 16 * W, 2);
}

// one 128 x 64 tile per WORKGROUP of WAVES waves: each wave moves 8 / WAVES of the tile's eight 8-row groups (fewer
// loads in flight per wave, more and shorter-lived waves -- the direction in which a linear copy differs from a tile copy)
template <int WAVES, int AUX_LD>
__global__ __launch_bounds__(64 * WAVES) void k_tiles_wg(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, Geo g)
{
    constexpr int PER = 8 / WAVES;
    const uint32_t b = blockIdx.x;
    uint32_t t = __builtin_amdgcn_readfirstlane(range_first(g.ntiles, b & 7u) + (b >> 3));
    const uint32_t tpf = g.tiles_x * g.tiles_y, frame = t / tpf, tt = t - frame * tpf;
    const uint32_t per = g.band * g.tiles_x, bd = tt / per, r = tt - bd * per;
    const uint32_t tx = r / g.band, ty = bd * g.band + (r - tx * g.band);
    const size_t fbytes = (size_t)g.W * g.H;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src) + frame * fbytes, 0, (uint32_t)fbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst + frame * fbytes, 0, (uint32_t)fbytes, 0x00020000);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, W = __builtin_amdgcn_readfirstlane(g.W);
    const uint32_t base = (ty * 64u + wave * PER * 8u + (lane >> 3)) * g.W + tx * 128u + (lane & 7u) * 16u;
    v4u v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, base, j * 8 * W, AUX_LD);
#pragma unroll
    for (int j = 0; j < PER; ++j) __builtin_amdgcn_raw_buffer_store_b128(v[j], rd, base, j * 8 * W, 2);
}

typedef v4u copy_v4u;
__global__ void k_linear(const copy_v4u *__restrict__ src, copy_v4u *__restrict__ dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

struct Timer {
    hipEvent_t a, b;
    Timer()
    {
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
    }
    template <class F>
    double ms(F f, int reps = 10)
    {
        f();
        f();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < reps; ++i) f();
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float t;
        CK(hipEventElapsedTime(&t, a, b));
        return t / reps;
    }
};

int main()
{
    const uint32_t W = 4096, H = 4096, F = 64;
    const size_t n = (size_t)W * H * F;
    hgi_ctx *ctx = nullptr;
    if (hgi_ctx_create(0, &ctx) != HGI_OK) return 2;
    void *planes[3];
    int separated = 0;
    if (hgi_planes_alloc(ctx, n, 3, planes, &separated) != HGI_OK) {
        fprintf(stderr, "%s\n", hgi_last_error());
        return 2;
    }
    uint8_t *a = (uint8_t *)planes[0], *b = (uint8_t *)planes[1], *c = (uint8_t *)planes[2];
    CK(hipMemset(a, 0x5A, n));
    printf("planes separated: %d   (a -> b: different regions; a -> c: the same region)\n", separated);
    Timer T;
    auto gbs = [&](double ms) { return 2.0 * n / ms * 1e-6; };
    {
        const size_t n16 = n / 16;
        double t1 = T.ms([&] { hipLaunchKernelGGL(k_linear, dim3((uint32_t)(n16 / 256)), dim3(256), 0, 0, (const copy_v4u *)a, (copy_v4u *)b, n16); });
        double t2 = T.ms([&] { hipLaunchKernelGGL(k_linear, dim3((uint32_t)(n16 / 256)), dim3(256), 0, 0, (const copy_v4u *)a, (copy_v4u *)c, n16); });
        printf("linear copy, 16 B per lane, 256-lane blocks, nt/nt : other region %7.1f GB/s (%.4f ms)   same region %7.1f GB/s (%.4f ms)\n",
               gbs(t1), t1, gbs(t2), t2);
    }
#define RUN(TWT, PH, LDS, LD, ST, BAND, CAP)                                                                                  \
    do {                                                                                                                       \
        Geo g = {W, H, W / (TWT), H / (8192 / (TWT)), 0, (uint32_t)(BAND)};                                                    \
        g.ntiles = g.tiles_x * g.tiles_y * F;                                                                                  \
        const size_t lds = ((LDS) ? 8192 : 0) + (size_t)(CAP);                                                                 \
        double t1 = T.ms([&] { hipLaunchKernelGGL((k_tiles<TWT, PH, LDS, LD, ST>), dim3(g.ntiles), dim3(64), lds, 0, a, b, g); }); \
        double t2 = T.ms([&] { hipLaunchKernelGGL((k_tiles<TWT, PH, LDS, LD, ST>), dim3(g.ntiles), dim3(64), lds, 0, a, c, g); }); \
        printf("tile %4d x %2d  phases %d  %s  ld %d st %d  band %2d  lds %5zu B : other region %7.1f GB/s (%.4f ms)   same %7.1f (%.4f ms)\n", \
               TWT, 8192 / (TWT), PH, (LDS) ? "via LDS  " : "registers", LD, ST, BAND, lds, gbs(t1), t1, gbs(t2), t2);        \
    } while (0)
    // the codec's shape: order
    RUN(128, 1, false, 0, 2, 0, 0);
    RUN(128, 1, false, 0, 2, 4, 0);
    RUN(128, 1, false, 0, 2, 8, 0);
    RUN(128, 1, false, 0, 2, 16, 0);
    // phases
    RUN(128, 2, false, 0, 2, 8, 0);
    RUN(128, 4, false, 0, 2, 8, 0);
    RUN(128, 8, false, 0, 2, 8, 0);
    // through LDS, and occupancy (8 KiB + cap per wave: 160 KiB / that = waves per CU)
    RUN(128, 1, true, 0, 2, 8, 0);
    RUN(128, 1, true, 0, 2, 8, 2048);
    RUN(128, 1, true, 0, 2, 8, 8192);
    RUN(128, 2, true, 0, 2, 8, 0);
    RUN(128, 1, false, 0, 2, 8, 5120);
    RUN(128, 1, false, 0, 2, 8, 8192);
    RUN(128, 1, false, 0, 2, 8, 16384);
    // cache policies
    RUN(128, 1, false, 2, 2, 8, 0);
    RUN(128, 1, false, 0, 0, 8, 0);
    RUN(128, 1, false, 3, 3, 8, 0);
    // shapes (band in tile rows of that shape)
    RUN(256, 1, false, 0, 2, 0, 0);
    RUN(256, 1, false, 0, 2, 8, 0);
    RUN(256, 1, false, 0, 2, 16, 0);
    RUN(512, 1, false, 0, 2, 0, 0);
    RUN(512, 1, false, 0, 2, 16, 0);
    RUN(512, 1, false, 0, 2, 32, 0);
    RUN(1024, 1, false, 0, 2, 0, 0);
    RUN(1024, 1, false, 0, 2, 32, 0);
    RUN(1024, 1, false, 0, 2, 64, 0);
    RUN(256, 2, false, 0, 2, 8, 0);
    RUN(512, 2, false, 0, 2, 16, 0);
    RUN(1024, 2, false, 2, 2, 0, 0);
#define RUN_ROWS(EL, ER, OD, SPLIT)                                                                                           \
    do {                                                                                                                       \
        Geo g = {W, H, W / 128, H / 64, 0, 8};                                                                                 \
        g.ntiles = g.tiles_x * g.tiles_y * F;                                                                                  \
        double t1 = T.ms([&] { hipLaunchKernelGGL((k_tiles_rows<EL, ER, OD, SPLIT>), dim3(g.ntiles), dim3(64), 0, 0, a, b, g); }); \
        printf("tile 128 x 64 by row class, band 8: even rows %s ld %d / %d, odd rows ld %d : %7.1f GB/s (%.4f ms)\n",       \
               (SPLIT) ? "as two half-line instructions" : "whole lines", EL, ER, OD, gbs(t1), t1);                            \
    } while (0)
    RUN_ROWS(0, 0, 0, false);
    RUN_ROWS(0, 0, 2, false);      // the codec today
    RUN_ROWS(2, 2, 2, false);
    RUN_ROWS(0, 0, 2, true);
    RUN_ROWS(0, 2, 2, true);       // right halves of the even rows nt as well
    RUN_ROWS(2, 2, 2, true);
    RUN_ROWS(0, 0, 2, false);
    RUN_ROWS(0, 2, 2, true);
#define RUN_WG(WV, LD)                                                                                                        \
    do {                                                                                                                       \
        Geo g = {W, H, W / 128, H / 64, 0, 8};                                                                                 \
        g.ntiles = g.tiles_x * g.tiles_y * F;                                                                                  \
        double t1 = T.ms([&] { hipLaunchKernelGGL((k_tiles_wg<WV, LD>), dim3(g.ntiles), dim3(64 * (WV)), 0, 0, a, b, g); });   \
        printf("tile 128 x 64 per workgroup of %d waves (%d loads per lane), ld %d st 2, band 8 : %7.1f GB/s (%.4f ms)\n", WV, \
               8 / (WV), LD, gbs(t1), t1);                                                                                     \
    } while (0)
    RUN_WG(1, 0);
    RUN_WG(2, 0);
    RUN_WG(4, 0);
    RUN_WG(8, 0);
    RUN_WG(1, 2);
    RUN_WG(4, 2);
    RUN_WG(8, 2);
    hgi_planes_free(ctx, 3, planes);
    hgi_ctx_destroy(ctx);
    return 0;
}
