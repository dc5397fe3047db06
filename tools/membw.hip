// HBM bandwidth microbenchmark for MI355X (gfx950): what a streaming kernel can actually reach on this
// box, to put the codec kernels' roofline fraction in context.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -o membw tools/membw.hip && ./membw [GiB per buffer]
// Prints GB/s for: read-only, write-only, and copies with different per-lane widths, unrolls, block
// sizes, cache policies and src/dst relative placements (channel aliasing).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));  \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

template <int UNROLL, int AUX_LD, int AUX_ST>
__global__ void k_copy(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint32_t bytes_lo, uint32_t chunks)
{
    // buffers < 4 GiB: raw buffer ops with the byte offset in a VGPR
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src), 0, bytes_lo, 0x00020000);
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, bytes_lo, 0x00020000);
    // each block moves UNROLL consecutive slabs of blockDim.x * 16 B
    const uint32_t slab = blockDim.x * 16u;
    for (uint32_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const uint32_t base = c * slab * UNROLL + threadIdx.x * 16u;
        v4u v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + u * slab, 0, AUX_LD);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) __builtin_amdgcn_raw_buffer_store_b128(v[u], rd, base + u * slab, 0, AUX_ST);
    }
}

template <int UNROLL>
__global__ void k_read(const uint8_t *__restrict__ src, uint32_t *__restrict__ sink, uint32_t bytes_lo, uint32_t chunks)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src), 0, bytes_lo, 0x00020000);
    const uint32_t slab = blockDim.x * 16u;
    uint32_t acc = 0;
    for (uint32_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const uint32_t base = c * slab * UNROLL + threadIdx.x * 16u;
        v4u v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + u * slab, 0, 0);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;   // never true for the test pattern; keeps the loads alive
}

template <int UNROLL, int AUX_ST>
__global__ void k_write(uint8_t *__restrict__ dst, uint32_t bytes_lo, uint32_t chunks)
{
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, bytes_lo, 0x00020000);
    const uint32_t slab = blockDim.x * 16u;
    const v4u v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (uint32_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const uint32_t base = c * slab * UNROLL + threadIdx.x * 16u;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) __builtin_amdgcn_raw_buffer_store_b128(v, rd, base + u * slab, 0, AUX_ST);
    }
}

// tile-shaped copy: one wave per 128 x 64 tile of a W-wide image, 8 row-groups of 8 rows (the codec's access shape)
template <int AUX_ST>
__global__ void k_copy_tiles(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint32_t bytes_lo, uint32_t W,
                             uint32_t tiles_x, uint32_t ntiles)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src), 0, bytes_lo, 0x00020000);
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, bytes_lo, 0x00020000);
    const uint32_t b = blockIdx.x, x = b & 7u, per = ntiles >> 3;
    const uint32_t t = x * per + (b >> 3);   // XCD-contiguous, ntiles % 8 == 0 assumed
    const uint32_t ty = t / tiles_x, tx = t - ty * tiles_x;
    const uint32_t lane = threadIdx.x;
    const uint32_t base = (ty * 64u + (lane >> 3)) * W + tx * 128u + (lane & 7u) * 16u;
    v4u v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + j * 8u * W, 0, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) __builtin_amdgcn_raw_buffer_store_b128(v[j], rd, base + j * 8u * W, 0, AUX_ST);
}

// general tile copy: one wave per TWT x THT tile (TWT * THT = 8192 B), rows of TWT / 16 lanes; order = 0: XCD-contiguous
// row-major tiles; 1: plain row-major block order
template <int TWT, int AUX_LD, int AUX_ST>
__global__ void k_copy_shape(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint32_t bytes_lo, uint32_t W,
                             uint32_t ntiles, int order)
{
    constexpr int LPR = TWT / 16, RPL = 64 / LPR, THT = 8192 / TWT, NLD = THT / RPL;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(src), 0, bytes_lo, 0x00020000);
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, bytes_lo, 0x00020000);
    const uint32_t tiles_x = W / TWT;
    const uint32_t b = blockIdx.x, per = ntiles >> 3;
    const uint32_t t = order == 0 ? (b & 7u) * per + (b >> 3) : b;
    const uint32_t ty = t / tiles_x, tx = t - ty * tiles_x;
    const uint32_t lane = threadIdx.x;
    const uint32_t base = (ty * THT + lane / LPR) * W + tx * TWT + (lane % LPR) * 16u;
    v4u v[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + j * RPL * W, 0, AUX_LD);
#pragma unroll
    for (int j = 0; j < NLD; ++j) __builtin_amdgcn_raw_buffer_store_b128(v[j], rd, base + j * RPL * W, 0, AUX_ST);
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

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 1.0;
    const size_t n = (size_t)(gib * (1ull << 30)) & ~((size_t)(1 << 20) - 1);   // bytes per buffer, < 4 GiB
    const size_t slack = 64u << 20;
    uint8_t *a, *b;
    uint32_t *sink;
    CK(hipMalloc(&a, n + slack));
    CK(hipMalloc(&b, n + slack));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 0x5A, n + slack));
    CK(hipMemset(b, 0, n + slack));
    Timer T;
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    printf("device %s, %d CUs, buffers %.2f GiB, a=%p b=%p\n", pr.gcnArchName, pr.multiProcessorCount, n / double(1 << 30), a, b);
    auto gbs = [&](double bytes, double ms) { return bytes / ms * 1e-6; };
    const uint32_t lo = (uint32_t)n;

#define RUN_COPY(U, BS, LD, ST, GRID, DOFF)                                                                       \
    do {                                                                                                           \
        const uint32_t chunks = (uint32_t)(n / ((size_t)(BS) * 16 * (U)));                                          \
        const uint32_t grid = (GRID) ? (uint32_t)(GRID) : chunks;                                                  \
        double t = T.ms([&] { hipLaunchKernelGGL((k_copy<U, LD, ST>), dim3(grid), dim3(BS), 0, 0, a, b + (DOFF), lo, chunks); }); \
        printf("copy  unroll %d block %4d ld_aux %d st_aux %d grid %8u dst_off %9zu : %7.1f GB/s (%.3f ms)\n", U, BS, LD, ST, grid, \
               (size_t)(DOFF), gbs(2.0 * n, t), t);                                                                \
    } while (0)

    // per-lane work / block size / exact grid
    RUN_COPY(1, 256, 0, 0, 0, 0);
    RUN_COPY(2, 256, 0, 0, 0, 0);
    RUN_COPY(4, 256, 0, 0, 0, 0);
    RUN_COPY(8, 256, 0, 0, 0, 0);
    RUN_COPY(8, 64, 0, 0, 0, 0);
    RUN_COPY(4, 64, 0, 0, 0, 0);
    RUN_COPY(4, 1024, 0, 0, 0, 0);
    // cache policy (2 = nt)
    RUN_COPY(4, 256, 0, 2, 0, 0);
    RUN_COPY(4, 256, 2, 2, 0, 0);
    RUN_COPY(4, 256, 2, 0, 0, 0);
    RUN_COPY(8, 64, 0, 2, 0, 0);
    RUN_COPY(4, 256, 1, 1, 0, 0);
    RUN_COPY(4, 256, 3, 3, 0, 0);
    // persistent grids
    RUN_COPY(4, 256, 0, 2, 256 * 4, 0);
    RUN_COPY(4, 256, 0, 2, 256 * 8, 0);
    RUN_COPY(4, 256, 0, 2, 256 * 16, 0);
    RUN_COPY(8, 64, 0, 2, 256 * 16, 0);
    RUN_COPY(8, 64, 0, 2, 256 * 32, 0);
    // relative placement of dst (channel aliasing between the read and the write stream)
    for (size_t off : {(size_t)0, (size_t)256, (size_t)1024, (size_t)4096, (size_t)16384, (size_t)65536, (size_t)(1 << 20),
                       (size_t)(3 << 20), (size_t)(17 << 20) + 4096}) {
        RUN_COPY(4, 256, 0, 2, 0, off);
    }
    {
        const uint32_t chunks = (uint32_t)(n / (256 * 16 * 4));
        double t = T.ms([&] { hipLaunchKernelGGL((k_read<4>), dim3(chunks), dim3(256), 0, 0, a, sink, lo, chunks); });
        printf("read  unroll 4 block 256 exact grid : %7.1f GB/s (%.3f ms)\n", gbs(1.0 * n, t), t);
        t = T.ms([&] { hipLaunchKernelGGL((k_read<8>), dim3(256 * 16), dim3(256), 0, 0, a, sink, lo, (uint32_t)(n / (256 * 16 * 8))); });
        printf("read  unroll 8 block 256 grid 4096  : %7.1f GB/s (%.3f ms)\n", gbs(1.0 * n, t), t);
        t = T.ms([&] { hipLaunchKernelGGL((k_write<4, 0>), dim3(chunks), dim3(256), 0, 0, b, lo, chunks); });
        printf("write unroll 4 block 256 aux 0      : %7.1f GB/s (%.3f ms)\n", gbs(1.0 * n, t), t);
        t = T.ms([&] { hipLaunchKernelGGL((k_write<4, 2>), dim3(chunks), dim3(256), 0, 0, b, lo, chunks); });
        printf("write unroll 4 block 256 aux nt     : %7.1f GB/s (%.3f ms)\n", gbs(1.0 * n, t), t);
    }
    {
        // tile-shaped copy of frames of 4096 x 4096 laid back to back (the codec's access pattern without any arithmetic)
        const uint32_t W = 4096, tiles_x = W / 128, ntiles = (uint32_t)(n / (128 * 64));
        for (size_t off : {(size_t)0, (size_t)4096, (size_t)(1 << 20)}) {
            double t = T.ms([&] { hipLaunchKernelGGL((k_copy_tiles<2>), dim3(ntiles), dim3(64), 0, 0, a, b + off, lo, W, tiles_x, ntiles); });
            printf("tile copy 128x64 per wave, nt stores, dst_off %8zu : %7.1f GB/s (%.3f ms)\n", off, gbs(2.0 * n, t), t);
        }
        double t = T.ms([&] { hipLaunchKernelGGL((k_copy_tiles<0>), dim3(ntiles), dim3(64), 0, 0, a, b, lo, W, tiles_x, ntiles); });
        printf("tile copy 128x64 per wave, default stores           : %7.1f GB/s (%.3f ms)\n", gbs(2.0 * n, t), t);
    }
    {
        const uint32_t W = 4096, ntiles = (uint32_t)(n / 8192);
#define RUN_SHAPE(TWT, LD, ST, ORD)                                                                                        \
    do {                                                                                                                    \
        double t_ = T.ms([&] { hipLaunchKernelGGL((k_copy_shape<TWT, LD, ST>), dim3(ntiles), dim3(64), 0, 0, a, b, lo, W, ntiles, ORD); }); \
        printf("shape copy %4d x %3d per wave, ld_aux %d st_aux %d, %s : %7.1f GB/s (%.3f ms)\n", TWT, 8192 / TWT, LD, ST,      \
               ORD ? "row-major blocks " : "XCD-contiguous   ", gbs(2.0 * n, t_), t_);                                        \
    } while (0)
        RUN_SHAPE(64, 0, 2, 0);
        RUN_SHAPE(128, 0, 2, 0);
        RUN_SHAPE(256, 0, 2, 0);
        RUN_SHAPE(512, 0, 2, 0);
        RUN_SHAPE(1024, 0, 2, 0);
        RUN_SHAPE(128, 0, 2, 1);
        RUN_SHAPE(256, 0, 2, 1);
        RUN_SHAPE(1024, 0, 2, 1);
        RUN_SHAPE(128, 2, 2, 0);
        RUN_SHAPE(256, 2, 2, 0);
        RUN_SHAPE(1024, 2, 2, 0);
        RUN_SHAPE(128, 0, 0, 0);
    }
    double t = T.ms([&] { CK(hipMemcpyAsync(b, a, n, hipMemcpyDeviceToDevice, 0)); });
    printf("hipMemcpyAsync D2D : %7.1f GB/s (%.3f ms)\n", gbs(2.0 * n, t), t);
    return 0;
}
