// What does a back-to-back loop of SMALL launches cost on MI355X, whatever the kernel does?  The single-frame numbers
// of tools/size_sweep.py (1920 x 1080: a few microseconds per call) are launches queued on one stream and timed by two
// events around the loop, so each iteration = the dependent-kernel boundary + the kernel's own critical path.  This
// tool takes that path apart on the shape of a 1920 x 1080 call (510 blocks of one wave, the 128 x 32 geometry):
//   empty        nothing
//   kernarg      the codec's argument block (256-B table by value) copied to LDS
//   tile copy    the tile's 32 rows loaded (4 x 16 B per lane) and stored
//   + N chains   ... with N dependent LDS round trips (write, read another lane's value) in between: what a level costs
// Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o build_tools/launch_floor
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));  \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
struct Lut {
    uint32_t w[64];
};

__global__ __launch_bounds__(64) void k_empty() {}

__global__ __launch_bounds__(64) void k_kernarg(Lut lut, uint32_t *sink)
{
    extern __shared__ uint32_t lds[];
    lds[threadIdx.x] = lut.w[threadIdx.x];
    asm volatile("" ::: "memory");
    if (lds[(threadIdx.x + 1) & 63] == 0xdeadbeefu) sink[0] = 1;
}

template <int CHAINS, int ROWS>
__global__ __launch_bounds__(64) void k_tile(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint32_t W, uint32_t tiles_x, Lut lut)
{
    extern __shared__ uint32_t lds[];
    lds[threadIdx.x] = lut.w[threadIdx.x];
    const uint32_t b = blockIdx.x, ty = b / tiles_x, tx = b - ty * tiles_x, lane = threadIdx.x;
    const uint32_t base = (ty * ROWS + (lane >> 3)) * W + tx * 128u + (lane & 7u) * 16u;
    constexpr int NLD = ROWS / 8;
    v4u v[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) v[j] = *reinterpret_cast<const v4u *>(src + base + j * 8 * W);
    uint32_t x = v[0].x;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
        lds[64 + lane] = x;
        asm volatile("" ::: "memory");
        x = lds[64 + ((lane + 1 + c) & 63)] + lds[x & 63];
        asm volatile("" ::: "memory");
    }
    v[0].x ^= (x & 0u);
#pragma unroll
    for (int j = 0; j < NLD; ++j) *reinterpret_cast<v4u *>(dst + base + j * 8 * W) = v[j];
}

// Dispatch throughput: many workgroups that do (almost) nothing, with the resources of a codec block -- `T` threads,
// dynamic LDS, a register budget (launch bounds) -- and an optional sleep so that slots are really occupied for a while.
template <int T, int SLEEP>
__global__ __launch_bounds__(T) void k_dispatch(uint32_t *sink)
{
    extern __shared__ uint32_t lds[];
    if (SLEEP) {
        for (int i = 0; i < SLEEP; ++i) __builtin_amdgcn_s_sleep(127);      // 127 x 64 cycles each
    }
    if (sink == nullptr) lds[threadIdx.x] = 1;
}

template <typename F>
static float loop_us(F launch, hipStream_t s, int reps = 400)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 50; ++i) launch();
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f / reps;
}

int main()
{
    const uint32_t W = 1920, H = 1088, tiles_x = W / 128;
    uint8_t *src, *dst;
    uint32_t *sink;
    CK(hipMalloc(&src, (size_t)W * H));
    CK(hipMalloc(&dst, (size_t)W * H));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(src, 7, (size_t)W * H));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    Lut lut = {};
    const int b32 = tiles_x * (H / 32), b16 = tiles_x * (H / 16), b64 = tiles_x * (H / 64);
    printf("launch_floor: back-to-back launches on one stream, us per launch (events around 400 launches); 1920 x 1088\n");
    printf("  empty kernel, %4d blocks x 64                      %6.2f\n", b32, loop_us([&] { hipLaunchKernelGGL(k_empty, dim3(b32), dim3(64), 0, s); }, s));
    printf("  empty kernel,    1 block  x 64                      %6.2f\n", loop_us([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }, s));
    printf("  256-B table by value -> LDS, %4d blocks            %6.2f\n", b32, loop_us([&] { hipLaunchKernelGGL(k_kernarg, dim3(b32), dim3(64), 512, s, lut, sink); }, s));
#define ROW(CH, ROWS, NB)                                                                                                         \
    printf("  tile copy %3d x %2d (%4d blocks) + %2d LDS chains      %6.2f\n", 128, ROWS, NB, CH,                                   \
           loop_us([&] { hipLaunchKernelGGL((k_tile<CH, ROWS>), dim3(NB), dim3(64), 1024, s, src, dst, W, tiles_x, lut); }, s))
    ROW(0, 32, b32);
    ROW(4, 32, b32);
    ROW(8, 32, b32);
    ROW(16, 32, b32);
    ROW(0, 16, b16);
    ROW(8, 16, b16);
    ROW(16, 16, b16);
    ROW(0, 64, b64);
    ROW(8, 64, b64);
    ROW(16, 64, b64);
    // ---- dispatch throughput ----
    {
        const int NB = 131072;
        printf("dispatch throughput, %d workgroups per launch (us per launch -> workgroups / us -> waves / us):\n", NB);
#define DROW(T, SLEEP, LDS, LABEL)                                                                                                  \
        do {                                                                                                                          \
            const int nb = NB * 64 / T;                                                                                               \
            const float us = loop_us([&] { hipLaunchKernelGGL((k_dispatch<T, SLEEP>), dim3(nb), dim3(T), LDS, s, sink); }, s, 20);       \
            printf("  %-58s %8.1f  %7.0f  %7.0f\n", LABEL, us, nb / us, nb * (T / 64) / us);                                             \
        } while (0)
        DROW(64, 0, 0, "64 threads, no LDS, returns at once");
        DROW(64, 0, 7680, "64 threads, 7.5 KiB LDS, returns at once");
        DROW(64, 0, 4864, "64 threads, 4.75 KiB LDS, returns at once");
        DROW(128, 0, 15360, "128 threads, 15 KiB LDS (same waves, half the workgroups)");
        DROW(256, 0, 30720, "256 threads, 30 KiB LDS (same waves, a quarter of the workgroups)");
        DROW(256, 0, 0, "256 threads, no LDS");
        DROW(64, 3, 7680, "64 threads, 7.5 KiB LDS, sleeps ~10 us (slots stay full)");
        DROW(128, 3, 15360, "128 threads, 15 KiB LDS, sleeps ~10 us");
        DROW(256, 3, 30720, "256 threads, 30 KiB LDS, sleeps ~10 us");
    }
    // one sync per launch: what a caller who waits for each frame sees
    {
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        float tot = 0;
        for (int i = 0; i < 200; ++i) {
            CK(hipEventRecord(a, s));
            hipLaunchKernelGGL((k_tile<8, 32>), dim3(b32), dim3(64), 1024, s, src, dst, W, tiles_x, lut);
            CK(hipEventRecord(b, s));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            tot += ms;
        }
        printf("  tile copy 128 x 32 + 8 chains, events around EACH launch, synchronised   %6.2f\n", tot * 1e3f / 200);
    }
    return 0;
}
