// HBM regions, part 5 (round 4): what does the class structure of DESIGN.md 5.1 look like at the scale of the literal C3
// batch (three planes of 8 GiB), and can a plane be BUILT from physical chunks of one class with the HIP virtual-memory
// API (hipMemCreate / hipMemAddressReserve / hipMemMap)?
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I include tools/regions.hip -L rustyhgi_amd -lhgi_hip -Wl,-rpath,$PWD/rustyhgi_amd -o build_tools/regions
//   build_tools/regions [big_gib=160] [chunks=64]
//
// Part A: ONE hipMalloc of `big_gib` GiB; hgi_probe_pair_u8_dev (a decode launch src -> dst over 1 GiB) from window 0, from the
//         middle window and from the last window to every other 1 GiB window -> the class pattern along one allocation.
// Part B: `chunks` physical allocations of 1 GiB (hipMemCreate), each mapped at its own slot of one reserved range;
//         the same probes between chunks; and the same probe on two plain hipMalloc'ed buffers as the yardstick of what a
//         mapping of this kind costs by itself.
// Part C: a 4 GiB plane mapped from chunks of ONE class that are not neighbours, against one of the other class: does the
//         composed plane stream like a plain allocation of that class?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

#include "hgi.h"

#define CK(e)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (e);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #e, hipGetErrorString(e_));          \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)
#define HK(e)                                                                                   \
    do {                                                                                        \
        if ((e) != HGI_OK) {                                                                    \
            printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #e, hgi_last_error());               \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

static const size_t GiB = (size_t)1 << 30;
static hgi_ctx *ctx;

static float probe(const void *a, void *b, size_t bytes = GiB)
{
    float ms = 0;
    HK(hgi_probe_pair_u8_dev(ctx, a, b, bytes, &ms));
    return ms;
}

static void pattern(const char *what, const std::vector<float> &t, int self)
{
    float lo = 1e9f, hi = 0;
    for (size_t i = 0; i < t.size(); ++i)
        if ((int)i != self) {
            lo = t[i] < lo ? t[i] : lo;
            hi = t[i] > hi ? t[i] : hi;
        }
    const float thr = (lo + hi) / 2;
    std::string p;
    for (size_t i = 0; i < t.size(); ++i) p += (int)i == self ? '.' : (t[i] < thr ? 'F' : 's');
    printf("%s: min %.4f max %.4f ms  spread %.1f %%\n  %s\n", what, lo, hi, 100.0 * (hi - lo) / lo, p.c_str());
}

int main(int argc, char **argv)
{
    const size_t big_gib = argc > 1 ? (size_t)atol(argv[1]) : 160;
    const int chunks = argc > 2 ? atoi(argv[2]) : 64;
    HK(hgi_ctx_create(0, &ctx));
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    printf("free %.1f GiB of %.1f GiB\n", free_b / (double)GiB, total_b / (double)GiB);
    // clocks
    {
        void *a, *b;
        CK(hipMalloc(&a, GiB));
        CK(hipMalloc(&b, GiB));
        for (int i = 0; i < 20; ++i) probe(a, b);
        printf("yardstick: two plain 1 GiB hipMalloc buffers: %.4f ms, reversed %.4f ms\n", probe(a, b), probe(b, a));
        CK(hipFree(a));
        CK(hipFree(b));
    }
    // ---- part A
    if (big_gib) {
        uint8_t *big = nullptr;
        CK(hipMalloc(reinterpret_cast<void **>(&big), big_gib * GiB));
        printf("part A: one hipMalloc of %zu GiB at %p\n", big_gib, (void *)big);
        for (int i = 0; i < 10; ++i) probe(big, big + GiB);
        for (size_t base : {(size_t)0, big_gib / 2, big_gib - 1}) {
            std::vector<float> t(big_gib, 0.f);
            for (size_t j = 0; j < big_gib; ++j)
                if (j != base) t[j] = probe(big + base * GiB, big + j * GiB);
            char what[64];
            snprintf(what, sizeof what, "window %zu -> window j", base);
            pattern(what, t, (int)base);
        }
        // an 8 GiB stream inside it: window pairs (i, i + 8 .. ) at 2 GiB probes -- what a 512-frame plane pair would see
        printf("  2 GiB probes, src = [0, 2) GiB, dst = [8 + 2 j, 10 + 2 j):");
        for (size_t j = 0; 10 + 2 * j <= big_gib && j < 24; ++j) printf(" %.4f", probe(big, big + (8 + 2 * j) * GiB, 2 * GiB));
        printf("\n");
        CK(hipFree(big));
    }
    // ---- part B
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("part B: hipMemCreate, granularity %zu bytes, %d chunks of 1 GiB\n", gran, chunks);
    std::vector<hipMemGenericAllocationHandle_t> h(chunks);
    for (int i = 0; i < chunks; ++i) CK(hipMemCreate(&h[i], GiB, &prop, 0));
    uint8_t *va = nullptr;
    CK(hipMemAddressReserve(reinterpret_cast<void **>(&va), (size_t)chunks * GiB, 0, nullptr, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int i = 0; i < chunks; ++i) CK(hipMemMap(va + (size_t)i * GiB, GiB, 0, h[i], 0));
    CK(hipMemSetAccess(va, (size_t)chunks * GiB, &acc, 1));
    for (int i = 0; i < 10; ++i) probe(va, va + GiB);
    std::vector<float> t0(chunks, 0.f), tm(chunks, 0.f);
    for (int j = 1; j < chunks; ++j) t0[j] = probe(va, va + (size_t)j * GiB);
    pattern("chunk 0 -> chunk j", t0, 0);
    for (int j = 0; j < chunks; ++j)
        if (j != chunks / 2) tm[j] = probe(va + (size_t)(chunks / 2) * GiB, va + (size_t)j * GiB);
    pattern("chunk mid -> chunk j", tm, chunks / 2);
    // ---- part C: two 4 GiB planes composed of chunks of one class each (relative to chunk 0), chunks taken from all over
    float lo = 1e9f, hi = 0;
    for (int j = 1; j < chunks; ++j) {
        lo = t0[j] < lo ? t0[j] : lo;
        hi = t0[j] > hi ? t0[j] : hi;
    }
    const float thr = (lo + hi) / 2;
    std::vector<int> same, other;      // same class as chunk 0 (slow against it) / the other class
    for (int j = 1; j < chunks; ++j) (t0[j] < thr ? other : same).push_back(j);
    printf("part C: %zu chunks in chunk 0's class, %zu in the other (spread %.1f %%)\n", same.size() + 1, other.size(), 100.0 * (hi - lo) / lo);
    if ((hi - lo) / lo > 0.02 && same.size() >= 8 && other.size() >= 4) {
        CK(hipMemUnmap(va, (size_t)chunks * GiB));
        uint8_t *pa = nullptr, *pb = nullptr, *pc = nullptr;
        CK(hipMemAddressReserve(reinterpret_cast<void **>(&pa), 4 * GiB, 0, nullptr, 0));
        CK(hipMemAddressReserve(reinterpret_cast<void **>(&pb), 4 * GiB, 0, nullptr, 0));
        CK(hipMemAddressReserve(reinterpret_cast<void **>(&pc), 4 * GiB, 0, nullptr, 0));
        // spread the picks over the list: every (size / 4)-th member
        for (int i = 0; i < 4; ++i) {
            CK(hipMemMap(pa + (size_t)i * GiB, GiB, 0, h[same[(size_t)i * (same.size() / 8)]], 0));
            CK(hipMemMap(pc + (size_t)i * GiB, GiB, 0, h[same[(size_t)(4 + i) * (same.size() / 8)]], 0));
            CK(hipMemMap(pb + (size_t)i * GiB, GiB, 0, h[other[(size_t)i * (other.size() / 4)]], 0));
        }
        CK(hipMemSetAccess(pa, 4 * GiB, &acc, 1));
        CK(hipMemSetAccess(pb, 4 * GiB, &acc, 1));
        CK(hipMemSetAccess(pc, 4 * GiB, &acc, 1));
        for (int i = 0; i < 10; ++i) probe(pa, pb, 2 * GiB);
        printf("  composed planes, 2 GiB probes at offsets 0 and 2 GiB:\n");
        for (size_t off : {(size_t)0, 2 * GiB}) {
            printf("   offset %zu GiB: same -> other %.4f  other -> same %.4f  same -> same' %.4f ms\n", off / GiB, probe(pa + off, pb + off, 2 * GiB),
                   probe(pb + off, pc + off, 2 * GiB), probe(pa + off, pc + off, 2 * GiB));
        }
        // the real thing: 4 GiB = 256 frames of 4096^2 through the product entry points on the composed planes
        uint8_t lut[256], err;
        HK(hgi_linear_lut(2, lut, &err));
        HK(hgi_synth_u8_dev(ctx, HGI_SYNTH_RAMP, 0x48474933u, 0, 4096, 4096, pa, 256, (size_t)4096 * 4096));
        auto step = [&](uint8_t *img, uint8_t *grid, uint8_t *out, const char *what) {
            float e = 0, d = 0;
            for (int i = 0; i < 12; ++i) {
                HK(hgi_encode_u8_dev(ctx, img, 4096, 4096, 4, HGI_INTERP_CROSSED, lut, grid, 256, (size_t)4096 * 4096));
                HK(hgi_decode_u8_dev(ctx, grid, 4096, 4096, 4, HGI_INTERP_CROSSED, out, 256, (size_t)4096 * 4096));
            }
            const int reps = 8;
            for (int i = 0; i < reps; ++i) {
                float ms;
                HK(hgi_timer_start(ctx));
                HK(hgi_encode_u8_dev(ctx, img, 4096, 4096, 4, HGI_INTERP_CROSSED, lut, grid, 256, (size_t)4096 * 4096));
                HK(hgi_timer_stop(ctx, &ms));
                e += ms;
                HK(hgi_timer_start(ctx));
                HK(hgi_decode_u8_dev(ctx, grid, 4096, 4096, 4, HGI_INTERP_CROSSED, out, 256, (size_t)4096 * 4096));
                HK(hgi_timer_stop(ctx, &ms));
                d += ms;
            }
            printf("   256 x 4096^2 L4 Medium, %s: encode %.4f ms decode %.4f ms (per 64 frames: %.4f / %.4f)\n", what, e / reps, d / reps, e / reps / 4,
                   d / reps / 4);
        };
        step(pa, pb, pc, "image same / grid OTHER / out same (composed)");
        step(pa, pc, pb, "image same / grid same / out other (composed)");
        uint8_t *q[3];
        for (auto &p : q) CK(hipMalloc(reinterpret_cast<void **>(&p), 4 * GiB));
        CK(hipMemcpy(q[0], pa, 4 * GiB, hipMemcpyDeviceToDevice));
        step(q[0], q[1], q[2], "three plain hipMalloc planes");
    } else {
        printf("  (no two classes among the chunks: nothing to compose)\n");
    }
    hgi_ctx_destroy(ctx);
    return 0;
}
