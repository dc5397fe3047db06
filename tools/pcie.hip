// Host <-> device transfer microbenchmark: what the host-pointer entry points (hgi_encode_u8 / hgi_decode_u8)
// can hope for on this box.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -o pcie tools/pcie.hip && ./pcie
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));  \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <class F>
static double best_us(F f, int reps = 7)
{
    f();
    double best = 1e30;
    for (int i = 0; i < reps; ++i) {
        double t0 = now_us();
        f();
        double t = now_us() - t0;
        if (t < best) best = t;
    }
    return best;
}

int main()
{
    hipStream_t s, s2;
    CK(hipStreamCreate(&s));
    CK(hipStreamCreate(&s2));
    const size_t sizes[] = {64u << 10, 2073600, 16u << 20, 256u << 20};
    for (size_t n : sizes) {
        uint8_t *d, *d2, *pin, *pin2;
        CK(hipMalloc(&d, n));
        CK(hipMalloc(&d2, n));
        CK(hipHostMalloc(&pin, n, hipHostMallocDefault));
        CK(hipHostMalloc(&pin2, n, hipHostMallocDefault));
        std::vector<uint8_t> page(n, 1), page2(n, 2);
        memset(pin, 3, n);
        memset(pin2, 4, n);
        auto gb = [&](double us) { return n / us * 1e-3; };
        printf("---- %zu bytes\n", n);
        double t;
        t = best_us([&] { memcpy(page2.data(), page.data(), n); });
        printf("host memcpy pageable->pageable          %9.1f us  %6.1f GB/s\n", t, gb(t));
        t = best_us([&] { memcpy(pin, page.data(), n); });
        printf("host memcpy pageable->pinned            %9.1f us  %6.1f GB/s\n", t, gb(t));
        t = best_us([&] { CK(hipMemcpyAsync(d, page.data(), n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); });
        printf("H2D pageable (hipMemcpyAsync + sync)    %9.1f us  %6.1f GB/s\n", t, gb(t));
        t = best_us([&] { CK(hipMemcpyAsync(page2.data(), d, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); });
        printf("D2H pageable                            %9.1f us  %6.1f GB/s\n", t, gb(t));
        t = best_us([&] { CK(hipMemcpyAsync(d, pin, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); });
        printf("H2D pinned                              %9.1f us  %6.1f GB/s\n", t, gb(t));
        t = best_us([&] { CK(hipMemcpyAsync(pin2, d, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); });
        printf("D2H pinned                              %9.1f us  %6.1f GB/s\n", t, gb(t));
        t = best_us([&] {
            CK(hipMemcpyAsync(d, pin, n, hipMemcpyHostToDevice, s));
            CK(hipMemcpyAsync(pin2, d2, n, hipMemcpyDeviceToHost, s2));
            CK(hipStreamSynchronize(s));
            CK(hipStreamSynchronize(s2));
        });
        printf("H2D + D2H pinned, concurrent streams    %9.1f us  %6.1f GB/s each way\n", t, gb(t));
        // pageable round trip as host_roundtrip does today (H2D, D2H back to back on one stream)
        t = best_us([&] {
            CK(hipMemcpyAsync(d, page.data(), n, hipMemcpyHostToDevice, s));
            CK(hipMemcpyAsync(page2.data(), d, n, hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
        });
        printf("pageable H2D then D2H, one stream       %9.1f us\n", t);
        // staged: user -> pinned chunks -> device, device -> pinned chunks -> user, chunked so that the CPU copy
        // of chunk i+1 overlaps the DMA of chunk i
        for (size_t chunk : {(size_t)256 << 10, (size_t)1 << 20, (size_t)4 << 20}) {
            if (chunk >= n && chunk != ((size_t)256 << 10)) continue;
            const size_t nch = (n + chunk - 1) / chunk;
            std::vector<hipEvent_t> ev(nch);
            for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            t = best_us([&] {
                for (size_t i = 0; i < nch; ++i) {
                    const size_t off = i * chunk, len = off + chunk <= n ? chunk : n - off;
                    memcpy(pin + off, page.data() + off, len);
                    CK(hipMemcpyAsync(d + off, pin + off, len, hipMemcpyHostToDevice, s));
                }
                for (size_t i = 0; i < nch; ++i) {
                    const size_t off = i * chunk, len = off + chunk <= n ? chunk : n - off;
                    CK(hipMemcpyAsync(pin2 + off, d + off, len, hipMemcpyDeviceToHost, s));
                    CK(hipEventRecord(ev[i], s));
                }
                for (size_t i = 0; i < nch; ++i) {
                    const size_t off = i * chunk, len = off + chunk <= n ? chunk : n - off;
                    CK(hipEventSynchronize(ev[i]));
                    memcpy(page2.data() + off, pin2 + off, len);
                }
            });
            printf("staged round trip, %4zu KiB chunks       %9.1f us\n", chunk >> 10, t);
            for (auto &e : ev) CK(hipEventDestroy(e));
        }
        // registering the user's pages in place
        t = best_us([&] { CK(hipHostRegister(page.data(), n, hipHostRegisterDefault)); CK(hipHostUnregister(page.data())); }, 3);
        printf("hipHostRegister + Unregister            %9.1f us\n", t);
        CK(hipFree(d));
        CK(hipFree(d2));
        CK(hipHostFree(pin));
        CK(hipHostFree(pin2));
    }
    return 0;
}
