// Internal host-side interface shared by the translation units behind the C ABI (include/hgi.h):
//   hgi_capi.hip          contexts, the codec entry points (device, host, host-batch, banded), harness helpers, timer
//   hgi_entropy_host.hip  the entropy stage's host pipeline (hgi_deflate_*, hgi_huffman_plan)
//   hgi_planes.hip        plane placement (hgi_planes_alloc / hgi_planes_free / hgi_probe_pair_u8_dev)
// Nothing here is exported: the library is built with -fvisibility=hidden and the version script hgi.map.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/hgi.h"
#include "hgi_kernels.h"
#include "hgi_knobs.h"

struct hgi_ctx {
    int device;
    hipStream_t own_stream, stream;
    hgi_path path;
    uint8_t *ws;
    size_t ws_bytes, ws_used;
    hipEvent_t ev0, ev1;      // hgi_timer_start / hgi_timer_stop, nothing else
    hipEvent_t ev_hist[2];    // entropy stage: "histograms of group set k are down"
    hipEvent_t ev_probe[2];   // placement probe
    // host-pointer batch calls (created on first use): pipe[0] uploads, pipe[1] runs the kernels and downloads;
    // three device slots, per slot one event "uploaded" and one "kernels done, input slot free"
    hipStream_t pipe[2];
    hipEvent_t ev_up[3], ev_free[3];
    hipEvent_t ev_band[16];   // banded single-frame calls: "band uploaded"
    bool have_pipe;
    uint8_t *pin;             // pinned host memory (entropy stage: histograms and stream sizes come down without stalling the host)
    size_t pin_bytes;
    int probe_resident_tiles; // -1 except inside the placement probe, whose decode launches run at a fixed occupancy (hgi_planes.hip)
    char planes_report[384];  // what the last hgi_planes_alloc on this ctx found and did (hgi_planes_report)
};

namespace hgi {
namespace host {

// status + thread-local message (hgi_last_error); printf-style
hgi_status fail(hgi_status st, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Scratch: a bump allocator over one device buffer that only grows between calls (growing frees the old block).
hgi_status ws_ensure(hgi_ctx *c, size_t bytes);
uint8_t *ws_take(hgi_ctx *c, size_t bytes);
// scratch bytes one encode or decode of this shape takes on this ctx's path (recursion of deep pyramids included)
size_t ws_need(const hgi_ctx *c, uint32_t w, uint32_t h, uint32_t levels, size_t batch, size_t stride);
// pinned host memory of the ctx, grown on demand
hgi_status pin_ensure(hgi_ctx *c, size_t bytes);
// the two internal streams + events of the host-pointer pipelines, created on first use
hgi_status pipe_ensure(hgi_ctx *c);

// The codec on device pointers, asynchronous on c->stream; scratch comes from ws_take (the caller has ensured ws_need and reset
// c->ws_used).  src/encoder.rs:39-71 / src/decoder.rs:18-46 per frame.
hgi_status encode_impl(hgi_ctx *c, const uint8_t *img, uint32_t w, uint32_t h, uint32_t levels, int interp, const uint8_t lut[256],
                       uint8_t *grid, size_t batch, size_t stride);
hgi_status decode_impl(hgi_ctx *c, const uint8_t *grid, uint32_t w, uint32_t h, uint32_t levels, int interp, uint8_t *img, size_t batch,
                       size_t stride);

}  // namespace host
}  // namespace hgi

#define HIP_TRY(expr)                                                                                            \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess)                                                                                    \
            return ::hgi::host::fail(e_ == hipErrorOutOfMemory ? HGI_ENOMEM : HGI_EDEVICE, "%s: %s", #expr,      \
                                     hipGetErrorString(e_));                                                     \
    } while (0)

#define HGI_TRY(expr)                  \
    do {                               \
        hgi_status s_ = (expr);        \
        if (s_ != HGI_OK) return s_;   \
    } while (0)
