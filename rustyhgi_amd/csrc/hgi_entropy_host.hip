// C ABI of libhgi_hip.so (include/hgi.h), part 2: the entropy stage on the device (the step behind src/archive.rs:34-40) --
// the host side of it: group pipeline, code construction per frame, downloads.  Kernels: hgi_entropy.hip; the code
// construction itself: hgi_huffman_host.h (plain C++, fuzzed under ASan/UBSan by the CPU suite).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <cstring>
#include <thread>
#include <vector>

#include "hgi_host.h"

using namespace hgi;
using namespace hgi::host;

#ifndef HGI_ENTROPY_GROUP_MIB
#define HGI_ENTROPY_GROUP_MIB 256     // stream buffer of one group of frames (hgi_knobs.h: a constant in the release library)
#endif

extern "C" {

// ---- entropy stage --------------------------------------------------------------------------------------------
hgi_status hgi_huffman_plan(const uint64_t hist[286], uint8_t lens[286], uint16_t codes[286], uint8_t *header, size_t header_cap,
                            size_t *header_bits)
{
    if (!hist || !lens || !codes || !header || !header_bits) return fail(HGI_EINVAL, "NULL argument");
    uint64_t any = 0;
    for (int i = 0; i < kDeflateSymbols; ++i) any |= hist[i];
    if (!any) return fail(HGI_EINVAL, "empty histogram");
    *header_bits = huffman_plan(hist, lens, codes, header, header_cap);
    if (!*header_bits) return fail(HGI_EINVAL, "header buffer too small (%zu bytes)", header_cap);
    return HGI_OK;
}

// ---- the stage itself: `batch` grids, phase by phase, so that the host waits three times per GROUP of frames, not per frame
namespace {

constexpr size_t kHistBytes = (kMatchThresholds + 1) * kDeflateSymbols * 8;      // per frame, contiguous (one download)

struct DeflateGeom {
    uint64_t n;
    uint32_t nchunks;
    size_t dev_cap;       // bytes of stream buffer per frame on the device
    size_t group;         // frames whose stream buffers live in scratch at once
    size_t need;          // scratch bytes
};

DeflateGeom deflate_geom(uint64_t n, size_t batch)
{
    DeflateGeom g;
    g.n = n;
    g.nchunks = huffman_chunks(n);
    g.dev_cap = align_up((size_t)(n + n / 4) + 4096, 256);      // an optimal code averages < 9 bits per byte
    // a group's stream buffers: 256 MiB (two groups are in flight: one being packed, one being downloaded; smaller groups
    // shorten the un-overlapped head and tail of the pipeline, more groups cost more synchronisations:
    // tools/entropy_packed_time.py on the knobs build, profiles/r03_entropy_groups.txt)
    const int group_knob = HGI_KNOB(HGI_ENTROPY_GROUP_MIB, HGI_ENTROPY_GROUP_MIB);
    const size_t group_mib = group_knob > 0 ? (size_t)group_knob : (size_t)256;
    size_t group = (group_mib << 20) / g.dev_cap;
    if (group < 1) group = 1;
    if (group > batch) group = batch ? batch : 1;
    if (group > 256) group = 256;
    // equal groups: the last one is not a straggler
    const size_t ngroups = batch ? (batch + group - 1) / group : 1;
    if (batch) group = (batch + ngroups - 1) / ngroups;
    g.group = group;
    const size_t sets = ngroups > 1 ? 2 : 1;
    g.need = group * (sets * (kHistBytes + g.dev_cap) + kPlanBytes + (size_t)g.nchunks * 12 + 64) + (batch ? batch : 1) * 8 + 4096;
    return g;
}

using huff::FramePlan;

// The stage over `batch` grids, in groups of g.group frames, software-pipelined so that the device always has the next
// thing queued while the host builds codes or waits for a download:
//     device, c->stream :  hist(0) | hist(1) pack(0) | hist(2) pack(1) | ...
//     host              :          | plan(0)         | plan(1)         | ...      (several threads, one frame each)
//     device, pipe[1]   :                            | streams(0) down | streams(1) down ...
// hist = token histograms (one launch per group), plan = codes + headers, pack = count / scan / pack (three launches per
// group).  Stream sizes are known from the histograms, so the downloads are queued without waiting for the pack.
// offsets == nullptr: stream f goes to out + f * out_stride (cap = room per stream).  offsets != nullptr (packed): the
// streams of a group lie back to back on the device (64-byte aligned starts) and come down with ONE copy per group into
// out + offsets[f]; cap = room in `out` altogether.
hgi_status deflate_frames(hgi_ctx *c, const uint8_t *d_grids, uint32_t w, uint32_t h, size_t batch, size_t stride, uint8_t *out,
                          size_t out_stride, size_t cap, size_t *sizes, size_t *offsets = nullptr)
{
    const bool packed = offsets != nullptr;
    size_t packed_at = 0;                      // packed: where the next group starts in `out`
    const DeflateGeom g = deflate_geom((uint64_t)w * h, batch);
    const uint64_t n = g.n;
    // the bincode image of Grid { buffer: Vec<u8>, width: usize } (src/grid.rs:2-5): u64 length, the bytes, u64 width
    uint8_t prefix[8], suffix[8];
    for (int i = 0; i < 8; ++i) {
        prefix[i] = (uint8_t)(n >> (8 * i));
        suffix[i] = (uint8_t)((uint64_t)w >> (8 * i));
    }
    // Group boundaries: equal groups.  (Small first groups that double up to the full size -- to shorten the pipeline's
    // un-overlapped head, the first group's own histogram + plan + pack -- were tried: 9 groups instead of 6 for the C3
    // shard cost more in synchronisations than the head gave back: packed / strided 0.92 against 0.90, profiles/r03_entropy_groups.txt.)
    std::vector<size_t> starts;
    for (size_t at = 0; at < batch; at += g.group) starts.push_back(at);
    starts.push_back(batch);
    const size_t ngroups = starts.size() - 1;
    const bool piped = ngroups > 1;
    if (!n) {
        // nothing for the device to code: the front, then the tail, here
        std::vector<uint64_t> hist0((kMatchThresholds + 1) * kDeflateSymbols, 0);
        FramePlan p;
        if (!huff::plan_frame(reinterpret_cast<uint64_t (*)[kDeflateSymbols]>(hist0.data()), false, prefix, suffix, p))
            return fail(HGI_EDEVICE, "block header does not fit");
        const size_t total_bytes = (size_t)((p.exact_bits + 7) / 8), slot = align_up(total_bytes, 64);
        if (packed ? slot * batch > cap : total_bytes > cap)
            return fail(HGI_EINVAL, "output buffer too small: %zu bytes needed", packed ? slot * batch : total_bytes);
        for (size_t f = 0; f < batch; ++f) {
            uint8_t *dst = out + (packed ? f * slot : f * out_stride);
            if (packed) offsets[f] = f * slot;
            std::memset(dst, 0, packed ? slot : total_bytes);
            std::memcpy(dst, p.block.front, p.block.front_bytes);
            uint64_t at = p.block.base_bits;
            const uint8_t *tail = reinterpret_cast<const uint8_t *>(p.block.tail);
            for (uint32_t i = 0; i < p.block.tail_bits; ++i, ++at) dst[at >> 3] |= (uint8_t)(((tail[i >> 3] >> (i & 7)) & 1u) << (at & 7));
            sizes[f] = total_bytes;
        }
        return HGI_OK;
    }
    HGI_TRY(ws_ensure(c, g.need));
    HGI_TRY(pin_ensure(c, 2 * g.group * kHistBytes + align_up(batch * 8, 256) + 2 * g.group * kPlanBytes));
    if (piped) HGI_TRY(pipe_ensure(c));
    c->ws_used = 0;
    uint8_t *d_hist[2], *d_outs[2];
    for (int k = 0; k < 2; ++k) d_hist[k] = (k == 0 || piped) ? ws_take(c, g.group * kHistBytes) : d_hist[0];
    uint8_t *d_plans = ws_take(c, g.group * kPlanBytes);
    uint64_t *d_totals = reinterpret_cast<uint64_t *>(ws_take(c, batch * 8));
    uint64_t *d_off = reinterpret_cast<uint64_t *>(ws_take(c, g.group * (size_t)g.nchunks * 8 + 8));
    uint32_t *d_cbits = reinterpret_cast<uint32_t *>(ws_take(c, g.group * (size_t)g.nchunks * 4 + 8));
    for (int k = 0; k < 2; ++k) d_outs[k] = (k == 0 || piped) ? ws_take(c, g.group * g.dev_cap) : d_outs[0];
    c->ws_used = 0;
    if (!d_hist[0] || !d_hist[1] || !d_plans || !d_totals || !d_off || !d_cbits || !d_outs[0] || !d_outs[1])
        return fail(HGI_ENOMEM, "scratch exhausted (entropy stage)");
    uint64_t *h_hist[2] = {reinterpret_cast<uint64_t *>(c->pin), reinterpret_cast<uint64_t *>(c->pin + g.group * kHistBytes)};
    uint64_t *h_totals = reinterpret_cast<uint64_t *>(c->pin + 2 * g.group * kHistBytes);
    const uint32_t dist_code = 0u | (1u << 24);      // distance symbol 0 (= distance 1): the one-bit code "0"
    std::vector<FramePlan> plans(g.group);
    // The plan blocks go up from pinned memory, two sets: the copy is then truly asynchronous (the host plans group gi + 1
    // while the device still packs group gi), and a set is rewritten only after ev_hist of two groups later -- which the
    // stream reaches behind this set's upload -- has been waited for.
    DeflatePlan *h_plans[2];
    h_plans[0] = reinterpret_cast<DeflatePlan *>(c->pin + 2 * g.group * kHistBytes + align_up(batch * 8, 256));
    h_plans[1] = h_plans[0] + g.group;
    std::vector<uint64_t> promised(batch), fixed_bits(batch);      // per frame: the stream's bits, and those that are not tokens
    std::vector<size_t> group_at(ngroups, 0), group_bytes(ngroups, 0);      // packed: a group's place in `out` and its length
    hipStream_t down = piped ? c->pipe[1] : c->stream;
    hipEvent_t *ev_hist = c->ev_hist;
    auto first_of = [&](size_t gi) { return starts[gi]; };
    auto count_of = [&](size_t gi) { return starts[gi + 1] - starts[gi]; };
    auto queue_hist = [&](size_t gi) -> hipError_t {
        const int set = (int)(gi & 1);
        hipError_t e = launch_token_histogram(d_grids + first_of(gi) * stride, n, stride, (uint32_t)count_of(gi),
                                              reinterpret_cast<unsigned long long *>(d_hist[set]), c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(h_hist[set], d_hist[set], count_of(gi) * kHistBytes, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipEventRecord(ev_hist[set], c->stream);
        return e;
    };
    auto queue_downloads = [&](size_t gi) -> hipError_t {
        const int set = (int)(gi & 1);
        hipError_t e = hipSuccess;
        if (piped) e = hipStreamWaitEvent(down, c->ev_free[set], 0);
        if (packed) {      // the group's streams are contiguous on the device: one copy
            if (e == hipSuccess && group_bytes[gi])
                e = hipMemcpyAsync(out + group_at[gi], d_outs[set], group_bytes[gi], hipMemcpyDeviceToHost, down);
        } else {
            for (size_t f = 0; f < count_of(gi) && e == hipSuccess; ++f) {
                const size_t frame = first_of(gi) + f;
                e = hipMemcpyAsync(out + frame * out_stride, d_outs[set] + f * g.dev_cap, (size_t)((promised[frame] + 7) / 8), hipMemcpyDeviceToHost, down);
            }
        }
        if (piped && e == hipSuccess) e = hipEventRecord(c->ev_up[set], down);
        return e;
    };
    // anything that fails after work was queued: drain before the host buffers the queue refers to go away
    auto bail = [&](hgi_status st) {
        (void)hipStreamSynchronize(c->stream);
        if (piped) (void)hipStreamSynchronize(down);
        return st;
    };
#define DF_TRY(expr)                                                                                                             \
    do {                                                                                                                         \
        hipError_t e_ = (expr);                                                                                                  \
        if (e_ != hipSuccess) return bail(fail(HGI_EDEVICE, "%s: %s", #expr, hipGetErrorString(e_)));                           \
    } while (0)
    DF_TRY(queue_hist(0));
    for (size_t gi = 0; gi < ngroups; ++gi) {
        const int set = (int)(gi & 1);
        const size_t g0 = first_of(gi), cnt = count_of(gi);
        DF_TRY(hipEventSynchronize(ev_hist[set]));
        if (gi + 1 < ngroups) DF_TRY(queue_hist(gi + 1));              // the device has this to do while the host plans
        // codes on the host, a frame per thread
        {
            std::vector<int> status(cnt, 0);
            auto work = [&](size_t t, size_t nt) {
                for (size_t f = t; f < cnt; f += nt)
                    status[f] = huff::plan_frame(reinterpret_cast<uint64_t (*)[kDeflateSymbols]>(h_hist[set] + f * (kHistBytes / 8)), true, prefix,
                                                 suffix, plans[f]) ? 0 : 1;
            };
            size_t nt = cnt / 2;
            if (nt > 8) nt = 8;
            if (nt <= 1) {
                work(0, 1);
            } else {
                std::vector<std::thread> pool;
                for (size_t t = 1; t < nt; ++t) pool.emplace_back(work, t, nt);
                work(0, nt);
                for (auto &th : pool) th.join();
            }
            for (size_t f = 0; f < cnt; ++f)
                if (status[f]) return bail(fail(HGI_EDEVICE, "block header does not fit"));
        }
        // the histograms say exactly how long each stream will be: never start packing into a buffer it would overrun
        size_t dev_at = 0;                     // packed: running offset inside the group's device buffer
        for (size_t f = 0; f < cnt; ++f) {
            const FramePlan &p = plans[f];
            const size_t bytes = (size_t)((p.exact_bits + 7) / 8);
            if (p.exact_bits / 8 + 64 > g.dev_cap)
                return bail(fail(HGI_EDEVICE, "entropy stage: stream of %llu bytes exceeds its scratch", (unsigned long long)(p.exact_bits / 8)));
            if (!packed && bytes > cap) return bail(fail(HGI_EINVAL, "output buffer too small: %zu bytes needed", bytes));
            h_plans[set][f] = p.block;
            const uint64_t off = packed ? dev_at : f * g.dev_cap;
            h_plans[set][f].out_off[0] = (uint32_t)off;
            h_plans[set][f].out_off[1] = (uint32_t)(off >> 32);
            if (packed) {
                offsets[g0 + f] = packed_at + dev_at;
                dev_at += align_up(bytes, 64);
            }
            promised[g0 + f] = p.exact_bits;
            fixed_bits[g0 + f] = p.block.base_bits + p.block.tail_bits;
        }
        if (packed) {
            if (packed_at + dev_at > cap)
                return bail(fail(HGI_EINVAL, "output buffer too small: %zu bytes needed for the first %zu frames", packed_at + dev_at, g0 + cnt));
            group_at[gi] = packed_at;
            group_bytes[gi] = dev_at;
            packed_at += dev_at;
        }
        // one upload of the plans, count / scan / pack over the whole group (its stream buffers are free once the group
        // two back has been downloaded)
        if (piped && gi >= 2) DF_TRY(hipStreamWaitEvent(c->stream, c->ev_up[set], 0));
        DF_TRY(hipMemcpyAsync(d_plans, h_plans[set], cnt * kPlanBytes, hipMemcpyHostToDevice, c->stream));
        DF_TRY(launch_huffman_pack(d_grids + g0 * stride, n, stride, (uint32_t)cnt, d_plans, dist_code, d_cbits, d_off, d_totals + g0, d_outs[set],
                                   c->stream));
        if (piped) DF_TRY(hipEventRecord(c->ev_free[set], c->stream));
        // downloads lag one group behind, so that the device has hist(gi + 1) and pack(gi) queued while they run
        if (piped) {
            if (gi >= 1) DF_TRY(queue_downloads(gi - 1));
        } else {
            DF_TRY(queue_downloads(gi));
        }
    }
    if (piped) DF_TRY(queue_downloads(ngroups - 1));
    DF_TRY(hipMemcpyAsync(h_totals, d_totals, batch * 8, hipMemcpyDeviceToHost, c->stream));
    DF_TRY(hipStreamSynchronize(c->stream));
    if (piped) DF_TRY(hipStreamSynchronize(down));
#undef DF_TRY
    for (size_t f = 0; f < batch; ++f) {
        const uint64_t got = fixed_bits[f] + h_totals[f];
        if (got != promised[f])
            return fail(HGI_EDEVICE, "entropy stage: packed %llu bits where the histograms promised %llu", (unsigned long long)got,
                        (unsigned long long)promised[f]);
        sizes[f] = (size_t)((promised[f] + 7) / 8);
    }
    // packed: a group came down as ONE copy, alignment gaps included, and the kernels only write inside a stream -- so the
    // up to 63 bytes between one stream's end and the next one's 64-byte aligned start hold whatever the device scratch held
    // before.  Cleared here (the caller may write the whole buffer out): bytes outside every [offsets[f], offsets[f] + sizes[f])
    // but inside the packed region are zero.
    if (packed)
        for (size_t f = 0; f < batch; ++f) std::memset(out + offsets[f] + sizes[f], 0, align_up(sizes[f], 64) - sizes[f]);
    return HGI_OK;
}

}  // namespace

// host-pointer form (what pairs with hgi_encode_u8): the grid goes up into scratch behind the stage's own buffers
hgi_status hgi_deflate_grid(hgi_ctx *c, const uint8_t *grid, uint32_t w, uint32_t h, uint8_t *out, size_t cap, size_t *bytes)
{
    if (!c || !out || !bytes) return fail(HGI_EINVAL, "NULL argument");
    const size_t n = (size_t)w * h;
    if (n && !grid) return fail(HGI_EINVAL, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    const size_t front = align_up(deflate_geom(n, 1).need, 256);
    HGI_TRY(ws_ensure(c, front + n + 256));
    uint8_t *staged = c->ws + front;
    if (n) HIP_TRY(hipMemcpyAsync(staged, grid, n, hipMemcpyHostToDevice, c->stream));
    return deflate_frames(c, staged, w, h, 1, n, out, cap, cap, bytes);
}

hgi_status hgi_deflate_grid_dev(hgi_ctx *c, const void *d_grid, uint32_t w, uint32_t h, uint8_t *out, size_t cap, size_t *bytes)
{
    if (!c || !out || !bytes) return fail(HGI_EINVAL, "NULL argument");
    if ((uint64_t)w * h && !d_grid) return fail(HGI_EINVAL, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    return deflate_frames(c, static_cast<const uint8_t *>(d_grid), w, h, 1, (size_t)w * h, out, cap, cap, bytes);
}

hgi_status hgi_deflate_grids_dev(hgi_ctx *c, const void *d_grids, uint32_t w, uint32_t h, size_t batch, size_t frame_stride, uint8_t *out,
                                 size_t out_stride, size_t *sizes)
{
    if (!c || (batch && (!out || !sizes))) return fail(HGI_EINVAL, "NULL argument");
    if (batch == 0) return HGI_OK;
    const size_t n = (size_t)w * h;
    if (n && !d_grids) return fail(HGI_EINVAL, "NULL buffer");
    if (batch > 1 && frame_stride < n) return fail(HGI_EINVAL, "frame_stride %zu < width*height", frame_stride);
    HIP_TRY(hipSetDevice(c->device));
    return deflate_frames(c, static_cast<const uint8_t *>(d_grids), w, h, batch, frame_stride, out, out_stride, out_stride, sizes);
}

hgi_status hgi_deflate_grids_packed_dev(hgi_ctx *c, const void *d_grids, uint32_t w, uint32_t h, size_t batch, size_t frame_stride,
                                        uint8_t *out, size_t cap, size_t *offsets, size_t *sizes)
{
    if (!c || (batch && (!out || !sizes || !offsets))) return fail(HGI_EINVAL, "NULL argument");
    if (batch == 0) return HGI_OK;
    const size_t n = (size_t)w * h;
    if (n && !d_grids) return fail(HGI_EINVAL, "NULL buffer");
    if (batch > 1 && frame_stride < n) return fail(HGI_EINVAL, "frame_stride %zu < width*height", frame_stride);
    HIP_TRY(hipSetDevice(c->device));
    return deflate_frames(c, static_cast<const uint8_t *>(d_grids), w, h, batch, frame_stride, out, 0, cap, sizes, offsets);
}

}  // extern "C"
