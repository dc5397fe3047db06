/*
 * hgi.h -- C ABI of the MI355X-native HGI encode/decode core (libhgi_hip.so).
 *
 * This is the drop-in boundary for the hot path of pl0q1n/RustyHGI: the
 * per-level grid interpolation + residual quantize loop.  The reference has no
 * FFI today; each entry point below names the reference interface it replaces
 * (paths relative to the reference repository).  INTEGRATION.md shows the
 * `extern "C"` block a maintainer adds on the Rust side.
 *
 * Conventions
 *  - plain pointers and sizes, no C++/torch types; never throws or aborts
 *    across the ABI: every call returns hgi_status, hgi_last_error() holds the
 *    thread-local message of the last failure;
 *  - images and grids are tightly packed row-major u8, stride == width
 *    (reference: GrayImage / Grid, src/grid.rs:2-27); a batch is `batch`
 *    frames `frame_stride` bytes apart;
 *  - there is NO CPU fallback in this library: without a usable HIP device
 *    hgi_ctx_create fails with HGI_EDEVICE;
 *  - a ctx is not thread-safe; distinct ctxs are independent;
 *  - encode never modifies its input (the reference consumes it by value,
 *    src/encoder.rs:39);
 *  - `levels` in 0..=31 (1 << e on u32, src/utils.rs:17); levels == 0 makes
 *    grid == image; width or height == 0 is a successful no-op.
 */
#ifndef HGI_H_
#define HGI_H_

#include <stddef.h>
#include <stdint.h>

/* The shared object exports the entry points declared here and nothing else (built with        */
/* -fvisibility=hidden and a linker version script; tests/test_abi.py compares both directions). */
#if defined(__GNUC__) || defined(__clang__)
#define HGI_API __attribute__((visibility("default")))
#else
#define HGI_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hgi_ctx hgi_ctx;

/* src/interpolator.rs:15 (LeftTop), :30 (Crossed) -- the two implemented predictors */
typedef enum { HGI_INTERP_LEFTTOP = 0, HGI_INTERP_CROSSED = 1 } hgi_interp;

/* src/quantizator.rs:3-8 QuantizationLevel (bincode variant index) */
typedef enum {
    HGI_QUANT_LOSSLESS = 0,
    HGI_QUANT_LOW = 1,
    HGI_QUANT_MEDIUM = 2,
    HGI_QUANT_HIGH = 3
} hgi_quant_level;

typedef enum {
    HGI_OK = 0,
    HGI_EINVAL = 1,
    HGI_ENOMEM = 2,
    HGI_EDEVICE = 3,
    HGI_EUNSUPPORTED = 4
} hgi_status;

/* Which device implementation serves encode/decode (testing / benchmarking knob).   */
/* AUTO = FUSED.  Both run on the GPU; neither is a CPU path.                       */
typedef enum {
    HGI_PATH_AUTO = 0,
    HGI_PATH_LEVELWISE = 1, /* one launch per level, global-memory stencil            */
    HGI_PATH_FUSED = 2      /* all levels of a tile in one launch, LDS-resident       */
} hgi_path;

/* synthetic inputs of the harness; XY is benches/bench.rs:26-28 */
typedef enum { HGI_SYNTH_XY = 0, HGI_SYNTH_NOISE = 1, HGI_SYNTH_RAMP = 2 } hgi_synth_kind;

/* ---- context ---------------------------------------------------------------------- */
/* Owns the device id, a stream and scratch memory.  device >= 0 is a HIP ordinal.     */
HGI_API hgi_status hgi_ctx_create(int device, hgi_ctx **out);
HGI_API void hgi_ctx_destroy(hgi_ctx *ctx);
/* Borrow the caller's hipStream_t, used verbatim: NULL is HIP's default (null) stream,   */
/* which is what torch.cuda.current_stream() is unless a side stream is active.           */
HGI_API hgi_status hgi_ctx_set_stream(hgi_ctx *ctx, void *hip_stream);
/* Return to the ctx's private non-blocking stream (the state after hgi_ctx_create).      */
HGI_API hgi_status hgi_ctx_use_own_stream(hgi_ctx *ctx);
HGI_API hgi_status hgi_ctx_set_path(hgi_ctx *ctx, hgi_path path);
/* Pre-size scratch so later *_dev calls allocate nothing (needed before graph capture). */
/* Scratch only grows, and growing it FREES the old block: a HIP graph captured from calls */
/* that used scratch (pyramids deeper than eight levels; the level-wise path) keeps        */
/* pointing at the block it was captured with, so reserve for the largest shape the ctx    */
/* will ever see BEFORE capturing, and do not let a later, larger call on the same ctx     */
/* grow it while such graphs are alive.                                                    */
/* Covers hgi_encode_u8_dev / hgi_decode_u8_dev on `batch` frames of this shape AND the       */
/* host-pointer calls hgi_encode_u8 / hgi_decode_u8 on one such frame (their staging, the      */
/* seed planes of a banded deep pyramid); the host-pointer BATCH calls size their own slots.   */
HGI_API hgi_status hgi_ctx_reserve(hgi_ctx *ctx, uint32_t width, uint32_t height, uint32_t levels,
                           size_t batch);
/* Bytes of device scratch the ctx owns right now (0 after hgi_ctx_create): memory accounting,  */
/* and the way to see that a reserved ctx does not re-allocate.                                */
HGI_API hgi_status hgi_ctx_scratch_bytes(hgi_ctx *ctx, size_t *bytes);
HGI_API hgi_status hgi_sync(hgi_ctx *ctx);
HGI_API const char *hgi_last_error(void);
HGI_API const char *hgi_version(void);

/* ---- quantizers (host side: any Quantizator is tabulated into 256 bytes) ----------- */
/* replaces Linear::from(QuantizationLevel) + Linear::error, src/quantizator.rs:41-63,71 */
HGI_API hgi_status hgi_linear_lut(int level, uint8_t lut[256], uint8_t *max_err);
/* replaces NoOp::quantize, src/quantizator.rs:26-29 */
HGI_API void hgi_noop_lut(uint8_t lut[256]);

/* ---- host-pointer, synchronous ------------------------------------------------------ */
/* replaces Encoder::<I,Q>::new(..).encode(image) -> Grid, src/encoder.rs:18,39          */
HGI_API hgi_status hgi_encode_u8(hgi_ctx *ctx, const uint8_t *img, uint32_t width, uint32_t height,
                         uint32_t levels, hgi_interp interp, const uint8_t lut[256],
                         uint8_t *grid_out);
/* replaces Decoder::<I>::new(..).decode((w,h), levels, &grid) -> GrayImage, src/decoder.rs:14,18 */
HGI_API hgi_status hgi_decode_u8(hgi_ctx *ctx, const uint8_t *grid, uint32_t width, uint32_t height,
                         uint32_t levels, hgi_interp interp, uint8_t *img_out);

/* ---- device-pointer, asynchronous on the ctx stream, batched -------------------------- */
/* Same semantics per frame; `lut` is a HOST pointer (256 bytes, copied into the launch).   */
/* frame_stride >= width*height; d_img / d_grid must not alias.                            */
HGI_API hgi_status hgi_encode_u8_dev(hgi_ctx *ctx, const void *d_img, uint32_t width, uint32_t height,
                             uint32_t levels, hgi_interp interp, const uint8_t lut[256],
                             void *d_grid, size_t batch, size_t frame_stride);
HGI_API hgi_status hgi_decode_u8_dev(hgi_ctx *ctx, const void *d_grid, uint32_t width, uint32_t height,
                             uint32_t levels, hgi_interp interp, void *d_img, size_t batch,
                             size_t frame_stride);

/* ---- host-pointer batch calls ------------------------------------------------------------ */
/* `batch` frames in HOST memory, frame f at base + f * frame_stride (frame_stride >= w*h). */
/* Synchronous like hgi_encode_u8 / hgi_decode_u8, but the frames are pipelined through the  */
/* device in chunks on two internal streams, so uploads overlap downloads (PCIe is full      */
/* duplex) and the kernels disappear behind the transfers.  Input and output must not alias. */
HGI_API hgi_status hgi_encode_u8_batch(hgi_ctx *ctx, const uint8_t *imgs, uint32_t width, uint32_t height,
                               uint32_t levels, hgi_interp interp, const uint8_t lut[256],
                               uint8_t *grids_out, size_t batch, size_t frame_stride);
HGI_API hgi_status hgi_decode_u8_batch(hgi_ctx *ctx, const uint8_t *grids, uint32_t width, uint32_t height,
                               uint32_t levels, hgi_interp interp, uint8_t *imgs_out, size_t batch,
                               size_t frame_stride);

/* ---- entropy front end (no counterpart in the reference, would precede the   */
/* DEFLATE of src/archive.rs:36) ---------------------------------------------------------- */
/* Byte histogram of each frame of a grid batch, on the device, async on the ctx stream:     */
/* d_hist[256*f + v] = number of pixels of frame f equal to v.  d_hist: 256*batch uint64 of  */
/* device memory (overwritten).  Any width, stride and alignment.                            */
HGI_API hgi_status hgi_histogram_u8_dev(hgi_ctx *ctx, const void *d_grid, uint32_t width, uint32_t height,
                                size_t batch, size_t frame_stride, void *d_hist);

/* ---- harness helpers (not part of the reference surface) ------------------------------ */
/* Fill `batch` frames with a synthetic pattern; frame f uses index first_frame + f.        */
HGI_API hgi_status hgi_synth_u8_dev(hgi_ctx *ctx, hgi_synth_kind kind, uint64_t seed,
                            uint64_t first_frame, uint32_t width, uint32_t height, void *d_out,
                            size_t batch, size_t frame_stride);
/* Streaming copy of n bytes (16-B vectors): the same-run HBM copy ceiling.                 */
HGI_API hgi_status hgi_copy_u8_dev(hgi_ctx *ctx, const void *d_src, void *d_dst, size_t n);
/* Per-frame statistics of before/after pairs as `hgi test` prints them (src/main.rs:84-92): */
/* out[3*f+0] = sum of squared differences, +1 = max abs difference, +2 = count of differing */
/* pixels.  d_out is device memory for 3*batch uint64.                                       */
HGI_API hgi_status hgi_diff_stats_dev(hgi_ctx *ctx, const void *d_before, const void *d_after,
                              uint32_t width, uint32_t height, size_t batch, size_t frame_stride,
                              void *d_out);
/* ---- entropy stage on the device (the step behind src/archive.rs:34-40) ------------------- */
/* Raw DEFLATE (RFC 1951: one dynamic-Huffman block of literals and run matches -- a byte that */
/* repeats its predecessor is covered by matches of distance 1) of the bincode image of a Grid */
/* -- u64 w*h, the w*h residual bytes at d_grid, u64 w -- i.e. exactly the bytes that follow   */
/* the metadata in a .hgi archive.  Tokens, histogram, bit counting, scan and bit packing run  */
/* on the device; the 286-symbol code and the block header are built on the host.  Any inflate */
/* reads the result (the reference's flate2 DeflateDecoder included).  Residual grids are      */
/* noise around zero, in which LZ77 finds runs and nothing else: the stream is within a few %  */
/* of DEFLATE at the best level on smooth images and smaller than it on busy ones (LENA /      */
/* Medium: 14.0 against 16.0 kB) -- and written two to four orders of magnitude sooner.        */
/* `out` is host memory; *bytes receives the stream length; HGI_EINVAL if cap is too small     */
/* (w*h + w*h/8 + 1024 always suffices).  Synchronous on the ctx stream.                        */
HGI_API hgi_status hgi_deflate_grid_dev(hgi_ctx *ctx, const void *d_grid, uint32_t width, uint32_t height,
                                uint8_t *out, size_t cap, size_t *bytes);
/* `batch` grids `frame_stride` bytes apart in device memory: stream f is written to            */
/* out + f * out_stride (host memory; out_stride is also each stream's capacity), sizes[f] = its */
/* length.  The batch is pipelined in groups of frames (256 MiB of stream buffer each): one    */
/* launch per phase per group, the host building a group's codes while the device histograms   */
/* the next, a group's streams going down while the next is coded.  The downloads bound the    */
/* call; they are faster into pinned host memory (a 1 GiB batch: 5.5 ms pageable, 4.5 pinned). */
/* `out` is written from a second stream internally; the call returns when everything is done. */
HGI_API hgi_status hgi_deflate_grids_dev(hgi_ctx *ctx, const void *d_grids, uint32_t width, uint32_t height, size_t batch,
                                 size_t frame_stride, uint8_t *out, size_t out_stride, size_t *sizes);
/* The same batch with the streams PACKED into one host buffer of `cap` bytes: stream f occupies */
/* out[offsets[f] .. offsets[f] + sizes[f]), offsets ascending and multiples of 64.  The sizes are */
/* known from the histograms before anything is packed, so a group's streams lie back to back on  */
/* the device and come down with ONE copy per group instead of one per frame (each copy costs a   */
/* fixed ~12 us on top of its bytes).  The gap between a stream's end and the next stream's start  */
/* (at most 63 bytes) is zero.  HGI_EINVAL when `cap` does not hold them all                       */
/* (batch * (w*h + w*h/8 + 1088) always suffices).                                                  */
HGI_API hgi_status hgi_deflate_grids_packed_dev(hgi_ctx *ctx, const void *d_grids, uint32_t width, uint32_t height, size_t batch,
                                                size_t frame_stride, uint8_t *out, size_t cap, size_t *offsets, size_t *sizes);
/* The same with the grid in host memory (what pairs with hgi_encode_u8).                      */
HGI_API hgi_status hgi_deflate_grid(hgi_ctx *ctx, const uint8_t *grid, uint32_t width, uint32_t height,
                            uint8_t *out, size_t cap, size_t *bytes);
/* The code construction alone (host only): lengths (<= 15) and bit-reversed canonical codes    */
/* of the 286 symbols -- literals 0..255, end of block 256 (hist[256]), match lengths 257..285 */
/* -- for the given counts, and the block header (BFINAL = 1, dynamic, 286 + 2 codes; distance */
/* code 0 = distance 1 is the one-bit code "0") that announces them; *header_bits = its length.*/
HGI_API hgi_status hgi_huffman_plan(const uint64_t hist[286], uint8_t lens[286], uint16_t codes[286],
                            uint8_t *header, size_t header_cap, size_t *header_bits);

/* ---- plane placement (no reference counterpart: the reference's buffers are Vec<u8>) ---- */
/* MI355X's HBM falls into a few classes of physical memory; a launch that streams one buffer */
/* in and another out runs 4-8 % faster when the two lie in different classes. */
/* hgi_planes_alloc returns `count` device buffers of at least `bytes` each such that          */
/* planes[i] and planes[i+1] lie in different classes over their whole length: image -> grid   */
/* -> image chains alternate through the array.  RELEASE THEM WITH hgi_planes_free ONLY.       */
/* Best effort, found by timing a decode launch between candidates against a same-class        */
/* yardstick; *separated (optional) is 1 when every neighbouring pair was seen to stream at    */
/* the fast rate, 0 when that could not be established -- the planes are valid either way.     */
/*  - planes below 128 MiB: plain hipMalloc, not probed (such streams live in the 256 MiB      */
/*    Infinity Cache);                                                                         */
/*  - 128 MiB up to 1 GiB: allocated at 1 GiB each so that the probe can tell (a launch over   */
/*    two of them no longer fits that cache: a lone 16384 x 16384 frame gains 2-3 %); if the   */
/*    device lacks the memory for that they come back at `bytes`, unplaced;                    */
/*  - exactly 1 GiB: whole hipMalloc allocations are the candidates (typically 30-50 ms; up   */
/*    to ten extra candidates and spacers when the first ones share a class);                  */
/*  - above 1 GiB: each plane is one reserved address range onto which physical chunks of      */
/*    1 GiB (hipMemCreate) are mapped, chosen chunk by chunk (size rounded up to whole GiB).   */
/*    Bounded: at most 3 x the requested bytes are ever created as candidates, plus at most    */
/*    96 GiB of never-mapped spacers when the driver keeps handing out one class; what is not  */
/*    handed out is released before the call returns (three planes of 8 GiB: 1-2 s).           */
/* HGI_NO_PLACEMENT=1 in the environment skips all of it (plain allocations): the one variable */
/* the library reads.  Call it while the device is otherwise idle: it measures.                */
HGI_API hgi_status hgi_planes_alloc(hgi_ctx *ctx, size_t bytes, uint32_t count, void **planes, int *separated);
/* One line on what the last hgi_planes_alloc of this ctx found and did -- candidates created,   */
/* how many share which class, which line-up the planes got -- for logs and bench records (how   */
/* fast a large encode runs follows it).  Owned by the ctx, overwritten by its    */
/* next hgi_planes_alloc; "" before the first.                                                   */
HGI_API const char *hgi_planes_report(hgi_ctx *ctx);
HGI_API hgi_status hgi_planes_free(hgi_ctx *ctx, uint32_t count, void **planes);
/* The probe itself: mean milliseconds of one decode launch streaming d_src -> d_dst over      */
/* min(bytes, 2 GiB).  Overwrites d_dst.  Compare pairings of the caller's own buffers with it. */
HGI_API hgi_status hgi_probe_pair_u8_dev(hgi_ctx *ctx, const void *d_src, void *d_dst, size_t bytes, float *ms);

/* hipEvent pair on the ctx stream: start, ...launches..., stop -> elapsed milliseconds.     */
HGI_API hgi_status hgi_timer_start(hgi_ctx *ctx);
HGI_API hgi_status hgi_timer_stop(hgi_ctx *ctx, float *elapsed_ms);

#ifdef __cplusplus
}
#endif
#endif /* HGI_H_ */
