//! Raw bindings of include/hgi.h (only what the codec surface needs).
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct HgiCtx {
    _private: [u8; 0],
}

pub const HGI_OK: c_int = 0;
pub const HGI_EINVAL: c_int = 1;
pub const HGI_ENOMEM: c_int = 2;
pub const HGI_EDEVICE: c_int = 3;
pub const HGI_EUNSUPPORTED: c_int = 4;
pub const HGI_INTERP_LEFTTOP: c_int = 0;
pub const HGI_INTERP_CROSSED: c_int = 1;

extern "C" {
    pub fn hgi_ctx_create(device: c_int, out: *mut *mut HgiCtx) -> c_int;
    pub fn hgi_ctx_destroy(ctx: *mut HgiCtx);
    pub fn hgi_last_error() -> *const c_char;
    pub fn hgi_linear_lut(level: c_int, lut: *mut u8, max_err: *mut u8) -> c_int;
    pub fn hgi_encode_u8(ctx: *mut HgiCtx, img: *const u8, width: u32, height: u32, levels: u32,
                         interp: c_int, lut: *const u8, grid_out: *mut u8) -> c_int;
    pub fn hgi_decode_u8(ctx: *mut HgiCtx, grid: *const u8, width: u32, height: u32, levels: u32,
                         interp: c_int, img_out: *mut u8) -> c_int;
    pub fn hgi_encode_u8_dev(ctx: *mut HgiCtx, d_img: *const c_void, width: u32, height: u32, levels: u32,
                             interp: c_int, lut: *const u8, d_grid: *mut c_void, batch: usize,
                             frame_stride: usize) -> c_int;
    pub fn hgi_decode_u8_dev(ctx: *mut HgiCtx, d_grid: *const c_void, width: u32, height: u32, levels: u32,
                             interp: c_int, d_img: *mut c_void, batch: usize, frame_stride: usize) -> c_int;
    pub fn hgi_sync(ctx: *mut HgiCtx) -> c_int;
    /// include/hgi.h: raw DEFLATE of a grid's bincode image, entropy-coded on the device (grid in host memory)
    pub fn hgi_deflate_grid(ctx: *mut HgiCtx, grid: *const u8, width: u32, height: u32, out: *mut u8, cap: usize,
                            bytes: *mut usize) -> c_int;
    /// include/hgi.h: `batch` frames in host memory, pipelined through the device (uploads overlap downloads)
    pub fn hgi_encode_u8_batch(ctx: *mut HgiCtx, imgs: *const u8, width: u32, height: u32, levels: u32, interp: c_int,
                               lut: *const u8, grids_out: *mut u8, batch: usize, frame_stride: usize) -> c_int;
    pub fn hgi_decode_u8_batch(ctx: *mut HgiCtx, grids: *const u8, width: u32, height: u32, levels: u32, interp: c_int,
                               imgs_out: *mut u8, batch: usize, frame_stride: usize) -> c_int;
    /// include/hgi.h: the entropy stage over a batch of device-resident grids, streams packed back to back into `out`
    /// (stream f = out[offsets[f] .. offsets[f] + sizes[f]], offsets multiples of 64): one download per group of frames
    pub fn hgi_deflate_grids_packed_dev(ctx: *mut HgiCtx, d_grids: *const c_void, width: u32, height: u32, batch: usize,
                                        frame_stride: usize, out: *mut u8, cap: usize, offsets: *mut usize,
                                        sizes: *mut usize) -> c_int;
    /// include/hgi.h: per-frame byte histogram of a grid batch on the device (d_hist: 256 * batch u64)
    pub fn hgi_histogram_u8_dev(ctx: *mut HgiCtx, d_grid: *const c_void, width: u32, height: u32, batch: usize,
                                frame_stride: usize, d_hist: *mut c_void) -> c_int;
    // ---- what a caller that owns device memory needs (examples/multi_device.rs: one thread + one ctx per GPU) ----
    /// pre-size the ctx's scratch for `batch` frames of this shape (nothing is allocated by later calls)
    pub fn hgi_ctx_reserve(ctx: *mut HgiCtx, width: u32, height: u32, levels: u32, batch: usize) -> c_int;
    pub fn hgi_ctx_scratch_bytes(ctx: *mut HgiCtx, bytes: *mut usize) -> c_int;
    /// `count` device planes of at least `bytes`, neighbours in different HBM classes (release with hgi_planes_free only)
    pub fn hgi_planes_alloc(ctx: *mut HgiCtx, bytes: usize, count: u32, planes: *mut *mut c_void, separated: *mut c_int) -> c_int;
    pub fn hgi_planes_free(ctx: *mut HgiCtx, count: u32, planes: *mut *mut c_void) -> c_int;
    /// one line on what the ctx's last hgi_planes_alloc found and did (owned by the ctx)
    pub fn hgi_planes_report(ctx: *mut HgiCtx) -> *const c_char;
    /// synthetic frames generated in place (kind 0 xy = benches/bench.rs:26-28, 1 noise, 2 ramp; frame f uses index first_frame + f)
    pub fn hgi_synth_u8_dev(ctx: *mut HgiCtx, kind: c_int, seed: u64, first_frame: u64, width: u32, height: u32,
                            d_out: *mut c_void, batch: usize, frame_stride: usize) -> c_int;
    /// per-frame [sum of squared differences, max |difference|, differing pixels] of before / after pairs (src/main.rs:84-92)
    pub fn hgi_diff_stats_dev(ctx: *mut HgiCtx, d_before: *const c_void, d_after: *const c_void, width: u32, height: u32,
                              batch: usize, frame_stride: usize, d_out: *mut c_void) -> c_int;
    pub fn hgi_timer_start(ctx: *mut HgiCtx) -> c_int;
    pub fn hgi_timer_stop(ctx: *mut HgiCtx, elapsed_ms: *mut f32) -> c_int;
}
