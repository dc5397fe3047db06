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
}
