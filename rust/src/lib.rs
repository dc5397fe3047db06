//! `hgi` -- the reference crate's codec surface (src/lib.rs:16-23 of pl0q1n/RustyHGI) served by the
//! MI355X library.  `benches/bench.rs` uses exactly: `hgi::interpolator::{self, Crossed,
//! InterpolationType}`, `interpolator::LeftTop`, `hgi::quantizator::{self, Linear, QuantizationLevel}`,
//! `quantizator::NoOp`, `hgi::{Decoder, Encoder}` -- all present here with the same signatures.
//! (`Archive`/`Metadata` are SURVEY.md 8(f1), not part of this shim yet.)
//!
//! SOURCE ONLY: never compiled (no Rust toolchain in the build image).
extern crate image;
#[macro_use]
extern crate serde_derive;

mod ffi;

use image::GrayImage;
use std::ffi::CStr;

fn check(status: i32) {
    if status != ffi::HGI_OK {
        let msg = unsafe { CStr::from_ptr(ffi::hgi_last_error()) }.to_string_lossy().into_owned();
        // the reference's encode/decode are infallible; a device failure has nowhere to go but a panic
        panic!("hgi: {}", msg);
    }
}

struct Ctx(*mut ffi::HgiCtx);
impl Ctx {
    fn new() -> Self {
        let mut p = std::ptr::null_mut();
        check(unsafe { ffi::hgi_ctx_create(0, &mut p) });
        Ctx(p)
    }
}
impl Drop for Ctx {
    fn drop(&mut self) {
        unsafe { ffi::hgi_ctx_destroy(self.0) }
    }
}

/// src/grid.rs:2-27
#[derive(Serialize, Deserialize, PartialEq, Eq, Debug)]
pub struct Grid {
    buffer: Vec<u8>,
    width: usize,
}

impl Grid {
    pub fn new(width: usize, height: usize) -> Self {
        Grid { buffer: vec![0; width * height], width }
    }
    #[inline(always)]
    pub unsafe fn set(&mut self, (column, line): (u32, u32), value: u8) {
        *self.buffer.get_unchecked_mut(line as usize * self.width + column as usize) = value;
    }
    #[inline(always)]
    pub unsafe fn get(&self, column: u32, line: u32) -> u8 {
        *self.buffer.get_unchecked(line as usize * self.width + column as usize)
    }
}

pub mod quantizator {
    use super::{check, ffi};

    /// src/quantizator.rs:1-9
    #[derive(Clone, Copy, Serialize, Deserialize, Debug, PartialEq, Eq)]
    pub enum QuantizationLevel {
        Lossless,
        Low,
        Medium,
        High,
    }

    /// src/quantizator.rs:12-15
    pub trait Quantizator: From<QuantizationLevel> {
        fn quantize(&self, value: u8) -> u8;
        fn error(&self) -> u8;
        /// Any quantizer crosses to the device as its 256-entry table.
        fn table(&self) -> [u8; 256] {
            let mut t = [0u8; 256];
            for (i, e) in t.iter_mut().enumerate() {
                *e = self.quantize(i as u8);
            }
            t
        }
    }

    pub struct NoOp;
    impl From<QuantizationLevel> for NoOp {
        fn from(_: QuantizationLevel) -> Self {
            NoOp
        }
    }
    impl Quantizator for NoOp {
        fn quantize(&self, value: u8) -> u8 {
            value
        }
        fn error(&self) -> u8 {
            0
        }
    }

    pub struct Linear {
        table: [u8; 256],
        error: u8,
    }
    impl From<QuantizationLevel> for Linear {
        fn from(level: QuantizationLevel) -> Self {
            let mut table = [0u8; 256];
            let mut error = 0u8;
            check(unsafe { ffi::hgi_linear_lut(level as i32, table.as_mut_ptr(), &mut error) });
            Linear { table, error }
        }
    }
    impl Quantizator for Linear {
        fn quantize(&self, value: u8) -> u8 {
            self.table[value as usize]
        }
        fn error(&self) -> u8 {
            self.error
        }
    }
}

pub mod interpolator {
    /// src/interpolator.rs:4-9
    #[derive(Clone, Serialize, Deserialize, Debug, PartialEq, Eq)]
    pub enum InterpolationType {
        Crossed,
        Line,
        Previous,
    }

    /// The reference's per-pixel `interpolate` cannot cross to a GPU; the zero-sized interpolator
    /// types select a device predictor instead (sealed: only the two the reference implements).
    pub trait Interpolator: private::Sealed {
        const KERNEL_ID: i32;
    }
    pub struct LeftTop;
    pub struct Crossed;
    impl Interpolator for LeftTop {
        const KERNEL_ID: i32 = super::ffi::HGI_INTERP_LEFTTOP;
    }
    impl Interpolator for Crossed {
        const KERNEL_ID: i32 = super::ffi::HGI_INTERP_CROSSED;
    }
    mod private {
        pub trait Sealed {}
        impl Sealed for super::LeftTop {}
        impl Sealed for super::Crossed {}
    }
}

use interpolator::Interpolator;
use quantizator::Quantizator;

/// src/encoder.rs:7-11
pub struct Encoder<I, Q> {
    #[allow(dead_code)]
    interpolator: I,
    table: [u8; 256],
    scale_level: usize,
    ctx: Ctx,
    _q: std::marker::PhantomData<Q>,
}

impl<I: Interpolator, Q: Quantizator> Encoder<I, Q> {
    /// src/encoder.rs:18
    pub fn new(interpolator: I, quantizator: Q, scale_level: usize) -> Self {
        Encoder { interpolator, table: quantizator.table(), scale_level, ctx: Ctx::new(), _q: std::marker::PhantomData }
    }

    /// src/encoder.rs:39 -- consumes the image, returns the residual grid.
    pub fn encode(&mut self, input: GrayImage) -> Grid {
        let (width, height) = input.dimensions();
        let mut grid = Grid::new(width as usize, height as usize);
        check(unsafe {
            ffi::hgi_encode_u8(self.ctx.0, input.as_ptr(), width, height, self.scale_level as u32, I::KERNEL_ID,
                               self.table.as_ptr(), grid.buffer.as_mut_ptr())
        });
        grid
    }
}

/// src/decoder.rs:6-8
pub struct Decoder<I> {
    #[allow(dead_code)]
    interpolator: I,
    ctx: Ctx,
}

impl<I: Interpolator> Decoder<I> {
    /// src/decoder.rs:14
    pub fn new(interpolator: I) -> Self {
        Decoder { interpolator, ctx: Ctx::new() }
    }

    /// src/decoder.rs:18
    pub fn decode(&mut self, (width, height): (u32, u32), levels: usize, grid: &Grid) -> GrayImage {
        let mut image = GrayImage::new(width, height);
        check(unsafe {
            ffi::hgi_decode_u8(self.ctx.0, grid.buffer.as_ptr(), width, height, levels as u32, I::KERNEL_ID,
                               image.as_mut_ptr())
        });
        image
    }
}
