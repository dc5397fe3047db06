//! `hgi` -- the crate root surface of pl0q1n/RustyHGI (`src/lib.rs:13-23`) served by the MI355X library
//! `libhgi_hip.so` through the C ABI of `include/hgi.h`.
//!
//! STATUS: SOURCE ONLY.  The build image has no Rust toolchain and no crates.io access, so this crate has never
//! been compiled or run; what is verified on hardware is the C ABI underneath it (ctypes and C++ callers in
//! `tests/`).  It is written against the reference's dependency versions (`Cargo.toml` there: image 0.19,
//! serde 1.0, bincode 1.0, byteorder 1.2, flate2 1.0) and its 2015-edition idioms.
//!
//! Surface, item by item (reference file:line -> here):
//!   * `pub mod interpolator` -- `InterpolationType` (src/interpolator.rs:4-9), `trait Interpolator` with the
//!     reference's `interpolate(&self, levels, level, at, &GrayImage) -> u8` (:11-13), `LeftTop` (:15-28),
//!     `Crossed` (:30-91).  `interpolate` keeps the reference's host semantics: it is the trait surface, never
//!     called on the hot path.  The device predictor is chosen by the extra provided method `kernel_id()`.
//!   * `pub mod quantizator` -- `QuantizationLevel` (src/quantizator.rs:1-9; the reference generates `FromStr`,
//!     `Display` and `variants()` with clap's `arg_enum!`, written out here so that src/options.rs:58-63 resolves
//!     without clap), `trait Quantizator` (:12-15), `NoOp` (:17-34), `Linear` (:36-74).
//!   * `pub use {Archive, Metadata}` (src/archive.rs:15-28) with `serialize_to_writer` / `deserialize_from_reader`
//!     (:31-56): same wire format (magic 0xBAADA555 LE, bincode metadata, raw-DEFLATE bincode grid).
//!   * `pub use {Encoder, Decoder}` (src/encoder.rs:18,39; src/decoder.rs:14,18), `Grid` (src/grid.rs:2-27; like
//!     the reference, public in signatures but not re-exported by name... it IS exported here, which is a superset).
//!
//! `benches/bench.rs` of the reference (`:9-11`: `hgi::interpolator::{self, Crossed, InterpolationType}`,
//! `hgi::quantizator::{self, Linear, QuantizationLevel}`, `hgi::{Archive, Decoder, Encoder, Metadata}`;
//! `bincode::serialized_size(&archive)` at `:119`) resolves against this file with ZERO unresolved imports.
//! The `#[cfg(test)]` module of the reference's src/lib.rs (`:33-34`: `Quantizator` for `.error()`) resolves too.
//!
//! Reference call sites that still would NOT compile against this crate unchanged:
//!   * `src/main.rs:22-29` declares `mod archive; mod decoder; ...` itself instead of using the lib crate; the
//!     binary needs those eight lines replaced by `extern crate hgi; use hgi::{...}` (its function bodies,
//!     `src/main.rs:41-128`, then compile as they are: `Grid` is exported for `Archive::<Grid>`, `:65`).
//!   * a user-defined `impl Interpolator` compiles but has no device predictor: `Encoder::encode` panics for it
//!     (`try_encode` returns `HgiError::Unsupported`).  There is deliberately no host fallback.
//!
//! Differences in behaviour, all at the edges: `encode`/`decode` are infallible in the reference; here a device
//! failure has no return channel in those signatures and panics with the library's message -- `try_encode` /
//! `try_decode` return it instead.  `Grid::new` zero-fills (the reference leaves the vector uninitialised,
//! src/grid.rs:10-11; every cell is overwritten by `encode` either way).
extern crate bincode;
extern crate byteorder;
extern crate flate2;
extern crate image;
extern crate serde;
#[macro_use]
extern crate serde_derive;

pub mod ffi;      // (public: examples/multi_device.rs drives the device entry points directly)

use image::GrayImage;
use std::cell::RefCell;
use std::error::Error;
use std::ffi::CStr;
use std::fmt;
use std::io::{Read, Write};
use std::rc::Rc;

// ---------------------------------------------------------------------------------------------------------------
// errors and the device context
// ---------------------------------------------------------------------------------------------------------------
/// What the C ABI reports (`hgi_status`, include/hgi.h) plus the argument checks made on this side of it.
#[derive(Debug, Clone, PartialEq, Eq)]
pub enum HgiError {
    Invalid(String),
    NoMemory(String),
    Device(String),
    Unsupported(String),
}

impl fmt::Display for HgiError {
    fn fmt(&self, f: &mut fmt::Formatter) -> fmt::Result {
        match *self {
            HgiError::Invalid(ref m) => write!(f, "hgi: invalid argument: {}", m),
            HgiError::NoMemory(ref m) => write!(f, "hgi: out of memory: {}", m),
            HgiError::Device(ref m) => write!(f, "hgi: device error: {}", m),
            HgiError::Unsupported(ref m) => write!(f, "hgi: unsupported: {}", m),
        }
    }
}

impl Error for HgiError {
    fn description(&self) -> &str {
        "hgi device library error"
    }
}

fn status(code: i32) -> Result<(), HgiError> {
    if code == ffi::HGI_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(ffi::hgi_last_error()) }.to_string_lossy().into_owned();
    Err(match code {
        ffi::HGI_EINVAL => HgiError::Invalid(msg),
        ffi::HGI_ENOMEM => HgiError::NoMemory(msg),
        ffi::HGI_EUNSUPPORTED => HgiError::Unsupported(msg),
        _ => HgiError::Device(msg),
    })
}

/// `hgi_ctx` (device, stream, scratch).  Not thread-safe by contract (include/hgi.h), hence `Rc`, not `Arc`:
/// every `Encoder` / `Decoder` of a thread shares ONE context, created on first use, so constructing codec
/// objects is as cheap as in the reference.
struct Ctx(*mut ffi::HgiCtx);

impl Drop for Ctx {
    fn drop(&mut self) {
        unsafe { ffi::hgi_ctx_destroy(self.0) }
    }
}

thread_local! {
    static CTX: RefCell<Option<Rc<Ctx>>> = RefCell::new(None);
}

fn thread_ctx() -> Result<Rc<Ctx>, HgiError> {
    CTX.with(|slot| {
        let mut slot = slot.borrow_mut();
        if let Some(ref ctx) = *slot {
            return Ok(ctx.clone());
        }
        let mut raw = std::ptr::null_mut();
        status(unsafe { ffi::hgi_ctx_create(0, &mut raw) })?;
        let ctx = Rc::new(Ctx(raw));
        *slot = Some(ctx.clone());
        Ok(ctx)
    })
}

const MAX_LEVELS: usize = 31; // shifts on u32, src/utils.rs:17

// ---------------------------------------------------------------------------------------------------------------
// Grid -- src/grid.rs:2-27
// ---------------------------------------------------------------------------------------------------------------
/// Residual plane with the geometry of the image: `(column, line)` lives at `line * width + column`.
/// Field order matters: bincode writes `buffer` (u64 length + bytes) then `width` (u64), as the reference does.
#[derive(Serialize, Deserialize, PartialEq, Eq, Debug)]
pub struct Grid {
    buffer: Vec<u8>,
    width: usize,
}

impl Grid {
    pub fn new(width: usize, height: usize) -> Self {
        Grid { buffer: vec![0; width * height], width: width }
    }

    #[inline(always)]
    pub unsafe fn set(&mut self, (column, line): (u32, u32), value: u8) {
        let at = line as usize * self.width + column as usize;
        *self.buffer.get_unchecked_mut(at) = value;
    }

    #[inline(always)]
    pub unsafe fn get(&self, column: u32, line: u32) -> u8 {
        let at = line as usize * self.width + column as usize;
        *self.buffer.get_unchecked(at)
    }

    /// src/grid.rs:29-33 (debug dump, one row per line)
    pub fn print(&self) {
        for row in self.buffer.chunks(self.width.max(1)) {
            println!("{:3?}", row);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// quantizator -- src/quantizator.rs
// ---------------------------------------------------------------------------------------------------------------
pub mod quantizator {
    use super::ffi;
    use std::fmt;
    use std::str::FromStr;

    /// src/quantizator.rs:1-9.  Variant order is the bincode tag (0..3) and the `level` argument of
    /// `hgi_linear_lut`.
    #[derive(Clone, Copy, Serialize, Deserialize, Debug, PartialEq, Eq)]
    pub enum QuantizationLevel {
        Lossless,
        Low,
        Medium,
        High,
    }

    impl QuantizationLevel {
        /// what clap's `arg_enum!` generates in the reference; used by src/options.rs:60
        pub fn variants() -> [&'static str; 4] {
            ["Lossless", "Low", "Medium", "High"]
        }
    }

    impl FromStr for QuantizationLevel {
        type Err = String;
        /// ASCII case-insensitive, like `arg_enum!`; not typo-tolerant ("loseless" is an error there too)
        fn from_str(s: &str) -> Result<Self, Self::Err> {
            match s.to_ascii_lowercase().as_str() {
                "lossless" => Ok(QuantizationLevel::Lossless),
                "low" => Ok(QuantizationLevel::Low),
                "medium" => Ok(QuantizationLevel::Medium),
                "high" => Ok(QuantizationLevel::High),
                _ => Err(format!("valid values: {}", Self::variants().join(", "))),
            }
        }
    }

    impl fmt::Display for QuantizationLevel {
        fn fmt(&self, f: &mut fmt::Formatter) -> fmt::Result {
            f.write_str(Self::variants()[*self as usize])
        }
    }

    /// src/quantizator.rs:12-15.  `table()` is the one addition: any quantizer is a pure u8 -> u8 map, and
    /// that map, tabulated, is what crosses to the device.
    pub trait Quantizator: From<QuantizationLevel> {
        fn quantize(&self, value: u8) -> u8;
        fn error(&self) -> u8;

        fn table(&self) -> [u8; 256] {
            let mut t = [0u8; 256];
            for (value, slot) in t.iter_mut().enumerate() {
                *slot = self.quantize(value as u8);
            }
            t
        }
    }

    /// src/quantizator.rs:17-34
    pub struct NoOp;

    impl From<QuantizationLevel> for NoOp {
        fn from(_: QuantizationLevel) -> Self {
            NoOp
        }
    }

    impl Quantizator for NoOp {
        #[inline(always)]
        fn quantize(&self, value: u8) -> u8 {
            value
        }

        #[inline(always)]
        fn error(&self) -> u8 {
            0
        }
    }

    /// src/quantizator.rs:36-74.  The table comes from `hgi_linear_lut`, which restates `:41-63` in the library
    /// (one definition for the C++, Python and Rust hosts); it needs no device.
    pub struct Linear {
        table: [u8; 256],
        error: u8,
    }

    impl From<QuantizationLevel> for Linear {
        fn from(level: QuantizationLevel) -> Self {
            let mut table = [0u8; 256];
            let mut error = 0u8;
            let rc = unsafe { ffi::hgi_linear_lut(level as i32, table.as_mut_ptr(), &mut error) };
            assert_eq!(rc, ffi::HGI_OK, "hgi_linear_lut rejects only levels outside 0..=3");
            Linear { table: table, error: error }
        }
    }

    impl Quantizator for Linear {
        #[inline(always)]
        fn quantize(&self, value: u8) -> u8 {
            self.table[value as usize]
        }

        #[inline(always)]
        fn error(&self) -> u8 {
            self.error
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// interpolator -- src/interpolator.rs
// ---------------------------------------------------------------------------------------------------------------
pub mod interpolator {
    use super::ffi;
    use image::GrayImage;

    /// src/interpolator.rs:4-9: metadata tag only (bincode tag 0..2); `Line` / `Previous` have no implementation
    /// in the reference either.
    #[derive(Clone, Serialize, Deserialize, Debug, PartialEq, Eq)]
    pub enum InterpolationType {
        Crossed,
        Line,
        Previous,
    }

    /// src/interpolator.rs:11-13, plus the device selector.  `level` is 1-based (callers pass `l + 1`).
    pub trait Interpolator {
        fn interpolate(&self, levels: usize, level: usize, at: (u32, u32), image: &GrayImage) -> u8;

        /// Which predictor of `libhgi_hip.so` computes exactly `interpolate`; `None` = no device predictor
        /// (the codec then reports `HgiError::Unsupported`; there is no host fallback).
        fn kernel_id(&self) -> Option<i32> {
            None
        }
    }

    fn cell_origin(levels: usize, level: usize, (x, y): (u32, u32)) -> (u32, u32, u32) {
        let step = 1u32 << (levels - level + 1);
        (x & !(step - 1), y & !(step - 1), step)
    }

    /// src/interpolator.rs:15-28: the pixel at the origin of the enclosing step-cell.
    pub struct LeftTop;

    impl Interpolator for LeftTop {
        fn interpolate(&self, levels: usize, level: usize, at: (u32, u32), image: &GrayImage) -> u8 {
            let (x0, y0, _) = cell_origin(levels, level, at);
            image.get_pixel(x0, y0).data[0]
        }

        fn kernel_id(&self) -> Option<i32> {
            Some(ffi::HGI_INTERP_LEFTTOP)
        }
    }

    /// src/interpolator.rs:30-91: the four sides of the enclosing step-cell, each averaged with round-half-up,
    /// then the mean of the four averages rounded down; corners outside the image count as 0.
    pub struct Crossed;

    impl Interpolator for Crossed {
        fn interpolate(&self, levels: usize, level: usize, at: (u32, u32), image: &GrayImage) -> u8 {
            let (x0, y0, step) = cell_origin(levels, level, at);
            let (w, h) = image.dimensions();
            let corner = |x: u32, y: u32| -> u32 {
                if x < w && y < h {
                    u32::from(image.get_pixel(x, y).data[0])
                } else {
                    0
                }
            };
            let half_up = |a: u32, b: u32| (a + b + 1) >> 1;
            let (c00, c01) = (corner(x0, y0), corner(x0, y0 + step));
            let (c10, c11) = (corner(x0 + step, y0), corner(x0 + step, y0 + step));
            let sides = half_up(c00, c10) + half_up(c11, c01) + half_up(c01, c00) + half_up(c11, c10);
            (sides >> 2) as u8
        }

        fn kernel_id(&self) -> Option<i32> {
            Some(ffi::HGI_INTERP_CROSSED)
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// archive -- src/archive.rs
// ---------------------------------------------------------------------------------------------------------------
mod archive {
    use bincode;
    use byteorder::{ReadBytesExt, WriteBytesExt, LE};
    use flate2::read::DeflateDecoder;
    use flate2::write::DeflateEncoder;
    use flate2::Compression;
    use interpolator::InterpolationType;
    use quantizator::QuantizationLevel;
    use serde::de::DeserializeOwned;
    use serde::Serialize;
    use std::error::Error;
    use std::io::{Read, Write};

    /// first four bytes of a `.hgi` file, little-endian (src/archive.rs:13)
    const MAGIC: u32 = 0xBAAD_A555;

    /// src/archive.rs:15-22.  bincode 1.x defaults: enum tags as u32, `usize` as u64 -> 24 bytes.
    #[derive(Clone, Serialize, Deserialize, Debug, PartialEq, Eq)]
    pub struct Metadata {
        pub quantization_level: QuantizationLevel,
        pub interpolation: InterpolationType,
        pub width: u32,
        pub height: u32,
        pub scale_level: usize,
    }

    /// src/archive.rs:24-28
    #[derive(Serialize, Deserialize, Debug, PartialEq, Eq)]
    pub struct Archive<G> {
        pub metadata: Metadata,
        pub grid: G,
    }

    impl<G: Serialize + DeserializeOwned> Archive<G> {
        /// src/archive.rs:31-41: magic, metadata in the clear, then the bincode image of the grid through raw
        /// DEFLATE at the best level.  (A host-side entropy stage: the device hands back the grid)
        pub fn serialize_to_writer<W: Write>(&self, mut w: &mut W) -> Result<(), Box<Error>> {
            w.write_u32::<LE>(MAGIC)?;
            bincode::serialize_into(&mut w, &self.metadata)?;
            let plain = bincode::serialize(&self.grid)?;
            let mut deflate = DeflateEncoder::new(Vec::with_capacity(plain.len() / 2), Compression::best());
            deflate.write_all(&plain)?;
            w.write_all(&deflate.finish()?)?;
            Ok(())
        }

        /// src/archive.rs:43-56
        pub fn deserialize_from_reader<R: Read>(mut r: &mut R) -> Result<Self, Box<Error>>
        where
            Archive<G>: 'static,
        {
            if r.read_u32::<LE>()? != MAGIC {
                return Err("incorrect magic number".into());
            }
            let metadata: Metadata = bincode::deserialize_from(&mut r)?;
            let grid: G = bincode::deserialize_from(DeflateDecoder::new(r))?;
            Ok(Archive { metadata: metadata, grid: grid })
        }
    }
}

pub use archive::{Archive, Metadata};

impl Archive<Grid> {
    /// `serialize_to_writer` with the DEFLATE stage on the device (`hgi_deflate_grid`, include/hgi.h): the same
    /// container -- magic, bincode metadata, raw DEFLATE of the grid's bincode image -- whose stream is one
    /// dynamic-Huffman block of literals and run matches written by the GPU instead of flate2 at `Compression::best()`
    /// (src/archive.rs:34-40).  `deserialize_from_reader` reads it like any other archive.  No reference counterpart.
    pub fn serialize_to_writer_device<W: Write>(&self, w: &mut W) -> Result<(), Box<Error>> {
        use byteorder::{WriteBytesExt, LE};
        w.write_u32::<LE>(0xBAAD_A555)?;
        bincode::serialize_into(&mut *w, &self.metadata)?;
        let (width, n) = (self.grid.width, self.grid.buffer.len());
        let height = if width == 0 { 0 } else { n / width };
        let mut out = vec![0u8; n + n / 8 + 1024];
        let mut bytes = 0usize;
        let ctx = thread_ctx()?;
        status(unsafe {
            ffi::hgi_deflate_grid(ctx.0, self.grid.buffer.as_ptr(), width as u32, height as u32, out.as_mut_ptr(), out.len(),
                                  &mut bytes)
        })?;
        w.write_all(&out[..bytes])?;
        Ok(())
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Encoder / Decoder -- src/encoder.rs, src/decoder.rs
// ---------------------------------------------------------------------------------------------------------------
use interpolator::Interpolator;
use quantizator::Quantizator;

fn predictor<I: Interpolator>(interpolator: &I) -> Result<i32, HgiError> {
    interpolator
        .kernel_id()
        .ok_or_else(|| HgiError::Unsupported("this Interpolator has no device predictor (LeftTop and Crossed do)".into()))
}

fn check_levels(levels: usize) -> Result<u32, HgiError> {
    if levels > MAX_LEVELS {
        return Err(HgiError::Invalid(format!("levels {} out of range 0..={}", levels, MAX_LEVELS)));
    }
    Ok(levels as u32)
}

/// src/encoder.rs:7-11
pub struct Encoder<I, Q> {
    interpolator: I,
    quantizator: Q,
    table: [u8; 256],
    scale_level: usize,
}

impl<I: Interpolator, Q: Quantizator> Encoder<I, Q> {
    /// src/encoder.rs:18.  Tabulates the quantizer once; touches no device.
    pub fn new(interpolator: I, quantizator: Q, scale_level: usize) -> Self {
        let table = quantizator.table();
        Encoder { interpolator: interpolator, quantizator: quantizator, table: table, scale_level: scale_level }
    }

    /// The quantizer this encoder was built with (its `error()` is the reconstruction bound).
    pub fn quantizator(&self) -> &Q {
        &self.quantizator
    }

    /// src/encoder.rs:39: consumes the image, returns the residual grid.  Panics if the device call fails
    /// (the signature has no error channel); `try_encode` is the fallible form.
    pub fn encode(&mut self, input: GrayImage) -> Grid {
        match self.try_encode(&input) {
            Ok(grid) => grid,
            Err(e) => panic!("{}", e),
        }
    }

    pub fn try_encode(&mut self, input: &GrayImage) -> Result<Grid, HgiError> {
        let kernel = predictor(&self.interpolator)?;
        let levels = check_levels(self.scale_level)?;
        let (width, height) = input.dimensions();
        let mut grid = Grid::new(width as usize, height as usize);
        if grid.buffer.is_empty() {
            return Ok(grid);
        }
        let ctx = thread_ctx()?;
        let pixels: &[u8] = input; // GrayImage derefs to its packed row-major bytes, stride == width
        debug_assert_eq!(pixels.len(), grid.buffer.len());
        status(unsafe {
            ffi::hgi_encode_u8(ctx.0, pixels.as_ptr(), width, height, levels, kernel, self.table.as_ptr(),
                               grid.buffer.as_mut_ptr())
        })?;
        Ok(grid)
    }
}

/// src/decoder.rs:6-8
pub struct Decoder<I> {
    interpolator: I,
}

impl<I: Interpolator> Decoder<I> {
    /// src/decoder.rs:14
    pub fn new(interpolator: I) -> Self {
        Decoder { interpolator: interpolator }
    }

    /// src/decoder.rs:18.  Panics on a device failure or a grid whose size is not `width * height`;
    /// `try_decode` is the fallible form.
    pub fn decode(&mut self, dimensions: (u32, u32), levels: usize, grid: &Grid) -> GrayImage {
        match self.try_decode(dimensions, levels, grid) {
            Ok(image) => image,
            Err(e) => panic!("{}", e),
        }
    }

    pub fn try_decode(&mut self, (width, height): (u32, u32), levels: usize, grid: &Grid) -> Result<GrayImage, HgiError> {
        let kernel = predictor(&self.interpolator)?;
        let levels = check_levels(levels)?;
        let pixels = width as usize * height as usize;
        // the library reads width * height bytes from the pointer it is given: never hand it a shorter buffer
        if grid.buffer.len() != pixels || (pixels != 0 && grid.width != width as usize) {
            return Err(HgiError::Invalid(format!("grid holds {} bytes at width {}, expected {}x{}", grid.buffer.len(),
                                                 grid.width, width, height)));
        }
        let mut image = GrayImage::new(width, height);
        if pixels == 0 {
            return Ok(image);
        }
        let ctx = thread_ctx()?;
        status(unsafe {
            ffi::hgi_decode_u8(ctx.0, grid.buffer.as_ptr(), width, height, levels, kernel, image.as_mut_ptr())
        })?;
        Ok(image)
    }
}

/// Drain the device work of this thread's context (the host-pointer calls above are synchronous already; this
/// exists for callers that mix in the `_dev` entry points of `ffi`).
pub fn sync() -> Result<(), HgiError> {
    let ctx = thread_ctx()?;
    status(unsafe { ffi::hgi_sync(ctx.0) })
}

#[allow(dead_code)]
fn _surface_check<R: Read, W: Write>(_: R, _: W) {}
