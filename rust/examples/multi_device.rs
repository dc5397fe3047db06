//! One process, N GPUs, through the C ABI: the Rust form of `bench_cpp --devices N` (benches/bench.cpp).
//!
//! The reference codes one frame per call on one thread (`Encoder::encode`, src/encoder.rs:39); a batch of independent frames
//! therefore shards by frame with no exchange between the shards.  Here every device gets one host thread and one `hgi_ctx`
//! ("a ctx is not thread-safe; distinct ctxs are independent", include/hgi.h), generates its shard of the batch in place,
//! places its three planes (image -> grid -> image) and codes it; the threads meet at a barrier around the timed steps.
//! SOURCE ONLY: this image has no Rust toolchain; the C++ twin is compiled and run by the GPU suite.
//!
//!     cargo run --release --example multi_device -- 8        # HGI_HIP_DIR=<dir of libhgi_hip.so>
use std::ffi::CStr;
use std::os::raw::{c_int, c_void};
use std::ptr;
use std::sync::{Arc, Barrier};
use std::time::Instant;

use hgi::ffi;

const S: u32 = 4096; // BASELINE configs[3]: 512 frames of 4096 x 4096, level 4, Medium
const LEVELS: u32 = 4;
const GLOBAL_FRAMES: usize = 512;

fn check(status: c_int, what: &str) {
    if status != ffi::HGI_OK {
        let msg = unsafe { CStr::from_ptr(ffi::hgi_last_error()) }.to_string_lossy().into_owned();
        panic!("{}: {}", what, msg);
    }
}

/// contiguous blocks that differ by at most one frame (rustyhgi_amd/batch.py: shard())
fn shard(global: usize, world: usize, rank: usize) -> (usize, usize) {
    let (base, extra) = (global / world, global % world);
    (rank * base + rank.min(extra), base + usize::from(rank < extra))
}

fn main() {
    let devices: usize = std::env::args().nth(1).map(|a| a.parse().expect("device count")).unwrap_or(1);
    let steps = 20;
    let barrier = Arc::new(Barrier::new(devices));
    let workers: Vec<_> = (0..devices)
        .map(|d| {
            let barrier = Arc::clone(&barrier);
            std::thread::spawn(move || unsafe {
                let (first, frames) = shard(GLOBAL_FRAMES, devices, d);
                let n = (S as usize) * (S as usize);
                let mut ctx: *mut ffi::HgiCtx = ptr::null_mut();
                check(ffi::hgi_ctx_create(d as c_int, &mut ctx), "hgi_ctx_create");
                let (mut lut, mut err) = ([0u8; 256], 0u8);
                check(ffi::hgi_linear_lut(2, lut.as_mut_ptr(), &mut err), "hgi_linear_lut"); // QuantizationLevel::Medium
                check(ffi::hgi_ctx_reserve(ctx, S, S, LEVELS, frames), "hgi_ctx_reserve");
                let mut planes: [*mut c_void; 3] = [ptr::null_mut(); 3];
                let mut separated: c_int = 0;
                check(ffi::hgi_planes_alloc(ctx, frames * n, 3, planes.as_mut_ptr(), &mut separated), "hgi_planes_alloc");
                let placed = std::ffi::CStr::from_ptr(ffi::hgi_planes_report(ctx)).to_string_lossy().into_owned();
                eprintln!("device {}: {}", d, placed);
                check(ffi::hgi_synth_u8_dev(ctx, 2, 0x4847_4930 + 3, first as u64, S, S, planes[0], frames, n), "hgi_synth_u8_dev");
                let step = || {
                    check(ffi::hgi_encode_u8_dev(ctx, planes[0], S, S, LEVELS, ffi::HGI_INTERP_CROSSED, lut.as_ptr(), planes[1], frames, n), "encode");
                    check(ffi::hgi_decode_u8_dev(ctx, planes[1], S, S, LEVELS, ffi::HGI_INTERP_CROSSED, planes[2], frames, n), "decode");
                };
                for _ in 0..40 {
                    step(); // the device's clocks ramp for ~25 ms after idle: untimed steps first
                }
                check(ffi::hgi_sync(ctx), "hgi_sync");
                barrier.wait();
                let t0 = Instant::now();
                for _ in 0..steps {
                    step();
                }
                check(ffi::hgi_sync(ctx), "hgi_sync");
                barrier.wait();
                let wall = t0.elapsed().as_secs_f64();
                check(ffi::hgi_planes_free(ctx, 3, planes.as_mut_ptr()), "hgi_planes_free");
                ffi::hgi_ctx_destroy(ctx);
                (frames, separated != 0, wall)
            })
        })
        .collect();
    let results: Vec<_> = workers.into_iter().map(|w| w.join().expect("device thread")).collect();
    let wall = results.iter().map(|r| r.2).fold(0.0, f64::max);
    let px = (GLOBAL_FRAMES as f64) * (S as f64) * (S as f64) * steps as f64;
    for (d, (frames, separated, _)) in results.iter().enumerate() {
        println!("device {}: {} frames, planes {}", d, frames, if *separated { "separated" } else { "not separated" });
    }
    println!("{} devices: {:.1} Mpixels/s encode+decode", devices, px / wall / 1e6);
}
