#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of HGI encode+decode on 4096x4096 u8 frames, level=4, Medium.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One step = one pass of the hot path over this rank's batch: encode `frames` frames, then decode
them (BASELINE.json config C3, sharded by frame; frames are generated in place on each GPU from the
global frame index, so no pixel ever crosses xGMI).  Inputs are resident in HBM before the timed
region.  RCCL is used only where the batch split needs it: broadcast of the quantizer table and
all-gather of per-rank checksums.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import rustyhgi_amd as H                      # noqa: E402
from rustyhgi_amd import _ffi, batch          # noqa: E402
from rustyhgi_amd.interpolator import Crossed  # noqa: E402
from rustyhgi_amd.quantizator import Linear, QuantizationLevel, Quantizator  # noqa: E402

SEED0 = 0x48474930
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (guide: MI355X_MICROARCH.md, chip-level parameters)


class TableQuantizator(Quantizator):
    """A quantizer received as its 256-entry table (what rank 0 broadcasts)."""

    def __init__(self, table, error):
        self._t, self._e = np.ascontiguousarray(table, np.uint8), int(error)

    def quantize(self, value):
        return int(self._t[value & 0xFF])

    def error(self):
        return self._e

    def table(self):
        return self._t.copy()


def pmc_traffic(kernel, frames, size, levels):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE with the
    gfx950 x2 correction + WRITE_SIZE; tools/profile.sh -> profiles/*_traffic.json), if that profile was
    taken on this exact workload; else None.  bench.py cannot run the profiler on itself."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        with open(path) as f:
            prof = json.load(f)
        if prof["workload"] != {"frames": frames, "size": size, "levels": levels}:
            return None
        return prof["kernels"][kernel]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(args, lut, gpu_check):
    """The oracle ('port' of the reference: scalar, one thread per image) timed on this host's cores
    over a bounded sample of the same workload.  Also re-checks the GPU result on those frames."""
    from oracle import hgi_oracle as O
    cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))
    frames = max(threads * 2, 8)
    imgs = np.stack([O.synth(O.SYNTH_RAMP, SEED0 + 3, f, args.size, args.size) for f in range(2)])
    imgs = np.concatenate([imgs] * ((frames + 1) // 2))[:frames]
    r = O.bench_batch(imgs, args.levels, lut, threads)
    for f in (0, 1):
        assert (r["grids"][f] == gpu_check["grid"][f]).all(), "GPU encode differs from the oracle"
        assert (r["outs"][f] == gpu_check["out"][f]).all(), "GPU decode differs from the oracle"
    px = frames * args.size * args.size
    return {"value": round(px / r["wall_s"] / 1e6, 1), "unit": "Mpixels/s", "cores": threads, "kind": "port",
            "sample": "%d frames %dx%d L%d %s, encode+decode, one oracle thread per frame, %.2f s wall"
                      % (frames, args.size, args.size, args.levels, args.quant, r["wall_s"]),
            "one_thread_mpix_s": round(px / (r["enc_cpu_s"] + r["dec_cpu_s"]) / 1e6, 1)}


def main():
    # stdout carries exactly ONE line, the JSON result of rank 0.  Native libraries write there too (RCCL prints a
    # five-line version banner when a communicator is created), so file descriptor 1 points at stderr for the whole
    # run and is switched back only around that one print.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(line, flush=True)
        os.dup2(2, 1)

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=64, help="frames per GPU per step (C3: 512 / 8)")
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--quant", default="medium")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--xgmi-scatter", action="store_true",
                    help="also time the labelled variant where all frames start and end on GPU 0 (scatter, code, gather)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # HGI_BENCH_FORCE_DIST=1: take the RCCL code path (init, broadcast, barrier, all-reduce, all-gather) with one rank too
    if world > 1 or os.environ.get("HGI_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    # ---- batch split: rank 0 owns the parameters; table + levels go out by RCCL broadcast ----
    level = QuantizationLevel.parse(args.quant)
    if rank == 0:
        q = Linear.from_level(level)
        lut, err, levels = batch.broadcast_params(dist, dev, q.table(), q.error(), args.levels)
    else:
        lut, err, levels = batch.broadcast_params(dist, dev)

    # a real (non-null) stream made current for torch: the codec launches, the timing events and the
    # torch ops around them all live on it
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    ctx = H.Context(local)
    ctx.set_stream(stream.cuda_stream)
    F, S = args.frames, args.size
    ctx.reserve(S, S, levels, F)
    imgs = torch.empty((F, S, S), dtype=torch.uint8, device=dev)
    grids = torch.empty_like(imgs)
    outs = torch.empty_like(imgs)
    # frame f of rank r is global frame r*F + f of config C3 (ramp(3)): produced where it is used
    first, count = batch.shard(world * F, world, rank)
    assert count == F
    _ffi.check(_ffi.lib().hgi_synth_u8_dev(ctx.handle, _ffi.SYNTH_RAMP, SEED0 + 3, first, S, S,
                                           imgs.data_ptr(), F, S * S))
    enc = H.Encoder(Crossed(), TableQuantizator(lut, err), levels, context=ctx)
    dec = H.Decoder(Crossed(), context=ctx)

    def step():
        enc.encode_batch(imgs, out=grids)
        dec.decode_batch(grids, levels, out=outs)

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        enc.encode_batch(imgs, out=grids)
        ev[k][1].record()
        dec.decode_batch(grids, levels, out=outs)
        ev[k][2].record()
    fence()
    elapsed = batch.max_over_ranks(dist, time.perf_counter() - t0, dev)

    enc_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    dec_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))

    # same-run reference point: a plain 16-B/lane streaming copy of the same 2 x F frames of traffic
    copy_ms = None
    if rank == 0:
        ce = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        n = F * S * S
        for i in range(6):
            if i == 1:
                ce[0].record()
            _ffi.check(_ffi.lib().hgi_copy_u8_dev(ctx.handle, imgs.data_ptr(), outs.data_ptr(), n))
        ce[1].record()
        torch.cuda.synchronize(dev)
        copy_ms = ce[0].elapsed_time(ce[1]) / 5
        dec.decode_batch(grids, levels, out=outs)      # restore the decoded frames the checks below read
        torch.cuda.synchronize(dev)

    # ---- optional, separately labelled: every frame starts and ends on GPU 0.  Bound by the
    # source GPU's xGMI links, not by the codec; never part of `value`.
    xgmi = None
    if args.xgmi_scatter:
        allf = torch.empty((world * F, S, S), dtype=torch.uint8, device=dev) if rank == 0 else None
        allo = torch.empty_like(allf) if rank == 0 else None
        if rank == 0:
            _ffi.check(_ffi.lib().hgi_synth_u8_dev(ctx.handle, _ffi.SYNTH_RAMP, SEED0 + 3, 0, S, S,
                                                   allf.data_ptr(), world * F, S * S))
        mine_in = torch.empty_like(imgs)

        def xstep():
            batch.scatter_frames(dist, allf, mine_in)
            enc.encode_batch(mine_in, out=grids)
            dec.decode_batch(grids, levels, out=outs)
            batch.gather_frames(dist, outs, allo)

        xstep()
        fence()
        t1 = time.perf_counter()
        nx = max(1, min(args.steps, 5))
        for _ in range(nx):
            xstep()
        fence()
        xs = batch.max_over_ranks(dist, time.perf_counter() - t1, dev) / nx
        if rank == 0:
            same = bool(torch.equal(allo[:F], outs)) and int((allf[:F].to(torch.int16) - allo[:F].to(torch.int16)).abs().max()) <= err
            xgmi = {"ms_per_step": round(xs * 1e3, 4), "value": round(world * F * S * S / xs / 1e6, 1), "unit": "Mpixels/s",
                    "bytes_over_links_per_step": 2 * (world - 1) * F * S * S, "roundtrip_ok": same,
                    "note": "frames scattered from and gathered to GPU 0 (torch.distributed scatter/gather over RCCL); "
                            "per-link bound, reported beside the sharded number, never as it"}
        del allf, allo, mine_in
        dec.decode_batch(grids, levels, out=outs)
        torch.cuda.synchronize(dev)

    # ---- per-rank checks + stats gather (RCCL all-gather) ----
    stats = torch.zeros(3 * F, dtype=torch.int64, device=dev)
    _ffi.check(_ffi.lib().hgi_diff_stats_dev(ctx.handle, imgs.data_ptr(), outs.data_ptr(), S, S, F, S * S,
                                             stats.data_ptr()))
    st = stats.view(F, 3)
    mine = torch.stack([st[:, 0].sum(), st[:, 1].max(), grids.view(-1)[::4099].to(torch.int64).sum()])
    allst = batch.gather_stats(dist, mine)
    if not os.environ.get("HGI_BENCH_NOCHECK"):   # timing-only experiments produce wrong pixels
        assert int(allst[:, 1].max()) <= err, "reconstruction error exceeds the quantizer bound"

    if rank == 0:
        px_step = world * F * S * S
        value = px_step * args.steps / elapsed / 1e6
        dom, dom_ms = ("encode", enc_ms) if enc_ms >= dec_ms else ("decode", dec_ms)
        alg_bytes = 2.0 * F * S * S               # 2 B/px per direction, one launch per batch
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        kernel = "k_%s_tiles" % ("enc" if dom == "encode" else "dec")
        traffic = pmc_traffic(kernel, F, S, levels)
        line = {
            "metric": "Mpixels/s encode+decode, 4K grayscale level=4 Medium", "value": round(value, 1),
            "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "C3 shard: %d frames/GPU of %dx%d u8 ramp(3), level=%d %s, Crossed, "
                                   "encode then decode, HBM-resident" % (F, S, S, levels, level.name),
                       "frames_per_gpu": F, "global_frames": world * F, "parallelism": "frames sharded x%d" % world,
                       "encode_ms": round(enc_ms, 4), "decode_ms": round(dec_ms, 4),
                       "max_abs_err": int(allst[:, 1].max()), "sq_err_sum": int(allst[:, 0].sum()),
                       "grid_checksums": [int(v) for v in allst[:, 2]]},
            "roofline": {"bound": "hbm", "kernel": kernel,
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(dom_ms, 4),
                         "other_kernel": {"kernel": "k_%s_tiles" % ("dec" if dom == "encode" else "enc"),
                                          "achieved": round(alg_bytes / (min(enc_ms, dec_ms) * 1e-3) / 1e9, 1),
                                          "avg_launch_ms": round(min(enc_ms, dec_ms), 4)},
                         "copy_same_run": {"achieved": round(alg_bytes / (copy_ms * 1e-3) / 1e9, 1),
                                           "avg_launch_ms": round(copy_ms, 4),
                                           "note": "16-B/lane copy kernel moving the same bytes"}},
        }
        if xgmi is not None:
            line["xgmi_scatter_gather"] = xgmi
        if world == 1 and not args.no_cpu:
            check = {"grid": grids[:2].cpu().numpy(), "out": outs[:2].cpu().numpy()}
            line["cpu_baseline"] = cpu_baseline(args, lut, check)
        emit(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
