#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of HGI encode+decode on 4096x4096 u8 frames, level=4, Medium.

  python bench.py --gpus N --steps K --warmup W

The workload is BASELINE.json configs[3] itself: a batch of 512 independent 4096 x 4096 frames, sharded by frame over the N
GPUs -- 512 // N frames per GPU, "scaling": "strong" -- so N = 1 runs the literal config on one GPU (three planes of 8 GiB) and
N = 8 the 64-frame shards of the 8 x MI355X form.  `--frames F` runs F frames per GPU instead ("scaling": "weak").

N > 1 without a rendezvous in the environment: this process becomes a LAUNCHER -- it never touches the GPU, starts
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` as a child (one rank per GPU over
RCCL), relays rank 0's JSON line and exits with the child's code.  Started under torchrun (WORLD_SIZE set) it is a
rank, as before.

One step = one pass of the hot path over this rank's shard: encode its frames, then decode
them (BASELINE.json config C3, sharded by frame; frames are generated in place on each GPU from the
global frame index, so no pixel ever crosses xGMI).  Inputs are resident in HBM before the timed
region.  RCCL is used only where the batch split needs it: broadcast of the quantizer table and
all-gather of per-rank checksums.  Rank 0 prints ONE JSON line.

--rehearse: the same launcher, rendezvous, broadcast, sharding, barrier/timing and gather code on the CPU with the
gloo backend and NO codec work (there is no CPU codec in the product): what the CPU test suite runs to show that
`python bench.py --gpus 2` starts two ranks and returns one line.  Its line carries "rehearsal": true and value null.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED0 = 0x48474930
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (guide: MI355X_MICROARCH.md, chip-level parameters)


def table_quantizator(table, error):
    """A quantizer received as its 256-entry table (what rank 0 broadcasts)."""
    from rustyhgi_amd.quantizator import Quantizator

    class TableQuantizator(Quantizator):
        def __init__(self, t, e):
            self._t, self._e = np.ascontiguousarray(t, np.uint8), int(e)

        def quantize(self, value):
            return int(self._t[value & 0xFF])

        def error(self):
            return self._e

        def table(self):
            return self._t.copy()

    return TableQuantizator(table, error)


def build_stamp():
    """What identifies the library build a number was measured on: SHA-256 over the library's SOURCES (csrc/*.hip, *.h, the
    Makefile, the version script, include/hgi.h) -- a rebuild of the same sources on another box keeps it, any change to a
    kernel or its flags moves it -- beside the hash of the shared object that is loaded and the library's own version string."""
    import hashlib
    csrc = os.path.join(ROOT, "rustyhgi_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h", ".map")) or f == "Makefile")
    files.append(os.path.join(ROOT, "include", "hgi.h"))
    h = hashlib.sha256()
    for path in files:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    stamp = {"source_sha256": h.hexdigest()[:16]}
    try:
        from rustyhgi_amd import _ffi
        with open(_ffi.LIB_PATH, "rb") as f:
            stamp["lib_sha256"] = hashlib.sha256(f.read()).hexdigest()[:16]
        stamp["lib"] = os.path.basename(_ffi.LIB_PATH)
        stamp["version"] = _ffi.lib().hgi_version().decode()
    except Exception as e:      # (the stamp is information: never let it cost the line)
        stamp["lib_error"] = str(e)
    return stamp


def pmc_traffic(kernel, frames, size, levels, stamp):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE with the gfx950 x2 correction +
    WRITE_SIZE; tools/profile.sh -> profiles/*_traffic.json) -- if that profile was taken on this exact workload AND on this
    build of the library (its recorded source hash equals the running one's); else None, with the reason.  bench.py cannot run
    the profiler on itself."""
    why = "no committed profile of this workload"
    for name in ("r04_traffic.json", "r04_traffic_64.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                prof = json.load(f)
            if prof["workload"] != {"frames": frames, "size": size, "levels": levels}:
                continue
            have = prof.get("build", {}).get("source_sha256")
            if have != stamp.get("source_sha256"):
                why = "profiles/%s was taken on another build of the library (source hash %s, running %s)" % (name, have, stamp.get("source_sha256"))
                continue
            return prof["kernels"][kernel]["hbm_bytes_per_launch"], "profiles/" + name
        except (OSError, KeyError, ValueError):
            continue
    return None, why


def cpu_baseline(args, lut, gpu_check):
    """The oracle ('port' of the reference: scalar, one thread per image, src/encoder.rs:45-68) timed on ALL of this host's
    cores over a bounded sample of the same workload: as many threads as this process may run on, two frames each (one each
    beyond 128 threads), about a second of wall time.  `cores` is the number of threads used; `host_cores` what the machine
    reports.  Also re-checks the GPU result on those frames."""
    from oracle import hgi_oracle as O
    host_cores = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = host_cores
    threads = max(1, usable)
    base = np.stack([O.synth(O.SYNTH_RAMP, SEED0 + 3, f, args.size, args.size) for f in range(2)])

    def run(nthreads):
        frames = max(nthreads * (2 if nthreads <= 128 else 1), 8)
        imgs = np.concatenate([base] * ((frames + 1) // 2))[:frames]
        r = O.bench_batch(imgs, args.levels, lut, nthreads)
        for f in (0, 1):
            assert (r["grids"][f] == gpu_check["grid"][f]).all(), "GPU encode differs from the oracle"
            assert (r["outs"][f] == gpu_check["out"][f]).all(), "GPU decode differs from the oracle"
        px = frames * args.size * args.size
        return {"value": round(px / r["wall_s"] / 1e6, 1), "threads": nthreads, "frames": frames, "wall_s": round(r["wall_s"], 3),
                "one_thread_mpix_s": round(px / (r["enc_cpu_s"] + r["dec_cpu_s"]) / 1e6, 1)}

    full = run(threads)
    res = {"value": full["value"], "unit": "Mpixels/s", "cores": threads, "host_cores": host_cores, "threads": threads, "kind": "port",
           "sample": "%d frames %dx%d L%d %s, encode+decode, one oracle thread per frame on %d threads (every core this process may "
                     "run on; the host reports %d), %.2f s wall"
                     % (full["frames"], args.size, args.size, args.levels, args.quant, threads, host_cores, full["wall_s"]),
           "one_thread_mpix_s": full["one_thread_mpix_s"]}
    if threads > 64:      # rounds 1-3 capped the leg at 64 threads: the same sample shape there, so that the lines compare
        res["at_64_threads"] = run(64)
    return res


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=None,
                    help="frames per GPU per step: WEAK scaling.  Default: none -- the batch of --global-frames is sharded over the GPUs "
                         "(512 // N per GPU), STRONG scaling: N = 1 is BASELINE configs[3] itself")
    ap.add_argument("--global-frames", type=int, default=512, help="frames of the whole batch when --frames is not given (C3: 512)")
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--quant", default="medium")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-pfine", action="store_true", help="skip the P_fine (levels = 1) leg")
    ap.add_argument("--no-c4", action="store_true", help="skip the C4 leg (one 16384x16384 frame, level 8, High)")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed steps and the copy yardstick: no P_fine leg, no plain-allocation comparison "
                         "(profiling passes: every k_*_tiles launch in the trace is then a launch of the headline workload)")
    ap.add_argument("--xgmi-scatter", action="store_true",
                    help="also time the labelled variant where all frames start and end on GPU 0 (scatter, code, gather)")
    ap.add_argument("--placement", choices=("planes", "torch", "same-region"), default="planes",
                    help="frame stacks from hgi_planes_alloc (neighbouring planes in different HBM regions), from torch, or -- "
                         "for the counter comparison of DESIGN.md 5.1 -- deliberately all three in ONE region (planes 0, 2, 4 of five)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="every rank uses cuda:0 and the collectives go over gloo (RCCL refuses two ranks on one device): "
                         "the whole N-rank path with the real codec on a ONE-GPU box; `value` then says nothing about scaling "
                         "(at most 6 ranks: the GPU boxes allow six processes on a card)")
    ap.add_argument("--rehearse", action="store_true",
                    help="CPU rehearsal of the multi-rank plumbing (gloo, no codec work, value null)")
    return ap.parse_args(argv)


def shard_plan(args, world):
    """("strong", frames of the whole batch) by default -- BASELINE configs[3]'s 512 frames sharded over the ranks (SURVEY 8(e):
    "1/2/4/8 GPUs x C3") -- or ("weak", world * F) when --frames F fixes the frames per GPU."""
    if args.frames is not None:
        return "weak", int(world) * int(args.frames)
    return "strong", int(args.global_frames)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(args, argv):
    """--gpus N > 1 and no rendezvous in the environment: start the N ranks.  This process makes no HIP / torch.cuda
    call and replaces nothing (no exec): the ranks are children of a torch.distributed.run child, their stdout is ours."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch(args, argv)
    return run_rank(args)


def run_rank(args):
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

    import torch
    from rustyhgi_amd import batch
    from rustyhgi_amd.quantizator import Linear, QuantizationLevel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        return 2
    rehearse = args.rehearse
    if args.share_gpu:
        local = 0
    dev = torch.device("cpu") if rehearse else torch.device("cuda", local)
    cdev = torch.device("cpu") if (rehearse or args.share_gpu) else dev      # where the collectives' tensors live
    dist = None
    # HGI_BENCH_FORCE_DIST=1: take the RCCL code path (init, broadcast, barrier, all-reduce, all-gather) with one rank too
    if world > 1 or os.environ.get("HGI_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        if rehearse or args.share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        print("bench.py: rank %d/%d joined (%s)" % (rank, world, dist.get_backend()), file=sys.stderr)
    if not rehearse:
        torch.cuda.set_device(local)
    if os.environ.get("HGI_BENCH_FAIL_RANK") == str(rank):      # tests: a rank that dies must fail the whole run
        print("bench.py: rank %d failing on request" % rank, file=sys.stderr)
        os._exit(3)

    # ---- batch split: rank 0 owns the parameters; table + levels go out by RCCL broadcast ----
    level = QuantizationLevel.parse(args.quant)
    if rank == 0:
        q = Linear.from_level(level)
        lut, err, levels = batch.broadcast_params(dist, cdev, q.table(), q.error(), args.levels)
    else:
        lut, err, levels = batch.broadcast_params(dist, cdev)
    if dist is not None:
        print("bench.py: rank %d has the broadcast parameters (levels %d, max error %d)" % (rank, levels, err), file=sys.stderr)

    S = args.size
    # frame f of this rank is global frame first + f of config C3 (ramp(3)): produced where it is used
    scaling, G = shard_plan(args, world)
    first, F = batch.shard(G, world, rank)
    if F < 1:
        print("bench.py: %d frames cannot be sharded over %d ranks" % (G, world), file=sys.stderr)
        return 2

    def sync():
        if not rehearse:
            torch.cuda.synchronize(dev)

    def fence():
        sync()
        if dist is not None:
            dist.barrier()
            sync()

    if rehearse:
        codec = None
    else:
        codec = Codec(args, dev, local, lut, err, levels, first, F)

    # The device's clocks ramp over the first ~25 ms of work after idle, and the encoder (75 % VALU utilisation at full
    # clock) runs up to 1.4x slower until they have (tools/ramp.py, profiles/r02_ramp.txt; the decoder, 45 %, does not
    # care).  W = 5 warm-up steps are 4 ms.  So the device is first kept busy with untimed steps of the same workload
    # until the step time has stopped falling (bounded; reported as config.settle): the timed steps then measure the
    # steady state a batch job runs in, not the power manager's ramp.
    # Ranks finish their setup (placement probing, allocations) at different times: absorb that skew at a barrier HERE, so
    # that the barrier in front of the timed region finds every rank busy and is short -- a rank left idling there would
    # start the timed steps with its clocks down again.
    if dist is not None:
        fence()
    settle = codec.settle() if codec else None
    for _ in range(args.warmup):
        if codec:
            codec.step()
    if codec:
        codec.make_events(args.steps)
    fence()
    gc_was_on = gc.isenabled()
    gc.disable()      # (a collection in the middle of the enqueue loop would leave the device idle)
    t0 = time.perf_counter()
    for k in range(args.steps):
        if codec:
            codec.timed_step(k)
    fence()
    elapsed = batch.max_over_ranks(dist, time.perf_counter() - t0, cdev)
    if gc_was_on:
        gc.enable()

    if rehearse:
        # what the ranks would gather: [squared error, max error, checksum] -- here the shard itself, so that the
        # launcher test can see that every rank took its own block
        mine = torch.tensor([first, F, rank], dtype=torch.int64)
        allst = batch.gather_stats(dist, mine)
        # ... and the per-rank timing record of the real run (see per_rank_record): here each rank sends numbers derived
        # from its rank, so that the launcher test can see every rank's row arrive in rank order
        per_rank = per_rank_record(batch, dist, [0.001 * (rank + 1), 1.0, 0.5 + rank, 0.25 + rank, 8.0 * (rank + 1), float(F)])
        if rank == 0:
            emit(json.dumps({
                "metric": "Mpixels/s encode+decode, 4K grayscale level=4 Medium", "value": None, "unit": "Mpixels/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4), "higher_is_better": True, "scaling": scaling,
                "vs_baseline": None, "dtype": "u8", "data": "none", "rehearsal": True,
                "config": {"workload": "rehearsal of the rank plumbing on the CPU (gloo): no codec work",
                           "frames_per_gpu": F, "global_frames": G, "parallelism": "frames sharded x%d" % world,
                           "levels": levels, "max_error": err, "table_sum": int(np.asarray(lut, np.int64).sum()),
                           "shards": [[int(a), int(b)] for a, b, _ in allst], "per_rank": per_rank}}))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return 0

    enc_ms, dec_ms = codec.mean_ms()
    copy_ms = codec.copy_ms() if rank == 0 else None
    pfine = codec.p_fine() if rank == 0 and not (args.no_pfine or args.no_extras) else None
    placement = codec.placement(compare=not args.no_extras) if rank == 0 else None
    xgmi = codec.xgmi(dist, world, rank, fence, args.steps, G // world) if args.xgmi_scatter and dist is not None else None
    entropy = codec.entropy_stage() if rank == 0 and world == 1 and not args.no_extras else None

    c4 = codec.c4() if rank == 0 and world == 1 and not (args.no_extras or args.no_c4) else None

    # ---- per-rank checks + stats gather (RCCL all-gather) ----
    # every rank's own view of its run, so that a multi-GPU line explains itself: how long plane placement took there and
    # whether it succeeded (setup skew between ranks), and its own launch times (a slow GPU shows up by rank)
    per_rank = per_rank_record(batch, dist, [codec.setup_s, 1.0 if codec.separated else 0.0, enc_ms, dec_ms,
                                             float(settle["steps"]) if settle else 0.0, float(F)], cdev)
    allst = batch.gather_stats(dist, codec.stats().to(cdev))
    if dist is not None:
        print("bench.py: rank %d gathered the statistics of %d ranks" % (rank, len(allst)), file=sys.stderr)
    if not os.environ.get("HGI_BENCH_NOCHECK"):   # timing-only experiments produce wrong pixels
        assert int(allst[:, 1].max()) <= err, "reconstruction error exceeds the quantizer bound"

    if rank == 0:
        px_step = G * S * S                       # every rank's shard: the whole batch once per step
        value = px_step * args.steps / elapsed / 1e6
        dom, dom_ms = ("encode", enc_ms) if enc_ms >= dec_ms else ("decode", dec_ms)
        alg_bytes = 2.0 * F * S * S               # 2 B/px per direction, one launch per shard (rank 0's: the largest)
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        kernel = "k_%s_tiles" % ("enc" if dom == "encode" else "dec")
        stamp = build_stamp()
        traffic, traffic_from = pmc_traffic(kernel, F, S, levels, stamp)
        if world == 1 and scaling == "strong" and G == 512:
            workload = "C3: %d frames of %dx%d u8 ramp(3), level=%d %s, Crossed, encode then decode, HBM-resident, one GPU" % (G, S, S, levels, level.name)
        else:
            workload = ("C3%s: %d frames of %dx%d u8 ramp(3) sharded over %d GPUs (%d per GPU), level=%d %s, Crossed, encode then decode, "
                        "HBM-resident" % (" shard" if scaling == "weak" else "", G, S, S, world, F, levels, level.name))
        line = {
            "metric": "Mpixels/s encode+decode, 4K grayscale level=4 Medium", "value": round(value, 1),
            "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload,
                       "frames_per_gpu": F, "global_frames": G, "parallelism": "frames sharded x%d" % world,
                       "collectives": "none" if dist is None else "%s: broadcast(258 B) + all_reduce(max) + all_gather(24 B)" % dist.get_backend(),
                       "encode_ms": round(enc_ms, 4), "decode_ms": round(dec_ms, 4),
                       "max_abs_err": int(allst[:, 1].max()), "sq_err_sum": int(allst[:, 0].sum()),
                       "grid_checksums": [int(v) for v in allst[:, 2]]},
            "roofline": {"bound": "hbm", "kernel": kernel,
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_from": traffic_from,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": round(dom_ms, 4),
                         "other_kernel": {"kernel": "k_%s_tiles" % ("dec" if dom == "encode" else "enc"),
                                          "achieved": round(alg_bytes / (min(enc_ms, dec_ms) * 1e-3) / 1e9, 1),
                                          "avg_launch_ms": round(min(enc_ms, dec_ms), 4)},
                         "copy_same_run": {"achieved": round(alg_bytes / (copy_ms * 1e-3) / 1e9, 1),
                                           "avg_launch_ms": round(copy_ms, 4),
                                           "note": "16-B/lane copy kernel moving the same bytes"}},
        }
        line["build"] = stamp
        line["config"]["placement"] = placement
        line["config"]["per_rank"] = per_rank
        line["config"]["per_step_ms"] = codec.per_step_ms()
        line["config"]["settle"] = settle
        if args.share_gpu:
            line["share_gpu"] = "all %d ranks time-share cuda:0, collectives over gloo: exercises the N-rank path on a one-GPU box; not a scaling number" % world
        if pfine is not None:
            line["p_fine"] = pfine
        if xgmi is not None:
            line["xgmi_scatter_gather"] = xgmi
        if entropy is not None:
            line["entropy_stage"] = entropy
        if c4 is not None:
            line["c4"] = c4
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(args, lut, codec.sample())
        emit(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    codec.close()
    return 0


def per_rank_record(batch, dist, values, device=None):
    """All-gather of [plane-placement seconds, planes separated (0/1), encode ms, decode ms, settle steps, frames] -> the
    `per_rank` block of the line: every rank's numbers in rank order plus min / max of the two launch times."""
    import torch
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device or "cpu")
    rows = batch.gather_stats(dist, t)
    return {"planes_alloc_s": [round(float(r[0]), 4) for r in rows], "planes_separated": [bool(r[1] > 0.5) for r in rows],
            "encode_ms": [round(float(r[2]), 4) for r in rows], "decode_ms": [round(float(r[3]), 4) for r in rows],
            "settle_steps": [int(r[4]) for r in rows], "frames": [int(r[5]) for r in rows],
            "encode_ms_min_max": [round(float(rows[:, 2].min()), 4), round(float(rows[:, 2].max()), 4)],
            "decode_ms_min_max": [round(float(rows[:, 3].min()), 4), round(float(rows[:, 3].max()), 4)]}


class Codec:
    """This rank's frames, contexts and timed steps on its GPU."""

    def __init__(self, args, dev, local, lut, err, levels, first, frames):
        import torch
        import rustyhgi_amd as H
        from rustyhgi_amd import _ffi
        from rustyhgi_amd.interpolator import Crossed
        self.torch, self.H, self._ffi, self.Crossed = torch, H, _ffi, Crossed
        self.args, self.dev, self.lut, self.err, self.levels = args, dev, lut, err, levels
        # a real (non-null) stream made current for torch: the codec launches, the timing events and the
        # torch ops around them all live on it
        self.stream = torch.cuda.Stream(dev)
        torch.cuda.set_stream(self.stream)
        self.ctx = ctx = H.Context(local)
        ctx.set_stream(self.stream.cuda_stream)
        F, S = int(frames), args.size
        self.F, self.S, self.first = F, S, first
        ctx.reserve(S, S, levels, F)
        # The three frame stacks.  Default: planes placed by the library (hgi_planes_alloc) so that each launch reads
        # one HBM region and writes another (+4-5 % on MI355X over planes that share a region, which is
        # what plain allocations give about half of the time).  --placement torch: plain torch allocations.
        self.planes = None
        t_setup = time.perf_counter()
        if args.placement == "planes":
            # best effort inside the library; if it could not establish the separation, ask again while the first set
            # is still held (the new candidates then come from elsewhere), up to twice
            held = []
            try:
                for attempt in range(3):
                    self.planes = H.Planes(ctx, F * S * S, 3)
                    if self.planes.separated:
                        break
                    if attempt < 2:
                        held.append(self.planes)
                for p in held:
                    p.close()
                self.imgs, self.grids, self.outs = (self.planes.torch(i, (F, S, S)) for i in range(3))
            except Exception as e:      # placement is an optimisation: never let it cost the run
                print("bench.py: plane placement failed (%s); falling back to plain allocations" % e, file=sys.stderr)
                for p in held + ([self.planes] if self.planes is not None else []):
                    try:
                        p.close()
                    except Exception:
                        pass
                self.planes = None
                args.placement = "torch (planes failed)"
                self.imgs = torch.empty((F, S, S), dtype=torch.uint8, device=dev)
                self.grids = torch.empty_like(self.imgs)
                self.outs = torch.empty_like(self.imgs)
        elif args.placement == "same-region":
            # three planes of seven whose two pairings (image -> grid, grid -> image) both probe SLOW: one region
            self.planes = H.Planes(ctx, F * S * S, 7)
            for _ in range(12):
                self.planes.probe_ms(0, 1)                  # clocks
            t = {(a, b): self.planes.probe_ms(a, b) for a in range(7) for b in range(7) if a != b}
            mid = (min(t.values()) + max(t.values())) / 2
            pick = next(((a, b, c) for a in range(7) for b in range(7) for c in range(7)
                         if len({a, b, c}) == 3 and t[(a, b)] > mid and t[(b, c)] > mid), (0, 2, 4))
            print("bench.py: same-region planes %s: probes %.4f / %.4f ms (fastest pairing %.4f)"
                  % (pick, t[(pick[0], pick[1])], t[(pick[1], pick[2])], min(t.values())), file=sys.stderr)
            self.imgs, self.grids, self.outs = (self.planes.torch(i, (F, S, S)) for i in pick)
        else:
            self.imgs = torch.empty((F, S, S), dtype=torch.uint8, device=dev)
            self.grids = torch.empty_like(self.imgs)
            self.outs = torch.empty_like(self.imgs)
        self.setup_s = time.perf_counter() - t_setup            # plane placement (probing included) or plain allocation
        self.separated = bool(self.planes is not None and self.planes.separated)
        _ffi.check(_ffi.lib().hgi_synth_u8_dev(ctx.handle, _ffi.SYNTH_RAMP, SEED0 + 3, first, S, S,
                                               self.imgs.data_ptr(), F, S * S))
        self.quant = table_quantizator(lut, err)
        self.enc = H.Encoder(Crossed(), self.quant, levels, context=ctx)
        self.dec = H.Decoder(Crossed(), context=ctx)
        self.ev = []

    def step(self):
        self.enc.encode_batch(self.imgs, out=self.grids)
        self.dec.decode_batch(self.grids, self.levels, out=self.outs)

    def settle(self, group=8, max_steps=160, tol=0.004):
        """Untimed steps in groups until two consecutive groups agree within `tol` (at least three groups)."""
        torch = self.torch
        series = []
        while len(series) * group < max_steps:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(group):
                self.step()
            b.record()
            b.synchronize()
            series.append(a.elapsed_time(b) / group)
            if len(series) >= 3 and abs(series[-1] - series[-2]) <= tol * series[-2]:
                break
        return {"steps": len(series) * group, "first_group_ms_per_step": round(series[0], 4),
                "last_group_ms_per_step": round(series[-1], 4),
                "note": "untimed steps of the same workload before the W warm-up steps, until the step time stops "
                        "falling: device clocks ramp for ~25 ms after idle and the encoder is clock-sensitive until then"}

    def make_events(self, steps):
        self.ev = [[self.torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
        # torch creates the HIP event behind an Event at its first record(): do that here, not inside the timed region
        # (60 hipEventCreate calls cost nothing on a warm box and have cost 10 ms of idle GPU on a freshly started one)
        for trio in self.ev:
            for e in trio:
                e.record()
        self.torch.cuda.synchronize(self.dev)

    def timed_step(self, k):
        ev = self.ev[k]
        ev[0].record()
        self.enc.encode_batch(self.imgs, out=self.grids)
        ev[1].record()
        self.dec.decode_batch(self.grids, self.levels, out=self.outs)
        ev[2].record()

    def mean_ms(self):
        return (float(np.mean([e[0].elapsed_time(e[1]) for e in self.ev])),
                float(np.mean([e[1].elapsed_time(e[2]) for e in self.ev])))

    def per_step_ms(self):
        """every timed step's two launches, in order (shows drift inside the timed region, if any)"""
        return {"encode": [round(e[0].elapsed_time(e[1]), 4) for e in self.ev],
                "decode": [round(e[1].elapsed_time(e[2]), 4) for e in self.ev]}

    def _timed(self, fn, reps, warm=1):
        torch = self.torch
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(warm):
            fn()
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize(self.dev)
        return a.elapsed_time(b) / reps

    def copy_ms(self):
        """same-run reference point: a plain 16-B/lane streaming copy of the same 2 x F frames of traffic"""
        n = self.F * self.S * self.S
        L, ctx = self._ffi.lib(), self.ctx
        ms = self._timed(lambda: self._ffi.check(L.hgi_copy_u8_dev(ctx.handle, self.imgs.data_ptr(), self.outs.data_ptr(), n)), 5)
        self._restore()
        return ms

    def _restore(self):
        self.enc.encode_batch(self.imgs, out=self.grids)
        self.dec.decode_batch(self.grids, self.levels, out=self.outs)      # the decoded frames the checks read
        self.torch.cuda.synchronize(self.dev)

    def p_fine(self):
        """The finest pass alone (SURVEY 8(d) `P_fine`; the north star's "level-0 interpolation pass", reference
        src/utils.rs:16-18 with e = 1): the product kernels at levels = 1 on the same frames.  At levels = 1 the lattice
        is the even/even quarter of the pixels, which the launch copies through, so it MOVES 2 B/px; the pass as SURVEY
        accounts it is 1.75 B/px (reads N, writes the 3/4 N new pixels).  On a batch of more than 64 frames the block also
        carries the same measurement on the first 64 frames (the shard one GPU holds at N = 8)."""
        res = self._p_fine_on(self.F)
        if self.F > 64:
            sub = self._p_fine_on(64)
            res["first_64_frames"] = {k: sub[k] for k in ("workload", "encode_ms", "decode_ms", "achieved", "frac", "moved_frac")}
        self._restore()
        return res

    def _p_fine_on(self, F):
        H, S = self.H, self.S
        imgs, grids, outs = self.imgs[:F], self.grids[:F], self.outs[:F]
        enc1 = H.Encoder(self.Crossed(), self.quant, 1, context=self.ctx)
        reps = max(5, min(self.args.steps, 20))
        # alternating encode -> decode like the bench step (same-direction launches back to back run ~4 % faster)
        ev = [[self.torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(reps)]
        for i in range(10 + reps):
            e = ev[max(i - 10, 0)]
            e[0].record()
            enc1.encode_batch(imgs, out=grids)
            e[1].record()
            self.dec.decode_batch(grids, 1, out=outs)
            e[2].record()
        self.torch.cuda.synchronize(self.dev)
        e_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
        d_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
        err1 = int((imgs[:2].to(self.torch.int16) - outs[:2].to(self.torch.int16)).abs().max())
        alg, moved = 1.75 * F * S * S, 2.0 * F * S * S
        slow = max(e_ms, d_ms)
        return {"workload": "finest pass alone: k_enc_tiles / k_dec_tiles at levels=1 on %s%d frames" % ("the same " if F == self.F else "the first ", F),
                "encode_ms": round(e_ms, 4), "decode_ms": round(d_ms, 4), "max_abs_err": err1,
                "algorithmic_bytes_per_launch": alg, "moved_bytes_per_launch": moved,
                "achieved": round(alg / (slow * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(alg / (slow * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "moved_frac": round(moved / (slow * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "note": "1.75 B/px algorithmic (3/4 of the pixels are new); the launch also copies the 1/4 lattice "
                        "through, so 2 B/px cross HBM; frac uses the slower direction"}

    def c4(self):
        """BASELINE config C4 -- ONE 16384 x 16384 u8 frame ramp(4), level 8, High, Crossed: the only config whose single
        launch exceeds the 256 MiB Infinity Cache.  Timed in the bench step's own pattern (encode, then decode of what
        was just written), per launch with events on the codec's stream, after the frame stacks of the headline workload
        have been released.  A deep pyramid, yet ONE launch per direction: the tile kernel runs four fused levels and
        rebuilds the four above a tile for itself (profiles/r03_c4_summary.md has the rocprofv3 rows)."""
        import hashlib
        torch, H, _ffi = self.torch, self.H, self._ffi
        L = _ffi.lib()
        W = 16384
        n = W * W
        res = {"workload": "C4: 1 x 16384x16384 u8 ramp(4), level=8 High, Crossed; encode then decode, HBM-resident"}
        try:
            err = np.zeros(1, np.uint8)
            lut = np.zeros(256, np.uint8)
            _ffi.check(L.hgi_linear_lut(3, lut.ctypes.data, err.ctypes.data))
            # three planes from the library's placement allocator, like the headline's frame stacks (planes of 256 MiB are
            # allocated at the probe's 1 GiB so that they can be placed: include/hgi.h)
            c4_planes = None
            if self.args.placement == "planes":
                try:
                    c4_planes = H.Planes(self.ctx, n, 3)
                    img, grid, out = (c4_planes.torch(i, (W, W)) for i in range(3))
                except Exception:
                    c4_planes = None
            if c4_planes is None:
                img = torch.empty((W, W), dtype=torch.uint8, device=self.dev)
                grid, out = torch.empty_like(img), torch.empty_like(img)
            res["planes_separated"] = bool(c4_planes is not None and c4_planes.separated)
            _ffi.check(L.hgi_synth_u8_dev(self.ctx.handle, _ffi.SYNTH_RAMP, SEED0 + 4, 0, W, W, img.data_ptr(), 1, n))
            self.ctx.reserve(W, W, 8, 1)

            def enc():
                _ffi.check(L.hgi_encode_u8_dev(self.ctx.handle, img.data_ptr(), W, W, 8, 1, lut.ctypes.data, grid.data_ptr(), 1, n))

            def dec():
                _ffi.check(L.hgi_decode_u8_dev(self.ctx.handle, grid.data_ptr(), W, W, 8, 1, out.data_ptr(), 1, n))

            reps, warm = 60, 120
            ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(reps)]
            for i in range(warm + reps):
                e = ev[max(i - warm, 0)]
                e[0].record()
                enc()
                e[1].record()
                dec()
                e[2].record()
            torch.cuda.synchronize(self.dev)
            e_all = np.array([e[0].elapsed_time(e[1]) for e in ev]) * 1e3
            d_all = np.array([e[1].elapsed_time(e[2]) for e in ev]) * 1e3
            e_us, d_us = float(e_all.mean()), float(d_all.mean())

            def pair():
                enc()
                dec()

            p_us = self._timed(pair, reps, warm=10) * 1e3      # the same pattern with no event between the two calls
            c_us = self._timed(lambda: _ffi.check(L.hgi_copy_u8_dev(self.ctx.handle, img.data_ptr(), out.data_ptr(), n)), 20, warm=5) * 1e3
            dec()
            torch.cuda.synchronize(self.dev)
            max_err = int((img[:4096].to(torch.int16) - out[:4096].to(torch.int16)).abs().max())
            sha = hashlib.sha256(grid.cpu().numpy().tobytes()).hexdigest()
            want = None
            try:
                with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
                    want = json.load(f)["ramp4_16384/L8/q3/i1"]["sha_grid"]
            except (OSError, KeyError, ValueError):
                pass
            alg = 2.0 * n

            def rate(us, series=None):
                r = {"us": round(us, 2), "achieved": round(alg / us / 1e3, 1), "unit": "GB/s", "frac": round(alg / us / 1e3 / HBM_PEAK_GBS, 4)}
                if series is not None:
                    r.update({"calls": int(series.size), "min_us": round(float(series.min()), 2), "median_us": round(float(np.median(series)), 2),
                              "max_us": round(float(series.max()), 2)})
                return r

            res.update({"algorithmic_bytes_per_call": alg, "peak": HBM_PEAK_GBS, "encode": rate(e_us, e_all), "decode": rate(d_us, d_all),
                        "pair": {"us": round(p_us, 2), "achieved": round(2 * alg / p_us / 1e3, 1), "unit": "GB/s",
                                 "frac": round(2 * alg / p_us / 1e3 / HBM_PEAK_GBS, 4),
                                 "note": "encode + decode per pair with ONE event pair around all pairs: an event record between "
                                         "two calls costs ~2 us, which the per-direction figures include"},
                        "copy_same_run": rate(c_us), "max_abs_err": max_err, "grid_sha256": sha[:16],
                        "grid_matches_golden": (sha == want) if want else None,
                        "launches_per_call": 1,
                        "note": "per CALL, mean of the timed calls (one launch each: four fused levels, the four above a tile "
                                "rebuilt in the kernel); decode reads the grid the encode before it has just written, as in the bench step"})
            del img, grid, out
            if c4_planes is not None:
                c4_planes.close()
        except Exception as e:      # an extra: never let it cost the line
            res["error"] = "%s: %s" % (type(e).__name__, e)
        self._restore()
        return res

    def xgmi(self, dist, world, rank, fence, steps, fx):
        """optional, separately labelled: every frame starts and ends on GPU 0 -- `fx` frames per rank, scattered
        from rank 0, coded where they land, the decoded frames gathered back.  Bound by the source GPU's xGMI links, not by
        the codec; never part of `value`.  Under --share-gpu (gloo, one device) the frames go through host memory: plumbing
        only, no bandwidth claim.  The line carries the SHA-256 of every rank's first gathered frame, which the GPU test
        compares with the oracle's decode(encode()) of that global frame."""
        import hashlib
        torch, _ffi = self.torch, self._ffi
        from rustyhgi_amd import batch
        S, dev = self.S, self.dev
        fx = int(min(fx, self.F))
        allf = torch.empty((world * fx, S, S), dtype=torch.uint8, device=dev) if rank == 0 else None
        allo = torch.empty_like(allf) if rank == 0 else None
        if rank == 0:
            _ffi.check(_ffi.lib().hgi_synth_u8_dev(self.ctx.handle, _ffi.SYNTH_RAMP, SEED0 + 3, 0, S, S,
                                                   allf.data_ptr(), world * fx, S * S))
        mine_in = torch.empty((fx, S, S), dtype=torch.uint8, device=dev)
        # what crosses the links lives in plain allocations (planes above 1 GiB are composed of mapped physical chunks, which a
        # peer cannot be assumed to reach); the grid, which stays on the device, is the placed plane
        grids, outs = self.grids[:fx], torch.empty((fx, S, S), dtype=torch.uint8, device=dev)

        def xstep():
            batch.scatter_frames(dist, allf, mine_in)
            self.enc.encode_batch(mine_in, out=grids)
            self.dec.decode_batch(grids, self.levels, out=outs)
            batch.gather_frames(dist, outs, allo)

        xstep()
        fence()
        t1 = time.perf_counter()
        nx = max(1, min(steps, 5))
        for _ in range(nx):
            xstep()
        fence()
        xs = batch.max_over_ranks(dist, time.perf_counter() - t1, torch.device("cpu") if self.args.share_gpu else dev) / nx
        res = None
        if rank == 0:
            same = bool(torch.equal(allo[:fx], outs)) and \
                int((allf.to(torch.int16) - allo.to(torch.int16)).abs().max()) <= self.err
            res = {"ms_per_step": round(xs * 1e3, 4), "value": round(world * fx * S * S / xs / 1e6, 1), "unit": "Mpixels/s",
                   "frames_per_rank": fx, "bytes_over_links_per_step": 2 * (world - 1) * fx * S * S, "roundtrip_ok": same,
                   "first_gathered_frame_sha256_by_rank": [hashlib.sha256(allo[r * fx].cpu().numpy().tobytes()).hexdigest()[:16] for r in range(world)],
                   "transport": ("gloo through host memory, all ranks on one device: PLUMBING ONLY, not a bandwidth figure" if self.args.share_gpu
                                 else "torch.distributed scatter / gather over RCCL (xGMI)"),
                   "note": "frames scattered from and gathered to GPU 0; per-link bound, reported beside the sharded number, never as it"}
        del allf, allo, mine_in, outs
        self._restore()
        return res

    def placement(self, compare=True):
        """How the frame stacks were placed, and -- for transparency -- the same step timed on plain torch allocations
        in this process (their regions are whatever the allocator gave: either pairing may come out fast or slow)."""
        torch, F, S = self.torch, self.F, self.S
        info = {"mode": self.args.placement}
        if self.planes is not None:
            info.update({"api": "hgi_planes_alloc(bytes, %d): neighbouring planes in different HBM regions (DESIGN.md 5.1)" % self.planes.count,
                         "separated": self.planes.separated, "report": self.planes.report})
        if not compare:
            return info
        a = torch.empty((F, S, S), dtype=torch.uint8, device=self.dev)
        b, c = torch.empty_like(a), torch.empty_like(a)
        a.copy_(self.imgs)
        reps, warm = 8, 40          # 40 untimed steps first: the allocations above let the clocks fall back (see settle())
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(reps)]
        for i in range(warm + reps):
            e = ev[max(i - warm, 0)]
            e[0].record()
            self.enc.encode_batch(a, out=b)
            e[1].record()
            self.dec.decode_batch(b, self.levels, out=c)
            e[2].record()
        torch.cuda.synchronize(self.dev)
        same = bool(torch.equal(b, self.grids)) and bool(torch.equal(c, self.outs))
        info["plain_torch_allocations"] = {"encode_ms": round(float(np.mean([e[0].elapsed_time(e[1]) for e in ev])), 4),
                                           "decode_ms": round(float(np.mean([e[1].elapsed_time(e[2]) for e in ev])), 4),
                                           "same_bytes_as_timed_run": same}
        del a, b, c
        return info

    def entropy_stage(self):
        """SURVEY 8(f4) and beyond: the archive's DEFLATE step (src/archive.rs:34-40) for this shard's grids, on the
        device (hgi_deflate_grids_dev) -- bytes out and time, streams into a pinned host buffer.  An
        extra beside the headline, never part of it; one stream is inflated with zlib as a check."""
        import ctypes
        import struct
        import time
        import zlib
        torch, _ffi, S = self.torch, self._ffi, self.S
        F = min(self.F, 64)      # an extra beside the headline: the first 64 grids of the shard bound its time and its pinned buffer
        try:
            cap = S * S // 2 + 4096
            out = torch.empty((F, cap), dtype=torch.uint8, pin_memory=True)
            sizes, offsets = (ctypes.c_size_t * F)(), (ctypes.c_size_t * F)()

            def strided():
                _ffi.check(_ffi.lib().hgi_deflate_grids_dev(self.ctx.handle, self.grids.data_ptr(), S, S, F, S * S, out.data_ptr(), cap, sizes))

            def packed():
                _ffi.check(_ffi.lib().hgi_deflate_grids_packed_dev(self.ctx.handle, self.grids.data_ptr(), S, S, F, S * S, out.data_ptr(), F * cap,
                                                                   offsets, sizes))

            def median_s(call):
                call()
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    call()
                    ts.append(time.perf_counter() - t0)
                return sorted(ts)[2]

            t_strided = median_s(strided)
            first_strided = bytes(out[0, :sizes[0]].numpy())
            t = median_s(packed)
            total = int(sum(sizes))
            flat = out.view(-1)
            first = bytes(flat[offsets[0]:offsets[0] + sizes[0]].numpy())
            last = bytes(flat[offsets[F - 1]:offsets[F - 1] + sizes[F - 1]].numpy())
            ok = True
            for f, stream in ((0, first), (F - 1, last)):
                body = zlib.decompressobj(-15).decompress(stream)
                ok = ok and body == struct.pack("<Q", S * S) + self.grids[f].cpu().numpy().tobytes() + struct.pack("<Q", S)
            res = {"api": "hgi_deflate_grids_packed_dev: raw DEFLATE (dynamic Huffman, literals + run matches) of each grid's bincode image; "
                          "a group's streams packed on the device and brought down with one copy",
                   "frames": F, "grid_bytes": F * S * S, "stream_bytes": total, "ratio": round(F * S * S / max(total, 1), 2),
                   "ms": round(t * 1e3, 3), "grid_gb_s": round(F * S * S / t / 1e9, 1), "stream_gb_s": round(total / t / 1e9, 1),
                   "host_buffer": "pinned", "one_copy_per_frame_ms": round(t_strided * 1e3, 3),
                   "same_streams_as_strided_call": first == first_strided,
                   "first_and_last_stream_inflate_to_their_grids": bool(ok)}
            del out
            return res
        except Exception as e:      # an extra: never let it cost the line
            return {"error": "%s: %s" % (type(e).__name__, e)}

    def stats(self):
        torch, _ffi, F, S = self.torch, self._ffi, self.F, self.S
        stats = torch.zeros(3 * F, dtype=torch.int64, device=self.dev)
        _ffi.check(_ffi.lib().hgi_diff_stats_dev(self.ctx.handle, self.imgs.data_ptr(), self.outs.data_ptr(), S, S, F, S * S,
                                                 stats.data_ptr()))
        st = stats.view(F, 3)
        return torch.stack([st[:, 0].sum(), st[:, 1].max(), self.grids.view(-1)[::4099].to(torch.int64).sum()])

    def sample(self):
        return {"grid": self.grids[:2].cpu().numpy(), "out": self.outs[:2].cpu().numpy()}

    def close(self):
        del self.imgs, self.grids, self.outs
        if self.planes is not None:
            self.planes.close()
        self.ctx.close()


if __name__ == "__main__":
    sys.exit(main())
