"""`python bench.py --gpus N` must start its N ranks by itself (the driver runs exactly that command), keep working
under torchrun, print ONE JSON line from rank 0 and fail when a rank fails.  Run here through bench.py's own
launcher / rank split with --rehearse: gloo on the CPU, every collective of the real run, no codec work (the product
has no CPU codec)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HGI_BENCH_FAIL_RANK")}
    env.update(extra)
    return env


def _one_line(stdout):
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_plain_command_launches_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse", "--steps", "2", "--warmup", "1", "--frames", "5"],
                       env=_env(), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 2 and d["rehearsal"] is True and d["value"] is None and d["scaling"] == "weak"      # --frames: weak
    assert d["config"]["global_frames"] == 10 and d["config"]["shards"] == [[0, 5], [5, 5]]
    assert d["config"]["levels"] == 4 and d["config"]["max_error"] == 20       # the broadcast reached rank 0's line
    for rank in (0, 1):
        assert "rank %d/2 joined (gloo)" % rank in r.stderr
    # every rank's own record arrives in rank order (the real run sends plane-placement time, the separated flag and its
    # launch times; the rehearsal sends numbers made from the rank), with min / max over the ranks beside it
    pr = d["config"]["per_rank"]
    assert pr["planes_alloc_s"] == [0.001, 0.002] and pr["planes_separated"] == [True, True]
    assert pr["encode_ms"] == [0.5, 1.5] and pr["decode_ms"] == [0.25, 1.25] and pr["settle_steps"] == [8, 16]
    assert pr["encode_ms_min_max"] == [0.5, 1.5] and pr["decode_ms_min_max"] == [0.25, 1.25] and pr["frames"] == [5, 5]


def test_strong_scaling_shard_table():
    """Default: BASELINE configs[3]'s 512 frames sharded over the ranks, "scaling": "strong" -- N = 1 is the literal config,
    N = 8 the 64-frame shards ("1/2/4/8 GPUs x C3"); --frames F fixes the frames per GPU instead ("weak")."""
    import bench
    from rustyhgi_amd import batch
    for world, per in ((1, 512), (2, 256), (4, 128), (8, 64)):
        args = bench.parse_args(["--gpus", str(world)])
        scaling, G = bench.shard_plan(args, world)
        assert (scaling, G) == ("strong", 512)
        table = [batch.shard(G, world, r) for r in range(world)]
        assert table == [(r * per, per) for r in range(world)]
    # a rank count that does not divide the batch (what a 6-rank rehearsal on one device runs): blocks differ by one frame,
    # nothing is dropped, nothing is coded twice
    table = [batch.shard(512, 6, r) for r in range(6)]
    assert [c for _, c in table] == [86, 86, 85, 85, 85, 85] and table[0][0] == 0
    assert all(table[r + 1][0] == table[r][0] + table[r][1] for r in range(5)) and table[5][0] + table[5][1] == 512
    args = bench.parse_args(["--gpus", "4", "--frames", "64"])
    assert bench.shard_plan(args, 4) == ("weak", 256)
    args = bench.parse_args(["--gpus", "2", "--global-frames", "48"])
    assert bench.shard_plan(args, 2) == ("strong", 48)


@pytest.mark.timeout(600)
def test_default_command_is_strong_scaling_over_512_frames():
    """`python bench.py --gpus 2` as the driver runs it (no --frames): two ranks, 256 frames each, one line."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse", "--steps", "1", "--warmup", "0"],
                       env=_env(), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_line(r.stdout)
    assert d["scaling"] == "strong" and d["config"]["global_frames"] == 512 and d["config"]["shards"] == [[0, 256], [256, 256]]
    assert d["config"]["per_rank"]["frames"] == [256, 256]


@pytest.mark.timeout(900)
def test_eight_rank_rehearsal():
    """The N = 8 form of the command, rehearsed on the CPU (gloo): eight ranks join, take the 64-frame shards of the 512-frame
    batch in rank order, every collective of the real run completes, rank 0 prints the one line."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--rehearse", "--steps", "1", "--warmup", "0"],
                       env=_env(), capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["global_frames"] == 512
    assert d["config"]["shards"] == [[64 * k, 64] for k in range(8)]
    assert d["config"]["per_rank"]["planes_alloc_s"] == [round(0.001 * (k + 1), 4) for k in range(8)]
    for rank in range(8):
        assert "rank %d/8 joined (gloo)" % rank in r.stderr


@pytest.mark.timeout(300)
def test_torchrun_wrapped_form_still_works():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--rehearse", "--steps", "1",
                        "--warmup", "0"], env=_env(OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _one_line(r.stdout)["n_gpus"] == 2


@pytest.mark.timeout(300)
def test_failing_rank_fails_the_run_and_mismatch_is_refused():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse", "--steps", "1", "--warmup", "0"],
                       env=_env(HGI_BENCH_FAIL_RANK="1"), capture_output=True, text=True, timeout=280)
    assert r.returncode != 0 and not r.stdout.strip()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse"], env=_env(WORLD_SIZE="3", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr


def test_launcher_parent_never_imports_the_gpu_stack():
    """The launcher branch runs before torch / the HIP library are imported (a process that has initialised the GPU must
    not start the ranks): bench.py imports them inside run_rank only."""
    src = open(BENCH).read()
    head = src[:src.index("def run_rank")]
    assert "import torch" not in head and "import rustyhgi_amd" not in head
    assert "os.exec" not in src


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_ranks_with_the_real_codec_on_one_device():
    """The whole N-rank path with the HIP codec, on a one-GPU box: `python bench.py --gpus 2 --share-gpu` -- launcher, two
    ranks (both on cuda:0), broadcast, per-rank shards coded on the device, barrier/timing, all-gather (over gloo: RCCL
    refuses two ranks on one device; its path is exercised by HGI_BENCH_FORCE_DIST, profiles/r02_bench_forcedist.txt)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--share-gpu", "--steps", "3", "--warmup", "1", "--frames", "4",
                        "--size", "1024", "--no-cpu", "--no-extras", "--placement", "torch"],
                       env=_env(), capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r.stdout)
    c = d["config"]
    assert d["n_gpus"] == 2 and c["global_frames"] == 8 and c["max_abs_err"] <= 20
    assert len(c["grid_checksums"]) == 2 and c["grid_checksums"][0] != c["grid_checksums"][1]      # two different shards
    # the same eight frames in one process: the per-rank checksums add up to the whole
    one = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "1", "--warmup", "0", "--frames", "8", "--size", "1024",
                          "--no-cpu", "--no-extras", "--placement", "torch"], env=_env(), capture_output=True, text=True, timeout=560)
    assert one.returncode == 0, one.stderr[-3000:]
    whole = _one_line(one.stdout)["config"]
    assert whole["sq_err_sum"] == c["sq_err_sum"]


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_four_ranks_share_one_device_with_scatter_and_gather(oracle):
    """The N-rank path of the strong-scaling command with the real codec on a one-GPU box, on an uneven shard table (50 frames
    over 4 ranks: 13, 13, 12, 12), plus the labelled scatter -> code -> gather variant (`--xgmi-scatter`; here gloo through host
    memory: plumbing only).  Four ranks, not eight: a GPU box allows six processes on its card, and this test process and the
    launcher's agent count.  The decoded frames that arrive back on rank 0 must be what the oracle makes of those global frames."""
    import hashlib
    from conftest import SEED0
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--share-gpu", "--steps", "2", "--warmup", "1", "--global-frames", "50",
                        "--size", "512", "--no-cpu", "--no-extras", "--placement", "torch", "--xgmi-scatter"],
                       env=_env(), capture_output=True, text=True, timeout=860)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r.stdout)
    c = d["config"]
    assert d["n_gpus"] == 4 and d["scaling"] == "strong" and c["global_frames"] == 50 and c["per_rank"]["frames"] == [13, 13, 12, 12]
    assert len(set(c["grid_checksums"])) == 4 and c["max_abs_err"] <= 20
    x = d["xgmi_scatter_gather"]
    assert x["roundtrip_ok"] is True and x["frames_per_rank"] == 12 and "PLUMBING ONLY" in x["transport"]
    lut = oracle.linear_lut(2)[0]
    for rank, got in enumerate(x["first_gathered_frame_sha256_by_rank"]):
        img = oracle.synth(oracle.SYNTH_RAMP, SEED0 + 3, rank * 12, 512, 512)      # the scatter leg deals 12 frames per rank from frame 0
        want = oracle.decode(oracle.encode(img, 4, lut), 4)
        assert got == hashlib.sha256(want.tobytes()).hexdigest()[:16], "rank %d: the gathered frame is not the oracle's" % rank


def test_traffic_record_is_tied_to_the_build(tmp_path, monkeypatch):
    """`roofline.traffic` comes from a committed rocprofv3 record; bench.py takes it only when the record was made on this
    workload AND on this build of the library (the hash over the library's sources recorded by tools/summarize_prof.py equals
    the running one's) -- otherwise null with the reason.  The committed round-4 records carry the stamp of the committed tree."""
    import bench
    stamp = bench.build_stamp()
    assert len(stamp["source_sha256"]) == 16 and stamp["lib"] == "libhgi_hip.so" and "gfx950" in stamp["version"]
    # the committed records belong to the committed sources (a kernel edit without a new profile shows up here)
    for name, frames in (("r04_traffic.json", 512), ("r04_traffic_64.json", 64)):
        rec = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert rec["workload"] == {"frames": frames, "size": 4096, "levels": 4}
        assert rec["build"]["source_sha256"] == stamp["source_sha256"], \
            "profiles/%s was taken on another build: re-run tools/profile.sh and commit its traffic.json" % name
        got, src = bench.pmc_traffic("k_dec_tiles", frames, 4096, 4, stamp)
        assert src == "profiles/" + name and 0.99 < got / (2.0 * frames * 4096 * 4096) < 1.05
    # another build: refused, with the reason; another workload: no record
    got, why = bench.pmc_traffic("k_dec_tiles", 512, 4096, 4, dict(stamp, source_sha256="0" * 16))
    assert got is None and "another build" in why
    got, why = bench.pmc_traffic("k_dec_tiles", 500, 4096, 4, stamp)
    assert got is None and "no committed profile" in why
    # the stamp follows the sources: one byte more in a kernel file moves it
    real = os.path.join(ROOT, "rustyhgi_amd", "csrc")
    fake = tmp_path / "rustyhgi_amd" / "csrc"
    fake.mkdir(parents=True)
    (tmp_path / "include").mkdir()
    import shutil
    for f in os.listdir(real):
        if f.endswith((".hip", ".h", ".map")) or f == "Makefile":
            shutil.copy(os.path.join(real, f), fake / f)
    shutil.copy(os.path.join(ROOT, "include", "hgi.h"), tmp_path / "include" / "hgi.h")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.build_stamp()["source_sha256"] == stamp["source_sha256"]
    with open(fake / "hgi_dev.h", "a") as f:
        f.write("\n")
    assert bench.build_stamp()["source_sha256"] != stamp["source_sha256"]
