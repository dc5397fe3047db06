"""CPU-side sanitizer runs (GPU AddressSanitizer does not exist on this pool, so these are the CPU
builds only): the oracle under ASan + UBSan against the committed goldens and odd shapes, and the entropy stage's host
planner (hgi_huffman_host.h, plain C++) fuzzed under ASan + UBSan, every block it emits parsed by zlib."""
import os
import struct
import subprocess
import sys
import zlib

import pytest

from conftest import ROOT


def _gcc_file(name):
    path = subprocess.check_output(["gcc", "-print-file-name=" + name], text=True).strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


CHILD = r"""
import os, sys, json, numpy as np
sys.path.insert(0, %(root)r)
from oracle import hgi_oracle as O
from oracle import hgi_numpy as NP
assert "asan" in O._SO, O._SO
small = dict(np.load(os.path.join(%(root)r, "tests", "golden", "small_cases.npz"), allow_pickle=False))
n = 0
for key in sorted(small):
    if not key.startswith("grid/"):
        continue
    _, case, lv, q, it = key.split("/")
    img = small["in/" + case]
    levels, interp = int(lv[1:]), int(it[1:])
    lut = O.noop_lut() if q == "qnoop" else O.linear_lut(int(q[1:]))[0]
    grid, rec, fb = O.encode(img, levels, lut, interp, want_rec=True)
    assert (grid == small[key]).all(), key
    assert (O.decode(grid, levels, interp) == small["dec/" + key[5:]]).all(), key
    n += 1
# odd shapes and level counts the goldens do not hold, against the numpy restatement: the unchecked indexing the
# reference does (src/grid.rs:20-27) is exactly what ASan watches in the C restatement
rng = np.random.default_rng(7)
for (w, h, levels) in [(1, 1, 0), (1, 1, 31), (2, 3, 5), (17, 1, 4), (1, 19, 4), (33, 31, 6), (64, 64, 6), (65, 63, 7), (130, 70, 3), (255, 257, 9)]:
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    for q in (0, 2):
        lut = O.linear_lut(q)[0]
        for interp in (0, 1):
            grid = O.encode(img, levels, lut, interp)
            assert (grid == NP.encode(img, levels, lut, interp)).all(), (w, h, levels, q, interp)
            assert (O.decode(grid, levels, interp) == NP.decode(grid, levels, interp)).all()
            n += 1
frames = np.stack([O.synth(O.SYNTH_RAMP, 5, f, 96, 40) for f in range(5)])
r = O.bench_batch(frames, 4, O.linear_lut(2)[0], 3)          # the threaded batch entry point bench.py times
assert (r["grids"][4] == O.encode(frames[4], 4, O.linear_lut(2)[0])).all()
print("asan-oracle ok", n)
"""


def test_oracle_under_asan_ubsan(tmp_path):
    asan = _gcc_file("libasan.so")
    if asan is None:
        pytest.skip("no libasan in this toolchain")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    so = os.path.join(ROOT, "oracle", "_build", "libhgi_oracle_asan.so")
    env = dict(os.environ, LD_PRELOAD=asan, HGI_ORACLE_SO=so,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",       # (the interpreter itself leaks by design)
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "asan-oracle ok" in p.stdout
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-4000:]


def test_huffman_planner_fuzz_under_asan_ubsan(tmp_path):
    """10 000 random 286-bin histograms through the planner built by plain g++ with both sanitizers; the program checks
    lengths <= 15, Kraft, prefix-freeness and plan_frame's bookkeeping itself, and zlib parses every header here."""
    exe, blocks = str(tmp_path / "fuzz_huffman"), str(tmp_path / "blocks.bin")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", os.path.join(ROOT, "tests", "cpp", "fuzz_huffman.cpp"), "-o", exe])
    p = subprocess.run([exe, "10000", "0x48474930", blocks], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr[-4000:]
    assert "10000 cases ok" in p.stdout
    data = open(blocks, "rb").read()
    at = n = 0
    while at < len(data):
        (nb,) = struct.unpack_from("<I", data, at)
        block = data[at + 4: at + 4 + nb]
        at += 4 + nb
        d = zlib.decompressobj(-15)
        assert d.decompress(block) == b"" and d.eof, "zlib rejects the block header of case %d" % n
        n += 1
    assert n == 10000


def test_archive_reader_on_hostile_input_under_asan_ubsan(tmp_path):
    """include/hgi_archive.hpp: a header that announces exabytes, every truncation, a grid whose width contradicts the
    metadata, 3 000 random bit flips -- ArchiveError or a consistent grid, never a crash or a wild allocation."""
    exe = str(tmp_path / "test_archive_hardening")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_archive_hardening.cpp"), "-lz", "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "archive hardening ok" in p.stdout, p.stdout + p.stderr[-4000:]


def test_plane_lineup_under_asan_ubsan(tmp_path):
    """csrc/hgi_lineup.h -- which classified 1 GiB chunk goes where in a set of composed planes (hgi_planes_alloc above 1 GiB) --
    is plain C++: 20 000 synthetic classifications (classes in runs, as the driver hands memory out, and shuffled) against its
    promises: every chunk at most once; a complete line-up has neighbouring planes on different classes at every offset; a
    two-sided one has EVERY chunk of a plane differ from EVERY chunk of its neighbours and is found whenever one exists; a
    partial one keeps what lined up; the stall detector fires exactly when the last chunks all joined the largest group."""
    exe = str(tmp_path / "test_lineup")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", os.path.join(ROOT, "tests", "cpp", "test_lineup.cpp"), "-o", exe])
    p = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "20000 cases ok" in p.stdout, p.stdout + p.stderr[-4000:]


def test_fastdiv_is_exact(tmp_path):
    """The tile kernels never divide: block -> tile index math goes through hgi_fastdiv.h with multipliers the host
    derives per launch.  Checked against the machine's division (UBSan on): every divisor to 70 000, 200 000 random ones."""
    exe = str(tmp_path / "test_fastdiv")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-fsanitize=undefined", "-fno-sanitize-recover=undefined",
                           os.path.join(ROOT, "tests", "cpp", "test_fastdiv.cpp"), "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and " 0 wrong" in p.stdout, p.stdout + p.stderr[-2000:]
