#!/usr/bin/env python3
"""Instruction mix per phase of the interior-tile path of the fused kernels (no GPU needed).

Compiles hgi_fused_{enc,dec}.hip with -DHGI_ANALYZE_K=<k> (interior tile only, constant depth: straight-line
code with `; HGI_MARK <phase>` comments in it) and counts, per phase of the main Crossed / table kernel, the
VALU, SALU, LDS and vector-memory instructions a wave executes.  One wave = one 128 x 64 tile, so
VALU x 4 cycles / tile is the SIMD time a tile costs.

usage: isa_phases.py [k] [extra -D flags ...]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"hgi_fused_enc.hip": "k_enc_tilesILi1ELb0ELi0ELi64", "hgi_fused_dec.hip": "k_dec_tilesILi1ELi0ELi64"}


def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def phases(asm, kernel):
    lines = asm.split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and kernel in l and l.rstrip().endswith(("EEj", "Ej:")) or
                 (l.startswith("_ZN") and kernel in l and ":" in l))
    out = collections.OrderedDict()
    cur = "prologue"
    branches = 0
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith(".Lfunc_end"):
            break
        m = re.match(r";\s*HGI_MARK (\w+)", t)
        if m:
            cur = m.group(1)
            if cur == "end":
                break
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        if t.startswith("s_cbranch") or t.startswith("s_branch"):
            branches += 1
        out.setdefault(cur, collections.Counter())[classify(t)] += 1
    return out, branches


def main():
    k = sys.argv[1] if len(sys.argv) > 1 else "4"
    extra = sys.argv[2:]
    hipcc = "/opt/rocm/bin/hipcc"
    for tu, kernel in KERNELS.items():
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "a.s")
            subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                                   "-DHGI_ANALYZE_K=" + k] + extra + [os.path.join(ROOT, "rustyhgi_amd", "csrc", tu), "-o", out],
                                  stderr=subprocess.DEVNULL)
            asm = open(out).read()
        ph, branches = phases(asm, kernel)
        print("## %s  (k = %s, branches in path: %d)" % (kernel, k, branches))
        print("| phase | VALU | SALU | LDS | VMEM | waitcnt | nop |")
        print("|---|---|---|---|---|---|---|")
        tot = collections.Counter()
        merged = collections.OrderedDict()
        for name, c in ph.items():
            merged.setdefault(name, collections.Counter()).update(c)
        for name, c in merged.items():
            tot.update(c)
            print("| %s | %d | %d | %d | %d | %d | %d |" % (name, c["valu"], c["salu"], c["lds"], c["vmem"], c["wait"], c["nop"]))
        print("| **total** | %d | %d | %d | %d | %d | %d |" % (tot["valu"], tot["salu"], tot["lds"], tot["vmem"], tot["wait"], tot["nop"]))
        print()


if __name__ == "__main__":
    main()
