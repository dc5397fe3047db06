#!/usr/bin/env python3
"""Static checks on the gfx950 ISA of the fused kernels (no GPU needed).

1. gfx940+ forwarding hazard: a VALU op that writes only part of a VGPR (SDWA dst_sel BYTE_n / WORD_n)
   must not be followed immediately by a VALU op that reads that VGPR (one wait state required).  The
   hand-written SDWA byte chains of hgi_fused_impl.h rely on the compiler padding between asm statements;
   this verifies the padding is there in the build that ships.
2. No scratch (spills) in any kernel.
3. No DPP instruction: the kernels' SDWA statements are inline asm, invisible to the compiler's hazard
   recognizer, and DPP has multi-cycle VALU -> read hazards (a DPP build was observed to corrupt data).
Usage: check_isa.py <file.s>   (hipcc --offload-arch=gfx950 -O3 --cuda-device-only -S ... -o file.s)
"""
import re
import sys


def check(path):
    text = open(path).read()
    lines = [l.strip() for l in text.split("\n")]
    real = [l for l in lines if l and not l.startswith((";", ".", "//")) and not l.endswith(":")]
    partial = adjacent = 0
    examples = []
    for cur, nxt in zip(real, real[1:]):
        m = re.match(r"(v_\w+_sdwa)\s+(v\d+)\b", cur)
        if not m or not re.search(r"dst_sel:(BYTE|WORD)_", cur):
            continue
        partial += 1
        ops = nxt.split(None, 1)
        if nxt.startswith("v_") and len(ops) > 1 and re.search(r"\b" + m.group(2) + r"\b", ops[1]):
            adjacent += 1
            examples.append((cur, nxt))
    dpp = sum(1 for l in real if re.search(r"_dpp\b|row_sh[lr]:|quad_perm:|row_bcast|wave_sh", l))
    scratch = [int(v) for v in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", text)]
    spills = [int(v) for v in re.findall(r"\.vgpr_spill_count:\s+(\d+)", text)]
    return dict(partial_writes=partial, adjacent_dependent=adjacent, examples=examples[:5],
                kernels=len(scratch), scratch_bytes=max(scratch or [0]), vgpr_spills=max(spills or [0]), dpp=dpp)


if __name__ == "__main__":
    r = check(sys.argv[1])
    print({k: v for k, v in r.items() if k != "examples"})
    for cur, nxt in r["examples"]:
        print("  ", cur, "\n     ->", nxt)
    sys.exit(1 if r["adjacent_dependent"] or r["scratch_bytes"] or r["vgpr_spills"] or r["dpp"] else 0)
