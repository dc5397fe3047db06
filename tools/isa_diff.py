#!/usr/bin/env python3
"""Per-kernel comparison of two device-ISA listings (hipcc -S --cuda-device-only): instructions only, comments dropped,
basic-block labels renumbered per function -- so that a source clean-up can be shown to leave the shipped kernels' code
unchanged.  Usage: isa_diff.py before_dir after_dir"""
import os
import re
import sys


def kernels(path):
    out, cur, body = {}, None, []
    for line in open(path):
        line = line.rstrip("\n")
        m = re.match(r"^(_Z\w+):\s", line + " ")
        if m:
            cur, body = m.group(1), []
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            out[cur], cur = body, None
            continue
        l = line.strip()
        if not l or l.startswith((";", ".", "//")):
            continue
        l = re.sub(r"\.LBB\d+_", ".LBB_", re.sub(r"\s*;.*$", "", l))
        # the kernel's index in the module's LDS lookup table, handed to non-inlined callees in s15: it follows the NUMBER of
        # kernels in the unit, not their code
        l = re.sub(r"^s_mov_b32 s15, \d+$", "s_mov_b32 s15, <kernel id>", l)
        body.append(l)
    return out


def main(before, after):
    bad = 0
    for name in sorted(os.listdir(before)):
        if not name.endswith(".s") or not os.path.exists(os.path.join(after, name)):
            continue
        a, b = kernels(os.path.join(before, name)), kernels(os.path.join(after, name))
        changed = [k for k in a if k in b and a[k] != b[k]]
        print("%-20s kernels %2d -> %2d  identical %2d  changed %d  removed %d  added %d"
              % (name, len(a), len(b), len([k for k in a if k in b and a[k] == b[k]]), len(changed),
                 len([k for k in a if k not in b]), len([k for k in b if k not in a])))
        for k in [k for k in a if k not in b]:
            print("    removed:", k)
        for k in changed:
            print("    CHANGED:", k)
        bad += len(changed)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
