#!/usr/bin/env python3
"""profiles/r03_c4_summary.md from the rocprofv3 CSVs of tools/c4_profile.sh: the launches that make up ONE C4 encode
and ONE C4 decode (order, gaps, durations: averages over the timed calls), each call's device span against the
roofline, and the HBM traffic of every kernel (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes)."""
import csv, glob, os, sys
from collections import defaultdict

N = 16384 * 16384
ALG = 2 * N


def short(name):
    for tok in ("(anonymous namespace)::", "void ", "hgi::"):
        name = name.replace(tok, "")
    return name.split("(")[0][:64]


def rows_of(out, sub, pattern):
    rows = []
    for path in glob.glob(os.path.join(out, sub, "**", pattern), recursive=True):
        rows += list(csv.DictReader(open(path)))
    return rows


def main(out):
    tr = rows_of(out, "trace", "*kernel_trace.csv")
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in tr))
    # a "call" = the launches between two main tile launches; the main launches are the long k_*_tiles ones
    calls, cur = [], []
    for s, e, k in ks:
        if k.startswith(("k_synth", "k_copy", "k_diff")) or "k_dec_tiles<0," in k:
            cur = []
            continue
        cur.append((s, e, k))
        if "_tiles" in k and e - s > 40000:      # > 40 us: the seeded main launch closes the call
            calls.append(cur); cur = []
    print("# C4 -- 16384 x 16384 u8 ramp(4), level 8, High, Crossed -- rocprofv3 record (%s)\n" % os.path.basename(out))
    for log in ("trace.log",):
        try:
            for line in open(os.path.join(out, log)):
                if line.startswith("C4 hipEvents"):
                    print("`%s` (this pass, profiler attached)\n" % line.strip())
        except OSError:
            pass
    for direction, tag in (("encode", "k_enc_tiles"), ("decode", "k_dec_tiles")):
        sel = [c for c in calls if tag in c[-1][2]]
        sel = sel[len(sel) // 2:]              # the timed half (warm)
        if not sel:
            continue
        shape = [k for _, _, k in sel[-1]]
        same = [c for c in sel if [k for _, _, k in c] == shape]
        print("## one %s call: %d launches (averages over %d calls of that shape)\n" % (direction, len(shape), len(same)))
        print("| # | kernel | start after call begin (us) | duration (us) | gap to previous end (us) |")
        print("|---|---|---|---|---|")
        for i, k in enumerate(shape):
            st = sum(c[i][0] - c[0][0] for c in same) / len(same) / 1e3
            du = sum(c[i][1] - c[i][0] for c in same) / len(same) / 1e3
            gap = sum(c[i][0] - c[i - 1][1] for c in same) / len(same) / 1e3 if i else 0.0
            print("| %d | `%s` | %.2f | %.2f | %s |" % (i + 1, k, st, du, "%.2f" % gap if i else "-"))
        span = sum(c[-1][1] - c[0][0] for c in same) / len(same) / 1e3
        main_d = sum(c[-1][1] - c[-1][0] for c in same) / len(same) / 1e3
        spans = sorted((c[-1][1] - c[0][0]) / 1e3 for c in same)
        print("\ndevice span of the call %.1f us = %.0f GB/s algorithmic (2 x %d B) = **%.3f of 8 TB/s**; the main launch alone %.1f us = %.3f; "
              "spans of the %d calls: min %.1f / median %.1f / max %.1f us\n" % (
                  span, ALG / span / 1e3, N, ALG / span / 8e6, main_d, ALG / main_d / 8e6, len(spans), spans[0], spans[len(spans) // 2], spans[-1]))
    # all kernels, stats
    dur = defaultdict(list)
    for s, e, k in ks:
        dur[k].append(e - s)
    print("## kernel trace, all launches of the run\n\n| kernel | launches | avg us | min us | max us |\n|---|---|---|---|---|")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print("| `%s` | %d | %.2f | %.2f | %.2f |" % (k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
    print("\n## HBM traffic per launch (PMC, separate passes)\n")
    print("| kernel | launches | FETCH_SIZE raw KB | x2 (gfx950) MB | WRITE_SIZE MB | total MB | vs algorithmic |\n|---|---|---|---|---|---|---|")
    per = defaultdict(dict)
    for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        acc = defaultdict(list)
        for r in rows_of(out, sub, "*counter_collection.csv"):
            if r["Counter_Name"] == ctr:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            per[k][ctr] = (sum(v) / len(v), len(v))
    for k, c in sorted(per.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", (0, 0))[0]):
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c or k.startswith("k_synth"):
            continue
        f, nl = c["FETCH_SIZE"]; w = c["WRITE_SIZE"][0]
        tot = (2 * f + w) / 1024
        big = "_tiles" in k and tot > 100
        print("| `%s` | %d | %.0f | %.1f | %.1f | %.1f | %s |" % (k, nl, f, 2 * f / 1024, w / 1024, tot,
                                                              "%.4fx of %.1f MB" % (tot * 1048576 / ALG, ALG / 1048576) if big else "-"))


if __name__ == "__main__":
    main(sys.argv[1])
