#!/usr/bin/env python3
"""Condenses the rocprofv3 CSVs written by tools/profile.sh into one markdown summary:
per-kernel launch statistics from the kernel trace, and per-kernel, per-launch averages of
every PMC counter collected.  FETCH_SIZE on gfx950 under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md, HBM section); both the raw and the corrected figure are printed."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for tok in ("(anonymous namespace)::", "void ", "hgi::"):
        name = name.replace(tok, "")
    return name.split("(")[0][:70]


def stamp():
    """The build the profile was taken on: bench.build_stamp() (source hash, .so hash, version) + git HEAD when there is one."""
    import subprocess
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    st = bench.build_stamp()
    try:
        st["git_head"] = subprocess.check_output(["git", "rev-parse", "--short=12", "HEAD"], cwd=bench.ROOT, text=True,
                                                 stderr=subprocess.DEVNULL).strip()
    except Exception:
        st["git_head"] = None      # (the GPU box receives a snapshot without .git)
    return st


def main(out):
    frames = int(os.environ.get("HGI_PROF_FRAMES", "512"))
    print("# rocprofv3 summary: %s\n" % os.path.basename(out))
    st = stamp()
    print("workload: `bench.py --frames %d` (%d x 4096^2 u8 ramp(3), level 4, Medium, Crossed); algorithmic bytes per launch %d\n"
          % (frames, frames, 2 * frames * 4096 * 4096))
    print("build: source %s, %s %s, %s\n" % (st.get("source_sha256"), st.get("lib"), st.get("lib_sha256"), st.get("version")))
    traces = glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True)
    dur = defaultdict(list)
    for path in traces:
        for row in csv.DictReader(open(path)):
            dur[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    print("## kernel trace (pass `trace`, --kernel-trace --stats)\n")
    print("| kernel | launches | avg us | min us | max us | total ms |")
    print("|---|---|---|---|---|---|")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print("| %s | %d | %.2f | %.2f | %.2f | %.3f |" % (k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3,
                                                           max(v) / 1e3, sum(v) / 1e6))
    alg = 2.0 * frames * 4096 * 4096
    print("\nagainst the roofline (2 B/px per direction, 8 TB/s), the workload's launches only (Crossed: `k_*_tiles<1, ...>`; warm-up and"
          " settle launches included):\n")
    for k, v in sorted(dur.items()):
        if k.startswith(("k_enc_tiles<1", "k_dec_tiles<1")):
            avg = sum(v) / len(v)
            print("- `%s`: %.2f us average -> %.0f GB/s = **%.3f** of 8 TB/s" % (k, avg / 1e3, alg / avg, alg / avg / 8000))
    pf = defaultdict(list)
    for path in glob.glob(os.path.join(out, "pfine", "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            pf[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    if pf:
        print("\n## P_fine (pass `pfine`: tools/pfine.py under --kernel-trace --stats)\n")
        print("`k_*_tiles` rows: the product kernels at levels = 1 on 64 x 4096^2 (the finest pass alone; 1.75 B/px algorithmic,")
        print("2 B/px moved).  `k_*_level` rows: the level-wise path at levels = 4, one launch per level (4 per direction; the")
        print("slowest of each is the finest pass).\n")
        print("| kernel | launches | avg us | min us | max us | 1.75 B/px GB/s (avg) | frac of 8 TB/s |")
        print("|---|---|---|---|---|---|---|")
        n = 64 * 4096 * 4096
        for k, v in sorted(pf.items(), key=lambda kv: -sum(kv[1])):
            if not k.startswith("k_"):
                continue
            avg = sum(v) / len(v)
            tiles = "_tiles" in k
            gbs = 1.75 * n / avg if tiles else float("nan")
            print("| %s | %d | %.2f | %.2f | %.2f | %s | %s |" % (k, len(v), avg / 1e3, min(v) / 1e3, max(v) / 1e3,
                                                                 "%.0f" % gbs if tiles else "-", "%.3f" % (gbs / 8000) if tiles else "-"))
        try:
            log = open(os.path.join(out, "pfine.log")).read().splitlines()
            print("\n```")
            for line in log:
                if line.startswith(("planes separated", "P_fine", "  encode", "  decode", "level-wise")):
                    print(line)
            print("```")
        except OSError:
            pass
    print("\n## PMC (one pass per group; averages per launch)\n")
    for name in ("fetch", "write", "sq1", "sq2", "tcc"):
        files = glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True)
        acc = defaultdict(lambda: defaultdict(list))
        for path in files:
            for row in csv.DictReader(open(path)):
                acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
        if not acc:
            print("pass `%s`: no counter rows\n" % name)
            continue
        print("### pass `%s`\n" % name)
        for k, ctrs in sorted(acc.items()):
            if not k.startswith("k_"):
                continue
            parts = []
            for c, vals in sorted(ctrs.items()):
                avg = sum(vals) / len(vals)
                if c == "FETCH_SIZE":
                    parts.append("FETCH_SIZE=%.1f KB raw (x2 gfx950 correction = %.1f MB)" % (avg, 2 * avg / 1024))
                elif c == "WRITE_SIZE":
                    parts.append("WRITE_SIZE=%.1f KB (= %.1f MB)" % (avg, avg / 1024))
                else:
                    parts.append("%s=%.4g" % (c, avg))
            print("- `%s` (%d launches): %s" % (k, len(next(iter(ctrs.values()))), "; ".join(parts)))
        print()


def traffic_json(out, frames, size, levels):
    """profiles/*_traffic.json: per-launch HBM bytes of the fused kernels (what bench.py reports as
    roofline.traffic).  FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE x2 is the gfx950 correction for
    wide coalesced reads (MI355X_MICROARCH.md, HBM section)."""
    import json
    per = defaultdict(dict)
    for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        acc = defaultdict(list)
        for path in glob.glob(os.path.join(out, name, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                name = short(row["Kernel_Name"])
                if name.startswith("k_dec_tiles<0,"):      # the placement probe (LeftTop decode on scratch), not the workload
                    continue
                if row["Counter_Name"] == ctr:
                    acc[name.split("<")[0]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            per[k][ctr] = sum(v) / len(v)
    kernels = {}
    for k, c in per.items():
        if k.startswith("k_") and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            kernels[k] = {"fetch_kb_raw": c["FETCH_SIZE"], "write_kb": c["WRITE_SIZE"],
                          "hbm_bytes_per_launch": int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)}
    print(json.dumps({"workload": {"frames": frames, "size": size, "levels": levels}, "build": stamp(),
                      "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024",
                      "kernels": kernels}, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--traffic":
        traffic_json(sys.argv[1], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
    else:
        main(sys.argv[1])
