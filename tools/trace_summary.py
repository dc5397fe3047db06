"""Print the launches of a rocprofv3 --kernel-trace CSV in time order (last N), with durations and gaps."""
import csv, glob, sys
d = sys.argv[1]; last = int(sys.argv[2]) if len(sys.argv) > 2 else 24
f = sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
for r in rows[-last:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%-58s %9.1f us   gap before %7.1f us   grid %s" % (r["Kernel_Name"][:58], (e - s) / 1e3, gap, r.get("Grid_Size", "?")))
    prev_end = e
