"""Kernel totals and a timeline of the LAST call out of a rocprofv3 (rocpd) database: python tools/prof_db.py x.db [first-kernel-substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
mark = sys.argv[2] if len(sys.argv) > 2 else "k_token_hist"
print("%-60s %6s %10s %10s" % ("kernel", "calls", "avg us", "total ms"))
for name, cnt, avg, tot in db.execute("select name, count(*), avg(end-start)/1e3, sum(end-start)/1e6 from kernels group by name order by 4 desc"):
    print("%-60s %6d %10.1f %10.2f" % (name.split("(")[0][-60:], cnt, avg, tot))
ks = list(db.execute("select start, end, name from kernels order by start"))
idx = [i for i, k in enumerate(ks) if mark in k[2]]
if idx:
    # the last call starts at the last marked kernel that is not directly preceded by another call's tail
    i0 = idx[-1]
    seq = ks[max(0, i0 - 2):]
    t0 = seq[0][0]
    print("\ntimeline of the last call (us from its first kernel):")
    for s, e, n in seq[:60]:
        print("%10.1f %9.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n.split("(")[0][-50:]))
    if len(seq) > 60:
        print("   ... %d more; last ends at %.1f us" % (len(seq) - 60, (seq[-1][1] - t0) / 1e3))
