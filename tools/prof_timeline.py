"""All device activity of the last N ms of a rocprofv3 database, merged per stream: python tools/prof_timeline.py x.db [ms]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
span = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 6e6
ks = list(db.execute("select start, end, name, stream_id, queue_id from kernels order by start"))
t1 = ks[-1][1]
ks = [k for k in ks if k[0] >= t1 - span]
t0 = ks[0][0]
def short(n):
    n = n.split("(anonymous namespace)::")[-1].split("(")[0]
    return n[-24:]
for s, e, n, st, q in ks:
    if "copyBuffer" in n and e - s < 20000:
        continue
    print("%9.1f %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, short(n)))
