"""Coarse timeline of the last step of a rocpd kernel trace: per queue, kernel
count / busy time in 10 ms buckets.   python rocpd_timeline.py db [span_ms]"""
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
span = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 250e6
rows = con.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
t1 = rows[-1][2]
rows = [r for r in rows if r[1] >= t1 - span]
t0 = rows[0][1]
B = 10e6
table = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for n, s, e, q, st in rows:
    b = int((s - t0) // B)
    table[(q, st)][b][0] += 1
    table[(q, st)][b][1] += (e - s) / 1e6
nb = int((t1 - t0) // B) + 1
for key in sorted(table):
    print("queue/stream", key)
    print("  " + " ".join(f"{table[key][b][0]:5d}" for b in range(nb)))
    print("  " + " ".join(f"{table[key][b][1]:5.1f}" for b in range(nb)))
if len(sys.argv) > 3:
    for key in sorted(table):
        ks = [r for r in rows if (r[3], r[4]) == key][: int(sys.argv[3])]
        print("first kernels of", key)
        for n, s, e, q, st in ks:
            print(f"   {(s - t0) / 1e6:9.3f} ms  {(e - s) / 1e3:9.1f} us  {n[:70]}")
