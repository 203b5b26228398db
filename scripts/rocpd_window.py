"""All kernels (every queue) in a window of a rocpd trace, in start order:
python scripts/rocpd_window.py db offset_ms_before_end length_ms"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
off, ln = float(sys.argv[2]) * 1e6, float(sys.argv[3]) * 1e6
rows = con.execute("select name, start, end, queue_id from kernels order by start").fetchall()
t_end = rows[-1][2]
t0 = t_end - off
for n, s, e, q in rows:
    if s >= t0 and s <= t0 + ln:
        k = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
        print("%9.1f us  dur %8.1f  q%-2d %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, k))
