"""Kernel sequence of one policy epoch (between two adam_apply kernels on the
policy stream) from a rocpd trace: python rocpd_sequence.py db"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end, stream_id, grid_x, workgroup_x from kernels order by start").fetchall()
# policy stream = the one with the most kernels
import collections
cnt = collections.Counter(r[3] for r in rows)
ps = cnt.most_common(1)[0][0]
rows = [r for r in rows if r[3] == ps]
idx = [i for i, r in enumerate(rows) if "adam_apply" in r[0]]
a, b = idx[-3], idx[-2]
seg = rows[a + 1:b + 1]
busy = sum(e - s for _, s, e, *_ in seg)
print("kernels in the epoch:", b - a, " busy %.1f us  span %.1f us" % (busy / 1e3, (seg[-1][2] - rows[a][2]) / 1e3))
prev = rows[a][2]
for n, s, e, st, gx, wx in seg:
    print(f"gap {(s - prev) / 1e3:6.1f}  {(e - s) / 1e3:7.1f} us  grid {gx // max(wx,1):6d} x {wx:4d}  {n[:100]}")
    prev = e
