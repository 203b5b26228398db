"""Kernel sequence around an anchor kernel in a rocpd trace, all queues interleaved:
    python scripts/rocpd_seq.py db ANCHOR_SUBSTR [OCCURRENCE_FROM_END] [N_BEFORE] [N_AFTER]"""
import re, sqlite3, sys
con = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2]
occ = int(sys.argv[3]) if len(sys.argv) > 3 else 50
nb = int(sys.argv[4]) if len(sys.argv) > 4 else 10
na = int(sys.argv[5]) if len(sys.argv) > 5 else 40
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = "stream_id" if "stream_id" in cols else "queue_id"
rows = con.execute("select name, start, end, %s from kernels order by start" % qcol).fetchall()
idx = [i for i, r in enumerate(rows) if anchor in r[0]]
c = idx[-occ]
rows = rows[max(c - nb, 0):c + na]
b = rows[0][1]
last = {}
for nm, s, e, q in rows:
    gap = (s - last[q]) / 1e3 if q in last else 0.0
    last[q] = e
    short = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", nm).split("(")[0][:52]
    print("q%-2s %9.2f us  dur %7.2f  gap(same q) %7.2f  %s" % (q, (s - b) / 1e3, (e - s) / 1e3, gap, short))
