"""Per-stream busy time / gaps from a rocpd kernel trace: python rocpd_gaps.py db [t0_frac]"""
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
print(cols)
q = "select name, start, end, " + ("stream_id" if "stream_id" in cols else "queue_id") + " from kernels order by start"
rows = con.execute(q).fetchall()
t0, t1 = rows[0][1], rows[-1][2]
cut = t0 + (t1 - t0) * float(sys.argv[2]) if len(sys.argv) > 2 else t0
by = collections.defaultdict(list)
for n, s, e, q in rows:
    if s >= cut:
        by[q].append((n, s, e))
for q, ks in by.items():
    busy = sum(e - s for _, s, e in ks)
    span = ks[-1][2] - ks[0][1]
    print(f"queue {q}: {len(ks)} kernels, busy {busy/1e6:.1f} ms, span {span/1e6:.1f} ms")
    agg = collections.defaultdict(lambda: [0, 0])
    for n, s, e in ks:
        agg[n][0] += 1; agg[n][1] += e - s
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"   {n[:80]:80s} {c:6d} {t/c/1e3:8.1f} us {t/1e6:8.1f} ms")
