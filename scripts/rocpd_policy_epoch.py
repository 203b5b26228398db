"""Per-kernel average duration and start-to-start wait on the policy stream's queue
during the last step of a rocpd trace: python scripts/rocpd_policy_epoch.py db"""
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end, queue_id from kernels order by start").fetchall()
t_end = rows[-1][2]
rows = [r for r in rows if r[1] > t_end - 150e6]
byq = collections.defaultdict(list)
for r in rows: byq[r[3]].append(r)
for q, rs in byq.items():
    if not any("pair_env" in r[0] for r in rs): continue
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    prev_end = None
    for n, s, e, _ in rs:
        k = n.replace("(anonymous namespace)::", "").replace("void ", "")
        k = k.split("(")[0][:60]
        a = agg[k]; a[0] += 1; a[1] += (e - s) / 1e3
        if prev_end is not None: a[2] += max(0.0, (s - prev_end)) / 1e3
        prev_end = e
    print("queue", q, len(rs), "kernels")
    for k, (c, d, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:25]:
        print("  %-62s n %4d  dur %8.1f us  gap-before %8.1f us" % (k, c, d / c, g / c))
