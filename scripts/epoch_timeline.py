"""Timeline of one policy epoch from a rocprofv3 kernel trace of
scripts/prof_policy.py (rocpd database or *_kernel_trace.csv):
    python scripts/epoch_timeline.py results.db | pol_kernel_trace.csv"""
import csv, sqlite3, sys
if sys.argv[1].endswith(".csv"):
    rows = sorted(((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
                   for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: r[1])
else:
    con = sqlite3.connect(sys.argv[1])
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "kl_cov_proj_fwd" in r[0]]
a, b = idx[-10], idx[-9]
first = max(i for i in range(a) if "mlp_critic_fwd_kernel" in rows[i][0] or "mlp_hidden" in rows[i][0])
ep = rows[first:first + (b - a)]
t0 = ep[0][1]
tot = 0
for n, s, e in ep:
    print("%8.1f %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n[:100]))
    tot += e - s
print("sum of kernel durations %.1f us, span %.1f us, launches %d" % (tot / 1e3, (ep[-1][2] - t0) / 1e3, len(ep)))
