"""Kernel-stats summary (same columns as rocprofv3's *_kernel_stats.csv) from a
rocprofv3 rocpd database:  python scripts/rocpd_stats.py results.db out.csv"""
import csv
import sqlite3
import statistics
import sys

db, out = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
rows = con.execute("select name, end - start from kernels").fetchall()
by = {}
for name, d in rows:
    by.setdefault(name, []).append(d)
total = sum(sum(v) for v in by.values())
with open(out, "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage",
                "MinNs", "MaxNs", "StdDev"])
    for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([name, len(v), sum(v), round(sum(v) / len(v), 3),
                    round(100.0 * sum(v) / total, 2), min(v), max(v),
                    round(statistics.pstdev(v), 3)])
print(f"{len(rows)} dispatches, {len(by)} kernels, {total / 1e6:.2f} ms")
