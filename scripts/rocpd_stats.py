"""Kernel-stats summary (same columns as rocprofv3's *_kernel_stats.csv) from a
rocprofv3 rocpd database:

    python scripts/rocpd_stats.py results.db out.csv [--by-grid] [--after-ms T]

--by-grid: one row per (kernel, workgroup count) instead of per kernel -- the
critic's persistent grid runs with 224 workgroups beside the policy stream and
with 256 once that stream has drained, and the two have different durations; a
column "Workgroups" is added.  --after-ms T: only dispatches that start at
least T ms after the first one (e.g. to leave the warm-up steps out).
--between-markers A B: only dispatches between the LAST tce_marker_kernel launch
of A workgroups and the first one of B workgroups after it (bench.py marks its
timed steps with 1 and 2: the summary then covers exactly the window `value` is
computed from); prints the window's length."""
import csv
import sqlite3
import statistics
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
db, out = args[0], args[1]
by_grid = "--by-grid" in sys.argv
after = 0.0
if "--after-ms" in sys.argv:
    after = float(sys.argv[sys.argv.index("--after-ms") + 1]) * 1e6
markers = None
if "--between-markers" in sys.argv:
    i = sys.argv.index("--between-markers")
    markers = (int(sys.argv[i + 1]), int(sys.argv[i + 2]))
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]


def pick(*names):
    for n in names:
        if n in cols:
            return n
    return None


gx, wx = pick("grid_x", "grid_size_x", "grid_size"), \
    pick("workgroup_x", "workgroup_size_x", "workgroup_size")
if by_grid and not (gx and wx):
    raise SystemExit("no grid / workgroup columns in `kernels`: %s" % cols)
if markers and not (gx and wx):
    raise SystemExit("no grid / workgroup columns in `kernels`: %s" % cols)
need_grid = by_grid or markers is not None
sel = "name, start, end - start" + (", %s, %s" % (gx, wx) if need_grid else "")
rows = con.execute("select %s from kernels" % sel).fetchall()
t0 = min(r[1] for r in rows)
lo, hi = None, None
if markers:
    marks = sorted((r[1], r[1] + r[2], r[3] // max(r[4], 1)) for r in rows
                   if "tce_marker_kernel" in r[0])
    starts = [m for m in marks if m[2] == markers[0]]
    if not starts:
        raise SystemExit("no tce_marker_kernel of %d workgroups in the trace" % markers[0])
    lo = starts[-1][1]
    ends = [m for m in marks if m[2] == markers[1] and m[0] > lo]
    if not ends:
        raise SystemExit("no closing tce_marker_kernel of %d workgroups" % markers[1])
    hi = ends[0][0]
    print("window between the markers: %.3f ms" % ((hi - lo) / 1e6))
by = {}
for r in rows:
    if r[1] - t0 < after:
        continue
    if lo is not None and not (lo <= r[1] < hi):
        continue
    if "tce_marker_kernel" in r[0]:
        continue
    key = (r[0], (r[3] // max(r[4], 1)) if by_grid else None)
    by.setdefault(key, []).append(r[2])
total = sum(sum(v) for v in by.values())
with open(out, "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name"] + (["Workgroups"] if by_grid else []) +
               ["Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs",
                "MaxNs", "StdDev"])
    for (name, wg), v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([name] + ([wg] if by_grid else []) +
                   [len(v), sum(v), round(sum(v) / len(v), 3),
                    round(100.0 * sum(v) / total, 2), min(v), max(v),
                    round(statistics.pstdev(v), 3)])
print(f"{sum(len(v) for v in by.values())} dispatches, {len(by)} rows, {total / 1e6:.2f} ms")
