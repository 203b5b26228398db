"""What runs between the critic epochs of two consecutive steps (the serial
part of an iteration: rollout, dataset processing, bookkeeping), from a
rocprofv3 rocpd kernel trace:   python scripts/rocpd_between.py results.db [name-substring]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
key = sys.argv[2] if len(sys.argv) > 2 else "mlp_critic_bwd_kernel"
rows = con.execute("select name, start, end, queue_id from kernels order by start").fetchall()
crit = [(s, e) for n, s, e, q in rows if key in n and "true" not in n.split(key)[1][:20]]
gaps = [(crit[i + 1][0] - crit[i][1], crit[i][1], crit[i + 1][0]) for i in range(len(crit) - 1)]
big = [g for g in gaps if g[0] > 0.5e6]
print("%d critic launches, %d gaps > 0.5 ms: %s ms" % (len(crit), len(big), [round(g[0] / 1e6, 2) for g in big]))
if not big:
    sys.exit(0)
g, t0, t1 = big[-1]
inside = [r for r in rows if r[1] >= t0 and r[2] <= t1]
busy = sum(e - s for n, s, e, q in inside)
print("last gap: %.3f ms, %d kernels, %.3f ms of kernel time (sum over queues)" % (g / 1e6, len(inside), busy / 1e6))
prev = t0
for n, s, e, q in inside:
    print("  +%8.1f us  gap %7.1f  dur %7.1f us  q%-3d %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, q, n[:90]))
    prev = max(prev, e)
print("  tail gap %.1f us" % ((t1 - prev) / 1e3))
