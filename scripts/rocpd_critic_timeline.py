"""Start times / gaps of the critic's backward launches of the last step in a rocpd trace,
with the policy tails' span:   python scripts/rocpd_critic_timeline.py db"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
gx = [c for c in ("grid_x", "grid_size_x", "grid_size") if c in cols][0]
wx = [c for c in ("workgroup_x", "workgroup_size_x", "workgroup_size") if c in cols][0]
rows = con.execute("select name, start, end, %s, %s from kernels order by start" % (gx, wx)).fetchall()
cr = [r for r in rows if "mlp_critic_bwd_kernel<1, 10, false>" in r[0]]
cr = cr[-50:]
t0 = cr[0][1]
tails = [r for r in rows if "policy_tail_kernel" in r[0] and r[1] >= t0]
print("policy tails: first %.2f ms, last %.2f ms after the first critic launch (%d)" % (
    (tails[0][1] - t0) / 1e6, (tails[-1][2] - t0) / 1e6, len(tails)))
prev = None
for i, r in enumerate(cr):
    gap = (r[1] - prev) / 1e3 if prev else 0.0
    prev = r[2]
    print("epoch %2d  start %8.2f ms  dur %7.1f us  wg %3d  gap before %8.1f us" % (
        i, (r[1] - t0) / 1e6, (r[2] - r[1]) / 1e3, r[3] // max(r[4], 1), gap))
