"""A dependent kernel chain in a rocpd kernel trace: durations and the gaps between
its links.

    python scripts/rocpd_chain.py db LINK1 LINK2 ... [--skip-frac F]

LINKi: substrings of kernel names in chain order (e.g. for a black-box policy epoch:
"smlp_epoch_kernel<32, 4, 2" smlp_reduce bb_diag_finish).  Every occurrence of the
sequence LINK1 -> ... -> LINKn on ONE queue (nothing else of that queue in between)
is one chain; reported: per link the average duration, per boundary the average
gap (start of the next link - end of the previous one), the gap from the last link
to the next chain's first, and  sum(durations) / period  (= 1 - gap_frac).
--skip-frac F: ignore the first F of the trace (warm-up)."""
import collections
import json
import sqlite3
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
skip = 0.0
if "--skip-frac" in sys.argv:
    skip = float(sys.argv[sys.argv.index("--skip-frac") + 1])
    args = [a for a in args if a != sys.argv[sys.argv.index("--skip-frac") + 1]]
db, links = args[0], args[1:]
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = "stream_id" if "stream_id" in cols else "queue_id"
rows = con.execute("select name, start, end, %s from kernels order by start" % qcol).fetchall()
t0, t1 = rows[0][1], rows[-1][2]
cut = t0 + (t1 - t0) * skip
by = collections.defaultdict(list)
for n, s, e, q in rows:
    if s >= cut:
        by[q].append((n, s, e))
best = None
for q, ks in by.items():
    chains = []
    i = 0
    while i + len(links) <= len(ks):
        if all(links[j] in ks[i + j][0] for j in range(len(links))):
            chains.append(ks[i:i + len(links)])
            i += len(links)
        else:
            i += 1
    if chains and (best is None or len(chains) > len(best[1])):
        best = (q, chains)
if best is None:
    raise SystemExit("chain not found")
q, chains = best
n = len(chains)


def med(v):
    v = sorted(v)
    return v[len(v) // 2] if v else 0.0


# medians: a trace holds other iterations too (balance-check epochs, whose chains are
# interleaved with no-Adam launches; a 64-env agent) -- the typical chain is wanted
dur = [med([c[j][2] - c[j][1] for c in chains]) / 1e3 for j in range(len(links))]
gap = [med([c[j + 1][1] - c[j][2] for c in chains]) / 1e3 for j in range(len(links) - 1)]
nxt = [b[0][1] - a[-1][2] for a, b in zip(chains, chains[1:])]
period = [b[0][1] - a[0][1] for a, b in zip(chains, chains[1:])]
out = {"queue": q, "chains": n, "links": links, "statistic": "median over the chains",
       "kernel_us": [round(d, 2) for d in dur],
       "gap_us": [round(g, 2) for g in gap],
       "gap_to_next_chain_us": round(med(nxt) / 1e3, 2),
       "kernel_us_per_chain": round(sum(dur), 2),
       "period_us": round(med(period) / 1e3, 2)}
out["gap_frac"] = round(1.0 - out["kernel_us_per_chain"] / out["period_us"], 4) \
    if out["period_us"] else None
print(json.dumps(out))
